#!/bin/bash
# GPU visit F of round 2: full GPU suite, smoke, bench (config 4 + config 5), training-step timing + kernel trace, adjoint pass counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu_f.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; cut -c1-400 gpurun_out/bench_f.json
python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_f_c5.json 2>/dev/null; cut -c1-300 gpurun_out/bench_f_c5.json
python scripts/adjoint_train_step.py both 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/adjoint_train_step.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_adjstep" -o r -- python "$R/scripts/adjoint_train_step.py" fused 5 > "$R/gpurun_out/prof_adjstep.log" 2>&1)
head -4 gpurun_out/prof_adjstep/r_kernel_stats.csv | cut -c1-160
ADJ_MODES="2 3" bash scripts/gpu_adj_b.sh 2>&1 | grep -E "bench|MFMA_BUSY|GUI_ACTIVE|WAIT_INST_ANY|WAVE_CYCLES|EA0_RDREQ"
