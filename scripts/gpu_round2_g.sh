#!/bin/bash
# GPU visit G of round 2 (confirmation of the final tree): full GPU suite, smoke, bench (config 4 + config 5), training step, adjoint soak
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu_g.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err; cut -c1-400 gpurun_out/bench_g.json
python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_g_c5.json 2>/dev/null; cut -c1-300 gpurun_out/bench_g_c5.json
python scripts/adjoint_train_step.py both 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/adjoint_train_step_g.log
timeout 100 python -u scripts/soak_adjoint.py 15 11 2>&1 | grep -v amdgpu.ids | tee gpurun_out/soak_adjoint_g.log
