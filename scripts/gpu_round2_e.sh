#!/bin/bash
# GPU visit E of round 2: the whole GPU suite, smoke(), the bench line, the training-step timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_e.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err; tail -c 1500 gpurun_out/bench_e.json
python scripts/adjoint_train_step.py both 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/adjoint_train_step.log
