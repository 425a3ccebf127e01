#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD
tests/c_abi/c_abi_smoke 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "dopri8_on_the_linear" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_adjoint_fused.py -m gpu -x -q 2>&1 | tail -5
MI_ODE_ADJOINT_BENCH=3,10 python scripts/adj_bench.py 2 2>&1 | grep bench
bash scripts/gpu_asan.sh
