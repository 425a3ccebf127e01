#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_adjoint_fused.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/adj_tests.log
cat gpurun_out/adj_tests.log
