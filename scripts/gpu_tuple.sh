#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_tuple_fused.py -m gpu -x -q 2>&1 | tail -30
