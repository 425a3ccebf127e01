#!/bin/bash
# first GPU contact of the fused adjoint kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
MI_ODE_ADJOINT_PROF=1 PYTHONPATH=. timeout 600 python scripts/adj_debug.py > gpurun_out/adj_debug.log 2>&1
echo "exit $?" >> gpurun_out/adj_debug.log
tail -40 gpurun_out/adj_debug.log
