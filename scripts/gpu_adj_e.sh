#!/bin/bash
# training step at config 5's shape: timings + rocprofv3 kernel trace of the fused step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD; export PYTHONPATH=$R
mkdir -p gpurun_out
timeout 600 python scripts/adjoint_train_step.py both 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/adjoint_train_step.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_adjstep" -o r -- python "$R/scripts/adjoint_train_step.py" fused 5 > "$R/gpurun_out/prof_adjstep.log" 2>&1)
head -12 gpurun_out/prof_adjstep/r_kernel_stats.csv | cut -c1-200
