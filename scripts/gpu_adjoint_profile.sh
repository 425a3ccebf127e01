#!/bin/bash
# rocprofv3 kernel trace of ten training steps of ODEBlock(adjoint=True) at config 5's shape (time-independent and time-dependent network),
# and the generic adjoint of the linear system for DESIGN section 8.  Output: gpurun_out/profiles_r04/r04_adjoint_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD TMPDIR=/tmp; P=$PWD/gpurun_out/profiles_r04; mkdir -p $P
for td in 0 1; do
  D=$PWD/gpurun_out/prof_adj$td; rm -rf $D
  (cd /tmp && ADJ_TD=$td rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $OLDPWD/scripts/adjoint_train_step.py fused 10 > $P/r04_adjoint_td${td}_train_step.txt 2>&1)
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" > $P/r04_adjoint_td${td}_kernel_stats.csv
  grep ms_per $P/r04_adjoint_td${td}_train_step.txt | cut -c1-250
  find $D -name "*.csv" -size +2M -delete
done
