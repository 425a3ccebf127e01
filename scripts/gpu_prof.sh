#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC passes) of bench.py; CSV summaries land in gpurun_out/prof*/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o r01 -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/rocprof.log" 2>&1)
echo "rocprof exit $?"
find gpurun_out/prof -type f | head -20
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cut -c1-250 "$f" | head -30; done
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete; true
if [ "${1:-}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$C
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_$C" -o r01 -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc_$C.log" 2>&1)
    echo "pmc $C exit $?"
    find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -2
  done
fi
