#!/bin/bash
# rocprofv3 of bench.py: kernel trace + stats, then separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# usage: gpu_prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>_<COUNTER>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-default}; shift || true
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -o r -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$R/gpurun_out/rocprof_$TAG.log" 2>&1)
echo "rocprof[$TAG] exit $?"
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); do cut -c1-200 "$f" | head -12; done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$C
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$C" -o r -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$R/gpurun_out/pmc_${TAG}_$C.log" 2>&1)
  echo "pmc[$TAG] $C exit $?"
done
