#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== rocm-smi ===" > gpurun_out/env.log
rocm-smi --showproductname --showmeminfo vram 2>&1 | head -30 >> gpurun_out/env.log
lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread" >> gpurun_out/env.log
echo "=== smoke ===" | tee gpurun_out/smoke.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/smoke.log 2>&1
echo "smoke exit $?" | tee -a gpurun_out/smoke.log
echo "=== pytest gpu ==="
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
echo "=== bench ==="
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --linear-variant 1 > gpurun_out/bench_valu.json 2>> gpurun_out/bench.err
cat gpurun_out/bench_valu.json
echo "=== rocprof ==="
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
echo "rocprof exit $?"
find gpurun_out/prof -name "*stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -25 "$f"; done
# keep the trace small: only stats files travel back
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
