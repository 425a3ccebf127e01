#!/bin/bash
# bench.py over several builds of the library (kernel-variant sweep): prints value / per-attempt kernel time / roofline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset TFDIFFEQ_AMD_LIB; else export TFDIFFEQ_AMD_LIB=$PWD/tfdiffeq_amd/$lib; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/sweep_$lib.json 2> gpurun_out/sweep_$lib.err
  python - "$lib" <<'PY'
import json, sys
lib = sys.argv[1]
try:
    r = json.load(open('gpurun_out/sweep_%s.json' % lib))
    print("%-22s value %.3e  ms/step %.3f  attempt-kernels %.4f ms  dominant %.4f ms (%.1f %s, frac %.3f, %s) attempts %d" % (
        lib, r['value'], r['ms_per_step'], r['config']['attempt_kernels_ms'],
        r['roofline']['avg_launch_ms'], r['roofline']['achieved'], r['roofline']['unit'], r['roofline']['frac'], r['roofline']['bound'], r['config']['attempts_per_step']))
except Exception as e:
    print(lib, 'FAILED', e); print(open('gpurun_out/sweep_%s.err' % lib).read()[-1500:])
PY
done
