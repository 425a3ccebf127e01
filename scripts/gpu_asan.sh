#!/bin/bash
# Runs the standalone C-ABI smoke test against the ASan/UBSan build (scripts/build_asan.sh) on the GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/asan
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
timeout 600 tests/c_abi/c_abi_smoke_asan > gpurun_out/asan/run.log 2>&1; echo "exit $?" >> gpurun_out/asan/run.log
tail -30 gpurun_out/asan/run.log
grep -c "ERROR: AddressSanitizer\|runtime error" gpurun_out/asan/run.log
