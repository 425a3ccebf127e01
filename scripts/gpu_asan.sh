#!/bin/bash
# Runs the standalone C-ABI smoke test against the ASan/UBSan build (scripts/build_asan.sh) on the GPU box.
# (tfdiffeq_amd/_asan is listed in .gpurunignore - 130 MB that no other call needs: take its lines out for the one gpurun call that runs this script.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/asan
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1
timeout 600 tests/c_abi/c_abi_smoke_asan > gpurun_out/asan/run.log 2>&1; echo "exit $?" >> gpurun_out/asan/run.log
# (ROCm's ASan keeps freed DEVICE allocations in its quarantine and may recycle them from the HSA runtime's own static destructors, after
#  the runtime marked itself unloaded: "CHECK failed: sanitizer_allocator_device.h ... dev_runtime_unloaded_" at process exit, in
#  libhsa-runtime64 frames only.  A second run without the quarantine shows the same program ending cleanly.)
if grep -q "dev_runtime_unloaded_" gpurun_out/asan/run.log; then
  echo "# second run, ASAN_OPTIONS += quarantine_size_mb=0:thread_local_quarantine_size_kb=0" >> gpurun_out/asan/run.log
  ASAN_OPTIONS=$ASAN_OPTIONS:quarantine_size_mb=0:thread_local_quarantine_size_kb=0 timeout 600 tests/c_abi/c_abi_smoke_asan >> gpurun_out/asan/run.log 2>&1; echo "exit $?" >> gpurun_out/asan/run.log
fi
tail -30 gpurun_out/asan/run.log
grep -c "ERROR: AddressSanitizer\|runtime error" gpurun_out/asan/run.log
