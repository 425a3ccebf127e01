#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of libmi_ode (the reference has no sanitizer
# tooling; the native library here does pointer / lifetime work the reference never did).  Device code is compiled as
# usual (-fno-gpu-sanitize).  Output: tfdiffeq_amd/_asan/libmi_ode.so and tests/c_abi/c_abi_smoke_asan (run it on a GPU
# box: scripts/gpu_asan.sh).  The translation units compile in parallel (round 6: one hipcc invocation over all of them took > 25 min).
set -e
cd "$(dirname "$0")/.."
mkdir -p tfdiffeq_amd/_asan/obj
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"
NAMES="mi_ode_api mi_ode_launch_f64 mi_ode_launch_f32 mi_ode_launch_mlp mi_ode_launch_mlp_persist mi_ode_launch_mlp64 mi_ode_adjoint mi_ode_opaque mi_ode_outer mi_ode_linadj"
for n in $NAMES; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -ffp-contract=off $SAN -c tfdiffeq_amd/csrc/$n.hip -o tfdiffeq_amd/_asan/obj/$n.o ) &
done
wait
OBJS=""; for n in $NAMES; do OBJS="$OBJS tfdiffeq_amd/_asan/obj/$n.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -shared -fPIC $OBJS -o tfdiffeq_amd/_asan/libmi_ode.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off $SAN -I include tests/c_abi/c_abi_smoke.cpp \
  -L tfdiffeq_amd/_asan -lmi_ode -Wl,-rpath,'$ORIGIN/../../tfdiffeq_amd/_asan' -o tests/c_abi/c_abi_smoke_asan
rm -rf tfdiffeq_amd/_asan/obj
ls -la tfdiffeq_amd/_asan/libmi_ode.so tests/c_abi/c_abi_smoke_asan
