#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of libmi_ode (the reference has no sanitizer
# tooling; the native library here does pointer / lifetime work the reference never did).  Device code is compiled as
# usual (-fno-gpu-sanitize).  Output: tfdiffeq_amd/_asan/libmi_ode.so and tests/c_abi/c_abi_smoke_asan (run it on a GPU
# box: scripts/gpu_asan.sh).  Takes ~4 minutes.
set -e
cd "$(dirname "$0")/.."
mkdir -p tfdiffeq_amd/_asan
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -ffp-contract=off $SAN -shared \
  tfdiffeq_amd/csrc/mi_ode_api.hip tfdiffeq_amd/csrc/mi_ode_launch_f64.hip tfdiffeq_amd/csrc/mi_ode_launch_f32.hip tfdiffeq_amd/csrc/mi_ode_launch_mlp.hip tfdiffeq_amd/csrc/mi_ode_launch_mlp_persist.hip tfdiffeq_amd/csrc/mi_ode_launch_mlp64.hip tfdiffeq_amd/csrc/mi_ode_adjoint.hip tfdiffeq_amd/csrc/mi_ode_opaque.hip tfdiffeq_amd/csrc/mi_ode_outer.hip tfdiffeq_amd/csrc/mi_ode_linadj.hip -o tfdiffeq_amd/_asan/libmi_ode.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off $SAN -I include tests/c_abi/c_abi_smoke.cpp \
  -L tfdiffeq_amd/_asan -lmi_ode -Wl,-rpath,'$ORIGIN/../../tfdiffeq_amd/_asan' -o tests/c_abi/c_abi_smoke_asan
ls -la tfdiffeq_amd/_asan/libmi_ode.so tests/c_abi/c_abi_smoke_asan
