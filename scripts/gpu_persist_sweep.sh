#!/bin/bash
# back-off sweep of the whole-integration kernel's hand-off (library built with -DMI_PERSIST_PROF)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TFDIFFEQ_AMD_LIB=$PWD/tfdiffeq_amd/libmi_ode_prof.so
for s0 in ${SWEEP0:-24 32 48 64 96}; do for s1 in ${SWEEP1:-2}; do
  echo "== sleep_first $s0 sleep_poll $s1"
  MI_ODE_PERSIST_SLEEP0=$s0 MI_ODE_PERSIST_SLEEP1=$s1 timeout 100 python scripts/persist_prof.py 2>&1 | grep -v amdgpu.ids | awk 'NR%3==0' | cut -c1-140
done; done
