#!/bin/bash
# fused adjoint: pass timings and one training step at config 5's shape, product library vs a variant (TFDIFFEQ_AMD_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD; V=$PWD/tfdiffeq_amd/_variants
for lib in "" $V/libmi_ode_${1:-adjrev}.so; do
  echo "== library: ${lib:-product}"
  for i in 1 2; do TFDIFFEQ_AMD_LIB=$lib python scripts/adjoint_train_step.py fused 10 2>&1 | grep ms_per | cut -c1-260; done
  TFDIFFEQ_AMD_LIB=$lib MI_ODE_ADJOINT_PROF=1 python scripts/adj_bench.py 3 2>&1 | grep -i "adjoint\|us" | tail -3 | cut -c1-300
  TFDIFFEQ_AMD_LIB=$lib MI_ODE_ADJOINT_BENCH=3,20 python scripts/adj_bench.py 3 2>&1 | grep -i "pass\|us" | tail -2 | cut -c1-300
done
