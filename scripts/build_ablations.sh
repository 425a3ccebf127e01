#!/bin/bash
# Variants of libmi_ode.so with parts of the linear tile kernels compiled out (MI_ABL bit mask, csrc/mi_ode_step_fused.h) and the
# in-kernel clock on (MI_PERSIST_PROF): what the whole-call kernel's attempt pass is made of.  Only the fp64 TU + api are rebuilt
# per variant; the numbers are "us per attempt pass" from the [persist prof] line (the step sequence changes with the arithmetic,
# the per-pass time does not depend on it).   usage: scripts/build_ablations.sh "0 1 2 4 8 3 ..."
set -e
cd "$(dirname "$0")/../tfdiffeq_amd/csrc"
mkdir -p ../_variants
for m in ${1:-0 1 2 4 8}; do
  make clean > /dev/null
  make -j8 EXTRA="-DMI_ABL=$m -DMI_PERSIST_PROF" OUT=../_variants/libmi_ode_abl$m.so > /dev/null 2>&1
  echo "built abl$m"
done
make clean > /dev/null
make -j8 > /dev/null 2>&1
echo "rebuilt the product library"
