#!/bin/bash
# Everything round 4 runs on the MI355X box, one stage per call:   gpurun -- 'bash scripts/gpu_round4.sh <stage>'
#   first     new callable-engine tests, bench.py (product + -DMI_LIN_FMA / -DMI_PERSIST_PROF variants), scripts/bench_callable.py,
#             DETEST on the Python-callable path (device controller vs the host controller of rounds 1-3)
#   tests     the whole GPU suite (-rs: skip reasons), smoke()
#   bench     bench.py for configs 4 (headline, with the CPU leg) and 1, 2, 3, 5
#   profiles  rocprofv3 kernel trace + FETCH/WRITE PMC passes of every BASELINE config -> gpurun_out/profiles_r04/
#   callable  scripts/bench_callable.py + DETEST on the callable path
#   adjoint   rocprofv3 of training steps on the fused adjoint, models.LinearODEFunc vs the autograd path, mi_ode_outer_reduce alone
#   soaks     randomised parity soaks (whole vs step, multistep, tuple states)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD TMPDIR=/tmp; R=$PWD
V=$PWD/tfdiffeq_amd/_variants
case "${1:-tests}" in
first)
  timeout 900 python -m pytest tests/test_gpu_callable_engine.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r04_callable_tests.txt
  python bench.py --no-cpu-baseline > gpurun_out/r04_bench_product.json 2> gpurun_out/r04_bench_product.err; cut -c1-300 gpurun_out/r04_bench_product.json
  for v in fma prof fmaprof; do
    if [ -f $V/libmi_ode_$v.so ]; then
      TFDIFFEQ_AMD_LIB=$V/libmi_ode_$v.so python bench.py --no-cpu-baseline > gpurun_out/r04_bench_$v.json 2> gpurun_out/r04_bench_$v.err
      echo "== variant $v"; cut -c1-200 gpurun_out/r04_bench_$v.json; grep "persist" gpurun_out/r04_bench_$v.err | tail -9
    fi
  done
  timeout 900 python scripts/bench_callable.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_callable.txt
  for g in auto host; do
    echo "== DETEST, Python callables, graph=$g"
    timeout 600 python scripts/detest_run.py --methods dopri5 --tols 1e-6 --graph-mode $g 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_detest_callable_$g.txt | tail -28
  done ;;
second)
  timeout 2400 python -m pytest tests -m gpu -q -rs 2>&1 | tail -60 | tee gpurun_out/r04_pytest_gpu.txt
  python bench.py > gpurun_out/r04_bench_config4.json 2> gpurun_out/r04_bench_config4.err; cut -c1-1500 gpurun_out/r04_bench_config4.json; tail -3 gpurun_out/r04_bench_config4.err
  TFDIFFEQ_AMD_LIB=$V/libmi_ode_prof.so python bench.py --no-cpu-baseline > gpurun_out/r04_bench_prof.json 2> gpurun_out/r04_bench_prof.err
  echo "== variant prof"; cut -c1-200 gpurun_out/r04_bench_prof.json; grep "persist" gpurun_out/r04_bench_prof.err | tail -9 ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q -rfs --tb=short > gpurun_out/r04_pytest_gpu_full.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04_pytest_gpu_full.txt | cut -c1-260 | tail -60
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
bench)
  python bench.py > gpurun_out/r04_bench_config4.json 2> gpurun_out/r04_bench_config4.err; cut -c1-400 gpurun_out/r04_bench_config4.json
  for c in 1 2 3 5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee -a gpurun_out/r04_bench_other_configs.jsonl | cut -c1-240; done ;;
profiles)
  P=gpurun_out/profiles_r04; rm -rf $P; mkdir -p $P
  run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $P/r04_$TAG --no-raw | tail -6; }
  run whole; run stage --fusion stage; run c5 --config 5
  find gpurun_out -name "r_kernel_trace.csv" -size +4M -delete; find gpurun_out -name "r_counter_collection.csv" -size +4M -delete
  ls -la $P ;;
callable)
  timeout 900 python scripts/bench_callable.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_callable.txt ;;
adjoint)
  # fused adjoint (rocprofv3 of training steps, both networks), the linear system under odeint_adjoint, the outer-product kernel alone
  bash scripts/gpu_adjoint_profile.sh 2>&1 | tail -6
  mkdir -p gpurun_out/profiles_r04
  (python scripts/linear_adjoint_generic.py; python scripts/bench_outer.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/profiles_r04/r04_linear_adjoint.txt ;;
soaks)
  (echo "== soak_whole_vs_step.py 11 100"; timeout 100 python scripts/soak_whole_vs_step.py 11 100 2>&1 | tail -1
   echo "== soak_multistep.py 11 40"; timeout 200 python scripts/soak_multistep.py 11 40 2>&1 | tail -1
   echo "== soak_tuple.py 100 11"; timeout 300 python scripts/soak_tuple.py 100 11 2>&1 | tail -1) | grep -v amdgpu.ids | tee gpurun_out/r04_soaks_final.txt ;;
*) echo "unknown stage $1"; exit 2 ;;
esac
