#!/bin/bash
# Everything round 5 runs on the MI355X box, one stage per call:   gpurun -- 'bash scripts/gpu_round5.sh <stage>'
#   tests     the whole GPU suite, smoke()
#   bench     bench.py for config 4 (headline, with the CPU leg) and 1, 2, 3, 5
#   power     scripts/power_trace.py for configs 4 and 5
#   linadj    the one-launch linear adjoint: timing script, its phase profile, rocprofv3 kernel trace
#   profiles  rocprofv3 kernel trace + PMC passes of configs 4 (whole, stage) and 5 -> gpurun_out/profiles_r05/
#   dist      the N > 1 path of bench.py with 2, 4 and 8 ranks sharing the box's one GPU (weak and strong)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out gpurun_out/profiles_r05; export PYTHONPATH=$PWD TMPDIR=/tmp; R=$PWD
P=gpurun_out/profiles_r05
case "${1:-tests}" in
tests)
  timeout 2700 python -m pytest tests -m gpu -q -rfs --tb=short > gpurun_out/r05_pytest_gpu_full.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05_pytest_gpu_full.txt | cut -c1-260 | tail -60
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
bench)
  python bench.py > $P/r05_bench_config4.json 2> gpurun_out/r05_bench_config4.err; cut -c1-600 $P/r05_bench_config4.json; tail -2 gpurun_out/r05_bench_config4.err
  rm -f $P/r05_bench_other_configs.jsonl
  for c in 1 2 3 5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee -a $P/r05_bench_other_configs.jsonl | cut -c1-240; done ;;
power)
  python scripts/power_trace.py 3.0 4 2>&1 | grep -v amdgpu.ids | tee $P/r05_power_trace_config4.txt | tail -30
  python scripts/power_trace.py 3.0 5 2>&1 | grep -v amdgpu.ids | tee $P/r05_power_trace_config5.txt | tail -12 ;;
linadj)
  (python scripts/linear_adjoint_generic.py; MI_ODE_LINADJ_PROF=1 LIN_ONLY=1 python scripts/linear_adjoint_generic.py 2>&1 | grep "linadj" | tail -12; python scripts/bench_outer.py) 2>&1 | grep -v amdgpu.ids | tee $P/r05_linear_adjoint.txt
  rm -rf gpurun_out/prof_linadj; LIN_ONLY=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_linadj -o r -- python scripts/linear_adjoint_generic.py > gpurun_out/prof_linadj.out 2>&1
  f=$(find gpurun_out/prof_linadj -name "r_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/r05_linear_adjoint_kernel_stats.csv && head -8 $f | cut -c1-220
  find gpurun_out/prof_linadj -name "r_kernel_trace.csv" -size +4M -delete ;;
profiles)
  run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $P/r05_$TAG --no-raw | tail -6; }
  run whole; run stage --fusion stage; run c5 --config 5
  find gpurun_out -name "r_kernel_trace.csv" -size +4M -delete; find gpurun_out -name "r_counter_collection.csv" -size +4M -delete
  ls -la $P ;;
dist)
  bash scripts/gpu_dist_check.sh 2>&1 | tail -60 | tee $P/r05_bench_dist_shared_gpu.txt ;;
*) echo "unknown stage $1"; exit 2 ;;
esac
