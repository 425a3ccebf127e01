#!/bin/bash
# Everything round 5 runs on the MI355X box, one stage per call:   gpurun -- 'bash scripts/gpu_round5.sh <stage>'
#   tests     the whole GPU suite, smoke()
#   bench     bench.py for config 4 (headline, with the CPU leg) and 1, 2, 3, 5
#   power     scripts/power_trace.py for configs 4 and 5
#   linadj    the one-launch linear adjoint: timing script, its phase profile, rocprofv3 kernel trace
#   profiles  rocprofv3 kernel trace + PMC passes of configs 4 (whole, stage) and 5 -> gpurun_out/profiles_r05/
#   callable  scripts/bench_callable.py (microseconds per attempt of the Python-callable engine by schedule)
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
  rm -rf gpurun_out/prof_linadj; (cd /tmp && LIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_linadj -o r -- python $R/scripts/linear_adjoint_generic.py > $R/gpurun_out/prof_linadj.out 2>&1)
  f=$(find gpurun_out/prof_linadj -name "r_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/r05_linear_adjoint_kernel_stats.csv && head -8 $f | cut -c1-220
  find gpurun_out/prof_linadj -name "r_kernel_trace.csv" -size +4M -delete ;;
profiles)
  run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $P/r05_$TAG --no-raw | tail -6; }
  run whole; run stage --fusion stage; run c5 --config 5
  rm -f $P/r05_sq_pmc.jsonl
  for CFG in 4 5 linadj; do
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    TAG=$(echo c${CFG}_$SET | tr ' ' '_' | cut -c1-44); rm -rf gpurun_out/pmcx_$TAG
    if [ $CFG = linadj ]; then CMD="python $R/scripts/linear_adjoint_generic.py"; export LIN_ONLY=1; else CMD="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline"; unset LIN_ONLY; fi
    (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/gpurun_out/pmcx_$TAG" -o r -- $CMD > "$R/gpurun_out/pmcx_$TAG.log" 2>&1)
    python - "$R/gpurun_out/pmcx_$TAG" "$R/$P/r05_sq_pmc.jsonl" $CFG <<'PY'
import csv, glob, json, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file'); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r['Kernel_Name']
    if 'k_persist_linear_mfma' in name or 'k_persist_mlp' in name or 'k_linadj' in name:
        agg[(name.split('(')[0], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(sys.argv[2], 'a') as out:
    for (kern, k), v in agg.items():
        v = [x for x in v if x[1] >= 0.5 * max(d for _, d in v)]
        vals = [a for a, _ in v]; durs = [d for _, d in v]
        rec = {'workload': sys.argv[3], 'kernel': kern, 'counter': k, 'mean': sum(vals) / len(vals), 'launches': len(vals), 'mean_kernel_ns': sum(durs) / len(durs)}
        out.write(json.dumps(rec) + '\n')
        print('  %-44s %-28s mean %.4e  (n=%d, mean kernel ns %.0f)' % (kern[-44:], k, rec['mean'], len(vals), rec['mean_kernel_ns']))
PY
    find gpurun_out/pmcx_$TAG -name "*.csv" -size +4M -delete
  done; done; unset LIN_ONLY
  find gpurun_out -name "r_kernel_trace.csv" -size +4M -delete; find gpurun_out -name "r_counter_collection.csv" -size +4M -delete
  ls -la $P ;;
callable)
  timeout 900 python scripts/bench_callable.py 2>&1 | grep -v amdgpu.ids | tee $P/r05_callable.txt ;;
dist)
  bash scripts/gpu_dist_check.sh 2>&1 | tail -60 | tee $P/r05_bench_dist_shared_gpu.txt ;;
*) echo "unknown stage $1"; exit 2 ;;
esac
