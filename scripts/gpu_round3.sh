#!/bin/bash
# Everything round 3 runs on the MI355X box, one stage per call:   gpurun -- 'bash scripts/gpu_round3.sh <stage>'
#   tests     the whole GPU suite (-rs: skip reasons), smoke()
#   bench     bench.py for configs 4 (headline, with the CPU leg) and 1, 2, 3, 5
#   profiles  rocprofv3 kernel trace + FETCH/WRITE PMC passes of every BASELINE config and config-4 schedule, MFMA / stall
#             counters of the two matrix-pipe kernels -> gpurun_out/profiles_r03/ (copy the summaries to profiles/)
#   timeline  in-kernel timeline of the config-4 attempt pass (needs tfdiffeq_amd/_variants/libmi_ode_trace.so: make EXTRA=-DMI_TRACE)
#   micro     scripts/micro/mfma_pair + mfma_overlap (what the two wavefronts of a SIMD overlap)
#   detest    scripts/detest_fused.py (DETEST on the one-launch kernels, CPU restatement beside)
#   bands     scripts/measure_fp32_bands.sh (re-records tests/golden/fp32_bands.json)
#   asan      the C-ABI smoke test against the ASan/UBSan host build (scripts/build_asan.sh first)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONPATH=$PWD TMPDIR=/tmp; R=$PWD
case "${1:-tests}" in
tests)
  timeout 2400 python -m pytest tests -m gpu -q -x -rs 2>&1 | tail -30 | tee gpurun_out/r03_pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
bench)
  python bench.py > gpurun_out/r03_bench_config4.json 2> gpurun_out/r03_bench_config4.err; cut -c1-400 gpurun_out/r03_bench_config4.json
  for c in 1 2 3 5; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tee -a gpurun_out/r03_bench_other_configs.jsonl | cut -c1-240; done ;;
profiles)
  P=gpurun_out/profiles_r03; rm -rf $P; mkdir -p $P
  run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $P/r03_$TAG --no-raw | tail -6; }
  run whole; run step --fusion step; run stage --fusion stage; run c1 --config 1; run c2 --config 2; run c3 --config 3; run c5 --config 5
  for CFG in 4 5; do
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    TAG=$(echo c${CFG}_$SET | tr ' ' '_' | cut -c1-44); rm -rf gpurun_out/pmcx_$TAG
    (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/gpurun_out/pmcx_$TAG" -o r -- python "$R/bench.py" --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmcx_$TAG.log" 2>&1)
    python - "$R/gpurun_out/pmcx_$TAG" "$R/$P/r03_mfma_pmc.jsonl" $CFG <<'PY'
import csv, glob, json, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file'); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r['Kernel_Name']
    if 'k_persist_linear_mfma' in name or 'k_persist_mlp' in name:
        agg[(name.split('(')[0], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(sys.argv[2], 'a') as out:
    for (kern, k), v in agg.items():
        v = [x for x in v if x[1] >= 0.5 * max(d for _, d in v)]
        vals = [a for a, _ in v]; durs = [d for _, d in v]
        rec = {'config': int(sys.argv[3]), 'kernel': kern, 'counter': k, 'mean': sum(vals) / len(vals), 'launches': len(vals), 'mean_kernel_ns': sum(durs) / len(durs)}
        out.write(json.dumps(rec) + '\n')
        print('  %-44s %-28s mean %.4e  (n=%d, mean kernel ns %.0f)' % (kern[-44:], k, rec['mean'], len(vals), rec['mean_kernel_ns']))
PY
    find gpurun_out/pmcx_$TAG -name "*.csv" -size +4M -delete
  done; done
  find gpurun_out -name "r_kernel_trace.csv" -size +4M -delete; find gpurun_out -name "r_counter_collection.csv" -size +4M -delete
  ls -la $P ;;
timeline)
  TFDIFFEQ_AMD_LIB=$PWD/tfdiffeq_amd/_variants/libmi_ode_trace.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "\[trace\]" | cut -c1-140 | tee gpurun_out/r03_config4_timeline.txt | head -30 ;;
micro)
  ./scripts/micro/mfma_pair | tee gpurun_out/r03_mfma_pair.txt; ./scripts/micro/mfma_overlap | tee gpurun_out/r03_mfma_overlap.txt ;;
detest)
  timeout 1400 python scripts/detest_fused.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_detest_fused_table.txt | tail -30 ;;
bands)
  bash scripts/measure_fp32_bands.sh ;;
asan)
  bash scripts/gpu_asan.sh ;;
*) echo "unknown stage $1"; exit 2 ;;
esac
