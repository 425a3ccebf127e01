#!/bin/bash
# GPU visit C of round 2: rocprofv3 kernel trace + FETCH/WRITE PMC for every BASELINE config and every config-4 schedule,
# MFMA / stall counters for the two matrix-pipe kernels; summaries are written to gpurun_out/profiles_r02/ (copy to profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
P=gpurun_out/profiles_r02; rm -rf $P; mkdir -p $P
run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $P/r02_$TAG --no-raw | tail -8; }
run whole
run step --fusion step
run stage --fusion stage
run c1 --config 1
run c2 --config 2
run c3 --config 3
run c5 --config 5
# matrix-pipe counters: config 4 (k_persist_linear_mfma) and config 5 (k_persist_mlp)
for CFG in 4 5; do
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  TAG=$(echo c${CFG}_$SET | tr ' ' '_' | cut -c1-44)
  rm -rf gpurun_out/pmcx_$TAG
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/gpurun_out/pmcx_$TAG" -o r -- python "$R/bench.py" --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmcx_$TAG.log" 2>&1)
  python - "$R/gpurun_out/pmcx_$TAG" "$R/$P/r02_mfma_pmc.jsonl" $CFG <<'PY'
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
done
done
# small artefacts only
find gpurun_out -name "r_kernel_trace.csv" -size +4M -delete
find gpurun_out -name "r_counter_collection.csv" -size +4M -delete
ls -la $P
