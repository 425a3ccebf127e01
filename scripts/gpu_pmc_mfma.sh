#!/bin/bash
# MFMA-pipe counters of the dominant kernel of bench.py (separate --pmc passes; kernel trace only).
# usage: gpu_pmc_mfma.sh [bench args...]   -> gpurun_out/pmcx_*/ and gpurun_out/mfma_pmc.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_out
rm -f gpurun_out/mfma_pmc.jsonl
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA"; do
  TAG=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmcx_$TAG
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/gpurun_out/pmcx_$TAG" -o r -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$R/gpurun_out/pmcx_$TAG.log" 2>&1)
  echo "[$SET] exit $?"
  python - "$R/gpurun_out/pmcx_$TAG" "$R/gpurun_out/mfma_pmc.jsonl" <<'PY'
import csv, glob, json, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file'); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r['Kernel_Name']
    if 'k_step_linear_mfma' in name or 'k_persist_linear_mfma' in name:
        agg[(name.split('(')[0], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(sys.argv[2], 'a') as out:
    for (kern, k), v in agg.items():
        vals = [a for a, _ in v]; durs = [d for _, d in v]
        rec = {'kernel': kern, 'counter': k, 'mean': sum(vals) / len(vals), 'launches': len(vals), 'mean_kernel_ns': sum(durs) / len(durs)}
        out.write(json.dumps(rec) + '\n')
        print('  %-40s %-28s mean %.4e  (n=%d, mean kernel ns %.0f)' % (kern[-40:], k, rec['mean'], len(vals), rec['mean_kernel_ns']))
PY
done
