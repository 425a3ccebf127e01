#!/bin/bash
# pass micro-benchmarks of the fused adjoint kernel + counters (separate --pmc passes; kernel trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_out
export PYTHONPATH=$R
{
MI_ODE_ADJOINT_BENCH=2,10 timeout 300 python scripts/adj_bench.py 2
MI_ODE_ADJOINT_BENCH=3,10 timeout 300 python scripts/adj_bench.py 2
} > gpurun_out/adj_bench.log 2>&1
grep bench gpurun_out/adj_bench.log
rm -f gpurun_out/adj_pmc.jsonl
for m in ${ADJ_MODES:-2 3}; do
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT"; do
    TAG=${m}_$(echo $SET | tr ' ' '_' | cut -c1-30)
    rm -rf gpurun_out/adjpmc_$TAG
    (cd /tmp && MI_ODE_ADJOINT_BENCH=$m,10 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/gpurun_out/adjpmc_$TAG" -o r -- python "$R/scripts/adj_bench.py" 1 > "$R/gpurun_out/adjpmc_$TAG.log" 2>&1)
    python - "$R/gpurun_out/adjpmc_$TAG" "$R/gpurun_out/adj_pmc.jsonl" $m <<'PY'
import csv, glob, json, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file for', sys.argv[1]); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_adjoint_mlp' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(sys.argv[2], 'a') as out:
    for k, v in agg.items():
        vals = [a for a, _ in v]; durs = [d for _, d in v]
        rec = {'mode': int(sys.argv[3]), 'counter': k, 'mean': sum(vals) / len(vals), 'launches': len(vals), 'mean_kernel_ns': sum(durs) / len(durs)}
        out.write(json.dumps(rec) + '\n')
        print('  mode %s %-28s mean %.4e  (n=%d, kernel ns %.0f)' % (sys.argv[3], k, rec['mean'], len(vals), rec['mean_kernel_ns']))
PY
  done
done
