#!/bin/bash
# Wave-tile attempt kernel vs the workgroup-tile kernel: bit-identity + timing + kernel trace + MFMA / stall / i-cache counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
O=gpurun_out/wt_micro; rm -rf $O; mkdir -p $O
B=$R/scripts/micro/wavetile_bench
timeout 300 $B > $O/run.log 2>&1; echo "bench exit $?" >> $O/run.log; cat $O/run.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/trace" -o t -- $B > "$R/$O/trace.log" 2>&1); echo "trace exit $?"
for f in $(find $O/trace -name "*kernel_stats.csv" | head -1); do cat "$f" | cut -c1-260; done
find $O/trace -name "*kernel_trace.csv" -size +8M -delete
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_INSTS_VALU\b\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_[A-Z]*" | sort -u > $O/counters_avail.txt; cat $O/counters_avail.txt | tr '\n' ' '; echo
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU"; do
  TAG=$(echo $SET | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$O/pmc_$TAG" -o r -- $B > "$R/$O/pmc_$TAG.log" 2>&1)
  echo "[$SET] exit $?"
  python - "$R/$O/pmc_$TAG" "$R/$O/pmc.jsonl" <<'PY'
import csv, glob, json, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file'); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r['Kernel_Name']
    if 'k_step_linear' in name and int(r.get('Grid_Size', r.get('Grid_Size_X', '0')) or 0) >= 256 * 64:
        agg[(name.split('(')[0][:48], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(sys.argv[2], 'a') as out:
    for (kern, k), v in sorted(agg.items()):
        v = v[len(v) // 2:]                      # the timed launches (config 4), not the small correctness cases
        vals = [a for a, _ in v]; durs = [d for _, d in v]
        rec = {'kernel': kern, 'counter': k, 'mean': sum(vals) / len(vals), 'launches': len(vals), 'mean_kernel_ns': sum(durs) / len(durs)}
        out.write(json.dumps(rec) + '\n')
        print('  %-48s %-28s mean %.4e  (n=%d, mean kernel ns %.0f)' % (kern, k, rec['mean'], len(vals), rec['mean_kernel_ns']))
PY
  find $O/pmc_$TAG -name "*.csv" -size +8M -delete
done
