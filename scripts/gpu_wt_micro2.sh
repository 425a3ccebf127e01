#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
O=gpurun_out/wt_micro2; rm -rf $O; mkdir -p $O
for B in wavetile_bench wavetile_benchDMI_WT_LOOK16 wavetile_benchDMI_WT_FILL4; do
  echo "=== $B ===" | tee -a $O/run.log
  timeout 300 $R/scripts/micro/$B >> $O/run.log 2>&1; echo "exit $?" >> $O/run.log
done
cat $O/run.log | grep -v records | grep -v "new {"
B=$R/scripts/micro/wavetile_bench
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  TAG=$(echo $SET | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$O/pmc_$TAG" -o r -- $B > "$R/$O/pmc_$TAG.log" 2>&1)
  echo "[$SET] exit $?"
  python - "$R/$O/pmc_$TAG" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('  no counter file'); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    name = r['Kernel_Name']
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if 'k_step_linear' in name and dur > 150000:
        bucket = 'out' if dur > 400000 else 'noout'
        agg[(name.split('(')[0][10:40], r['Counter_Name'], bucket if 'wt' in name else '')].append((float(r['Counter_Value']), dur))
for (kern, k, b), v in sorted(agg.items()):
    vals = [a for a, _ in v]; durs = [d for _, d in v]
    print('  %-32s %-6s %-22s mean %.4e  (n=%d, mean kernel ns %.0f)' % (kern, b, k, sum(vals) / len(vals), len(vals), sum(durs) / len(durs)))
PY
  find $O/pmc_$TAG -name "*.csv" -size +8M -delete
done
