cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD TMPDIR=/tmp LIN_ONLY=1; R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_linadj_$C
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_linadj_$C" -o r -- python $R/scripts/linear_adjoint_generic.py > "$R/gpurun_out/pmc_linadj_$C.log" 2>&1)
  python - "$R/gpurun_out/pmc_linadj_$C" $C <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if r['Counter_Name'] == sys.argv[2] and ('k_linadj' in r['Kernel_Name'] or 'k_persist_linear' in r['Kernel_Name']):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('%s %s KiB per launch: mean %.1f (n=%d, min %.1f max %.1f)' % (k, sys.argv[2], sum(v) / len(v), len(v), min(v), max(v)))
PY
done
