#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/prof_small
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_small" -o s -- python "$R/scripts/small_prof.py" ${1:-step} > "$R/gpurun_out/prof_small.log" 2>&1)
echo "rocprof exit $?"; tail -5 gpurun_out/prof_small.log
find gpurun_out/prof_small -type f | head
for f in $(find gpurun_out/prof_small -name "*kernel_stats.csv" | head -1); do cut -c1-220 "$f" | head -8; done
python - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/prof_small/**/*kernel_trace.csv', recursive=True)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # gaps between consecutive step kernels
    ks = [r for r in rows if 'k_step_rowlocal' in r['Kernel_Name']]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ks]
    g = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows[:-1], rows[1:])]
    import statistics as st
    print('step kernels: n=%d median dur %.2f us p90 %.2f' % (len(d), st.median(d), sorted(d)[int(0.9 * len(d))]))
    print('gaps between consecutive kernels: median %.2f us p90 %.2f max %.1f' % (st.median(g), sorted(g)[int(0.9 * len(g))], max(g)))
PY
find gpurun_out/prof_small -name "*kernel_trace.csv" -delete
