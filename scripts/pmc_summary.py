#!/usr/bin/env python3
"""Summarise rocprofv3 output of bench.py into profiles/: per-kernel durations (kernel trace) and HBM traffic per
launch from the FETCH_SIZE / WRITE_SIZE PMC passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2; WRITE_SIZE as is; both in KiB.
Usage: python scripts/pmc_summary.py gpurun_out <tag> profiles/r02_<tag> [--no-raw]"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]
RAW = '--no-raw' not in sys.argv          # copy the raw per-dispatch counter CSVs next to the summary
os.makedirs(os.path.dirname(dst) or '.', exist_ok=True)
out = {'note': 'real launches only (no-op launches after done are excluded by taking values >= 50% of the max)'}
ks = os.path.join(src, 'prof_' + tag, 'r_kernel_stats.csv')
if os.path.exists(ks):
    shutil.copy(ks, dst + '_kernel_stats.csv')
tr = os.path.join(src, 'prof_' + tag, 'r_kernel_trace.csv')
if os.path.exists(tr):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out['kernel_us'] = {}
    for k, v in d.items():
        real = [x for x in v if x >= 0.5 * max(v)]
        out['kernel_us'][k] = {'launches': len(v), 'real_launches': len(real), 'avg_real_us': sum(real) / len(real),
                               'max_us': max(v), 'min_real_us': min(real)}
pm = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = os.path.join(src, 'pmc_' + tag + '_' + C, 'r_counter_collection.csv')
    if not os.path.exists(f):
        continue
    if RAW:
        shutil.copy(f, dst + '_pmc_' + C + '.csv')
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == C:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        real = [x for x in v if x >= 0.5 * max(v)]
        pm.setdefault(k, {})[C + '_KiB_per_launch'] = sum(real) / len(real)
for k, v in pm.items():
    if 'FETCH_SIZE_KiB_per_launch' in v and 'WRITE_SIZE_KiB_per_launch' in v:
        v['hbm_bytes_per_launch'] = (2.0 * v['FETCH_SIZE_KiB_per_launch'] + v['WRITE_SIZE_KiB_per_launch']) * 1024.0
out['pmc'] = pm
json.dump(out, open(dst + '_summary.json', 'w'), indent=1, sort_keys=True)
print('wrote', dst + '_summary.json')
for k, v in sorted(out.get('kernel_us', {}).items(), key=lambda kv: -kv[1]['avg_real_us'])[:12]:
    t = pm.get(k, {}).get('hbm_bytes_per_launch')
    print('%-95s %8.1f us  traffic %s' % (k[:95], v['avg_real_us'], ('%.1f MB' % (t / 1e6)) if t else '-'))
