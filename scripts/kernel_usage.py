#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel: python kernel_usage.py mi_ode_launch_f64.hip [filter]"""
import os, re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + os.environ.get("EXTRA", "").split() + ["-c", src,
       '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    txt = m.group(1).strip()
    if txt.startswith('Function Name:'):
        cur = {'name': txt.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in txt:
        k, v = txt.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    print('%-100s VGPR %-4s SGPR %-4s spill %s/%s scratch %s occ %s' % (name[:100], r.get('VGPRs'), r.get('TotalSGPRs'),
          r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]')))
