#!/usr/bin/env python3
"""Power / clock / temperature trace of the headline kernel under sustained load (round-4 review, item 2a): the config-4 odeint call in a
loop for a few seconds, the amdgpu hwmon files (what rocm-smi reads) sampled every 10 ms, the kernel's own clock (its cycle counter against
the 100 MHz constant clock) per call beside them.  Answers: is a 2150-2250 MHz grant a power cap tripped by the kernel itself?
Usage: python scripts/power_trace.py [seconds] [config]   ->   a table on stdout (keep it under profiles/)."""
import sys
import threading
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
config = int(sys.argv[2]) if len(sys.argv) > 2 else 4


class A(object):
    batch, scaling, linear_variant, fusion = bench.BATCH, 'weak', 0, 'auto'


dev = torch.device('cuda:0')
from tfdiffeq_amd import odeint  # noqa: E402
f, y0, t, kw, desc = bench.workload(config, A, 0, 1, dev)
opts = dict(kw.pop('options', None) or {})
opts.update({'fusion': 'auto'})
hw = bench.Hwmon()
print('# %s' % desc)
print('# hwmon cards: %s' % hw.dirs)
print('# idle reading before the loop: %s' % hw.read())
rows, calls, stop = [], [], threading.Event()
t0 = time.perf_counter()


def sampler():
    while not stop.is_set():
        rows.append((time.perf_counter() - t0, hw.read()))
        stop.wait(0.01)


th = threading.Thread(target=sampler, daemon=True)
th.start()
while time.perf_counter() - t0 < seconds:
    c0 = time.perf_counter()
    odeint(f, y0, t, options=opts, **kw)
    st = dict(odeint.last_stats)
    calls.append((c0 - t0, time.perf_counter() - c0, st.get('clock_mhz', 0.0)))
torch.cuda.synchronize()
stop.set()
th.join()
time.sleep(0.3)
print('# idle reading 0.3 s after the loop: %s' % hw.read())
print('# %d calls in %.2f s; wall per call: mean %.3f ms, min %.3f, max %.3f; kernel clock: mean %.0f MHz, min %.0f, max %.0f' % (
    len(calls), seconds, 1e3 * sum(c[1] for c in calls) / len(calls), 1e3 * min(c[1] for c in calls), 1e3 * max(c[1] for c in calls),
    sum(c[2] for c in calls) / len(calls), min(c[2] for c in calls), max(c[2] for c in calls)))
print('# t_s   power_W  cap_W  sclk_MHz  mclk_MHz  T_junction_C  T_memory_C  | kernel clock (mean of the calls in the last 100 ms) MHz, ms per call')
step = max(1, len(rows) // 60)
for i in range(0, len(rows), step):
    ts, r = rows[i]
    r0 = r[0] if r else {}
    near = [c for c in calls if ts - 0.1 <= c[0] <= ts]
    kc = sum(c[2] for c in near) / len(near) if near else float('nan')
    ms = 1e3 * sum(c[1] for c in near) / len(near) if near else float('nan')
    print('%6.2f  %7.1f  %5.0f  %8.0f  %8.0f  %12.1f  %10.1f  | %8.0f  %6.3f' % (ts, r0.get('power_w') or -1, r0.get('power_cap_w') or -1, r0.get('sclk_mhz') or -1,
                                                                            r0.get('mclk_mhz') or -1, r0.get('temp_junction_c') or -1, r0.get('temp_memory_c') or -1, kc, ms))
