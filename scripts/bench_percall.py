"""Per-call wall times of bench.py's step() for config 4 (diagnostic for the timed region: which calls are slow?)."""
import gc
import sys
import time
import torch
sys.path.insert(0, '.')
import bench


class A(object):
    batch, scaling, linear_variant, fusion = bench.BATCH, 'weak', 0, 'auto'


dev = torch.device('cuda:0')
from tfdiffeq_amd import odeint
f, y0, t, kw, desc = bench.workload(4, A, 0, 1, dev)
opts = dict(kw.pop('options', None) or {})
opts.update({'profile': True, 'fusion': 'auto', 'linear_variant': 0})


def step():
    out = odeint(f, y0, t, options=opts, **kw)
    return out, dict(odeint.last_stats)


step(); torch.cuda.synchronize()
gc.collect(); gc.disable()
for _ in range(30):
    step()
torch.cuda.synchronize()
for _ in range(2):
    step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(40):
    a = time.perf_counter()
    out, st = step()
    ts.append((time.perf_counter() - a, st.get('clock_mhz', 0), st['profile'][0]))
torch.cuda.synchronize()
print('total per call %.4f ms' % (1e3 * (time.perf_counter() - t0) / 40))
print(' '.join('%.3f' % (1e3 * x[0]) for x in ts))
print(' '.join('%.0f' % x[1] for x in ts))
print(' '.join('%.3f' % x[2] for x in ts))
