import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import models, odeint
dev = torch.device('cuda:0'); torch.manual_seed(0)
tt = torch.tensor([0., 1.])
def bench(fn, n=40):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2], ts[-1]
for dt in (torch.float32, torch.float64):
    f = models.ODEFunc(64, 128, non_linearity='tanh').to(dev).to(dt)
    r = f.device_rhs()
    for batch in (512, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
        x = torch.randn(batch, 64, dtype=dt, device=dev)
        with torch.no_grad():
            a = bench(lambda: odeint(r, x, tt, rtol=1e-3, atol=1e-3, method='dopri5'))
        print(dt, batch, 'tiles', (batch + 31) // 32, 'whole min/med/max %.3f %.3f %.3f ms' % a, flush=True)
