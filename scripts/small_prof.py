import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs
dev = torch.device('cuda:0')
rng1 = np.random.default_rng(1)
y3 = torch.tensor(np.array([1., 1., 1.]) + 1e-3 * rng1.standard_normal((65536, 3)), device=dev)
y2 = torch.tensor(np.random.default_rng(0).uniform(-2, 2, size=(4096, 2)), device=dev)
A2 = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
for fusion in sys.argv[1:] or ['step']:
    for name, f, y, t, kw in (('C3', rhs.Lorenz(), y3, [0., 10.], dict(rtol=1e-6, atol=1e-9, method='dopri5')),
                              ('C2', rhs.CubicLinear(A2), y2, [0., 25.], dict(method='dopri5'))):
        for _ in range(2):
            odeint(f, y, torch.tensor(t), options={'fusion': fusion}, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            odeint(f, y, torch.tensor(t), options={'fusion': fusion}, **kw)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        st = odeint.last_stats
        print('%s fusion=%-10s %.3f ms/call  %d attempts  %.2f us/attempt  launches %d polls %d' % (name, fusion, ms, st['n_attempts'], 1e3 * ms / st['n_attempts'], st['n_launches'], st['n_polls']))
