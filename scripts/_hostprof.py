import cProfile, pstats, os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, rhs
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
D, B = 128, 512
S = rng.standard_normal((D, D)); A = -0.5 * np.eye(D) + 0.5 * (S - S.T) / np.sqrt(D)
f = rhs.Linear.from_matrix(torch.tensor(A, device=dev))
y0 = torch.tensor(rng.standard_normal((B, D)), device=dev)
t = torch.tensor([0., 1.])
for _ in range(20): odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
torch.cuda.synchronize()
print('per call us', (time.perf_counter() - t0) / n * 1e6, odeint.last_stats.get('n_attempts'))
pr = cProfile.Profile(); pr.enable()
for _ in range(n): odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
