"""Where the host side of one backward pass of the linear adjoint goes (65536 x 128, float64): every step of adjoint._linear_backward timed
with a synchronisation after it (an upper bound per step), against the whole backward call."""
import time
import torch
from tfdiffeq_amd import adjoint as ADJ
from tfdiffeq_amd import models, odeint_adjoint

dev = torch.device('cuda:0')
B, D = 65536, 128
torch.manual_seed(0)
func = models.LinearODEFunc(D, bias=False).to(dev)
y0 = torch.randn(B, D, dtype=torch.float64, device=dev)
t = torch.tensor([0., 1.], dtype=torch.float64)


def sync_time(fn, n=5):
    out = None
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, out


for it in range(4):
    func.weight.grad = None
    yi = y0.clone().requires_grad_(True)
    out = odeint_adjoint(func, yi, t, rtol=1e-6, atol=1e-9, method='dopri5')
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out[-1].sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
print('whole backward call: %.0f us' % ((t2 - t1) * 1e6))
eng = list(ADJ._LIN_ENGINES.values())[-1]
g = torch.zeros(2, B, D, dtype=torch.float64, device=dev); g[-1] = 1.0
ans = out.detach()
W = func.weight.detach()
us, _ = sync_time(lambda: torch.zeros(2, B, D, dtype=torch.float64, device=dev)); print('zeros [2,B,D] (autograd of out[-1]): %.0f us' % us)
us, _ = sync_time(lambda: g[-1].reshape(B, D).contiguous()); print('g[-1].contiguous(): %.0f us' % us)
from tfdiffeq_amd.fixed_grid import Euler
from tfdiffeq_amd.solvers import _FusedEngine, _cached_engine, _tableau_key
rhs_y = func.device_rhs()
proto = ans[0]
key = ('rhs evaluation', rhs_y.cache_key(proto.dtype, proto.device), (B, D), proto.dtype, str(proto.device), _tableau_key(Euler._fused_tableau, None))
ev = _cached_engine(key, lambda: _FusedEngine(rhs_y, proto, False, Euler._fused_tableau))
us, f = sync_time(lambda: ev.eval_rhs(ans[1])); print('eval_rhs: %.0f us' % us)
us, d = sync_time(lambda: torch.dot(f.reshape(-1), g[1].reshape(-1))); print('torch.dot: %.0f us' % us)
adj_t = torch.zeros((), dtype=torch.float64, device=dev)
theta = torch.zeros(D * D, dtype=torch.float64, device=dev)
a_in = g[-1].contiguous()
us, res = sync_time(lambda: eng.segment(W, None, ans[1], a_in, adj_t, theta, 1.0, 0.0)); print('eng.segment (allocations + launch + wait): %.0f us' % us)
us, _ = sync_time(lambda: res[0] + g[0]); print('adj_y + g[0]: %.0f us' % us)
us, _ = sync_time(lambda: torch.cat([adj_t.reshape(1), adj_t.reshape(1)]).to(dtype=t.dtype, device=t.device)); print('time_vjps cat + to(cpu): %.0f us' % us)
us, _ = sync_time(lambda: (adj_t - d, d.reshape(1))); print('adj_time - dLd: %.0f us' % us)
import ctypes as C
st = eng.stats
print('kernel clock %.0f MHz' % st.clock_mhz, eng.profile())
