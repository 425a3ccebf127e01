"""One-launch linear adjoint (mi_ode_linadj) against the callable-engine path (round 4) and, at config 4's size, its timing."""
import sys
import time

import torch

from tfdiffeq_amd import adjoint as ADJ
from tfdiffeq_amd import models, odeint_adjoint

dev = torch.device('cuda:0')


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


def grads(func, y0, t, w, one_launch, **kw):
    ADJ.LINEAR_ONE_LAUNCH = one_launch
    for p in func.parameters():
        p.grad = None
    yi = y0.clone().requires_grad_(True)
    tt = t.clone().requires_grad_(True)
    sol = odeint_adjoint(func, yi, tt, **kw)
    (sol * w).sum().backward()
    return sol.detach(), yi.grad.clone(), [p.grad.clone() for p in func.parameters()], tt.grad.clone(), dict(odeint_adjoint.last_backward_stats)


def case(batch, dim, bias, dtype, tpts, tol):
    torch.manual_seed(dim)
    func = models.LinearODEFunc(dim, bias=bias, dtype=dtype).to(dev)
    if bias:
        with torch.no_grad():
            func.bias.normal_(0.0, 0.1)
    g = torch.Generator().manual_seed(batch)
    y0 = torch.randn(batch, dim, generator=g, dtype=dtype).to(dev)
    t = torch.tensor(tpts, dtype=torch.float64)
    w = torch.randn(len(tpts), batch, dim, generator=g, dtype=dtype).to(dev)
    a = grads(func, y0, t, w, True, **tol)
    b = grads(func, y0, t, w, False, **tol)
    segs = a[4].get('segments', [])
    att_a = [s_['n_attempts'] for s_ in segs]
    print('batch %6d dim %3d bias %d %s  engine: %s' % (batch, dim, bias, str(dtype)[6:], a[4]['engine'][:60]))
    print('   attempts per interval (one launch) %s | callable engine last interval %s' % (att_a, b[4].get('last_segment', {}).get('n_attempts')))
    print('   dL/dy0 rel %.2e  dL/dt abs %.2e  params rel %s' % (rel(a[1], b[1]), float((a[3] - b[3]).abs().max()), ['%.2e' % rel(x, y) for x, y in zip(a[2], b[2])]))
    return a, b


if __name__ == '__main__':
    tol64 = dict(rtol=1e-7, atol=1e-9, method='dopri5')
    case(64, 8, True, torch.float64, [0.0, 1.0], tol64)
    case(64, 8, True, torch.float64, [0.0, 0.4, 1.0], tol64)
    case(1000, 33, False, torch.float64, [0.0, 0.4, 1.0], tol64)
    case(5000, 128, True, torch.float64, [0.0, 0.4, 1.0], tol64)
    case(300, 16, True, torch.float32, [0.0, 0.4, 1.0], dict(rtol=1e-4, atol=1e-5, method='dopri5'))
    case(70001, 128, True, torch.float64, [0.0, 1.0], dict(rtol=1e-6, atol=1e-9, method='dopri5'))
    case(70001, 128, False, torch.float64, [0.0, 1.0], dict(rtol=1e-6, atol=1e-9, method='dopri5'))
    case(3000, 64, True, torch.float64, [0.0, 1.0], dict(rtol=1e-6, atol=1e-9, method='dopri5'))
    case(4100, 100, True, torch.float64, [0.0, 1.0], dict(rtol=1e-6, atol=1e-9, method='dopri5'))
    if '--time' in sys.argv:
        torch.manual_seed(0)
        func = models.LinearODEFunc(128, bias=False).to(dev)
        y0 = torch.randn(65536, 128, dtype=torch.float64, device=dev)
        t = torch.tensor([0.0, 1.0], dtype=torch.float64)
        for one in (True, False):
            ADJ.LINEAR_ONE_LAUNCH = one
            for it in range(4):
                for p in func.parameters():
                    p.grad = None
                yi = y0.clone().requires_grad_(True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sol = odeint_adjoint(func, yi, t, rtol=1e-6, atol=1e-9, method='dopri5')
                loss = sol[-1].pow(2).sum()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                loss.backward()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                st = odeint_adjoint.last_backward_stats
                print('one_launch=%s call %d: forward %.2f ms backward %.2f ms  [%s] %s' % (one, it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), st['engine'][:70],
                      st.get('last_segment', {}).get('n_attempts')))
            if one:
                eng = list(ADJ._LIN_ENGINES.values())[-1]
                print('   profile of the last segment:', eng.profile(), 'clock', eng.stats.clock_mhz)
