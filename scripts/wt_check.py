"""The experimental wave-tile layout of the whole-call MLP kernel (csrc/mi_ode_mlp_wt.h, TFDIFFEQ_AMD_MLP_LAYOUT=wave) against the
default kernel: same attempt sequence, values to float32 rounding, and the time per odeint call at config 5's shape."""
import os
import time

import torch

from tfdiffeq_amd import odeint, rhs


def make(layout, act, td, g):
    os.environ['TFDIFFEQ_AMD_MLP_LAYOUT'] = layout

    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return ((torch.rand(i, o, generator=g) * 2 - 1) * lim).cuda()
    W1 = glorot(64 + (1 if td else 0), 128)
    Ws = [W1, glorot(128, 128), glorot(128, 64)]
    bs = [(0.1 * torch.randn(n, generator=g)).cuda() for n in (128, 128, 64)]
    return Ws, bs


for act, td, method, tol, batch in (('tanh', False, 'dopri5', 1e-3, 32768), ('relu', False, 'dopri5', 1e-3, 32768), ('tanh', True, 'tsit5', 1e-5, 1000),
                                    ('softplus', False, 'dopri5', 1e-6, 4097)):
    g = torch.Generator().manual_seed(4)
    Ws, bs = make('workgroup', act, td, g)
    y0 = torch.randn(batch, 64, generator=torch.Generator().manual_seed(5)).cuda()
    t = torch.tensor([0., 0.4, 1.0])
    res = {}
    for layout in ('workgroup', 'wave'):
        os.environ['TFDIFFEQ_AMD_MLP_LAYOUT'] = layout
        f = rhs.MLP(Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], activation=act, time_dependent=td)
        sol = odeint(f, y0, t, rtol=tol, atol=tol, method=method)
        st = dict(odeint.last_stats)
        for _ in range(5):
            odeint(f, y0, t, rtol=tol, atol=tol, method=method)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            odeint(f, y0, t, rtol=tol, atol=tol, method=method)
        torch.cuda.synchronize()
        res[layout] = (sol, st, (time.perf_counter() - t0) / 20 * 1e3)
    a, b = res['workgroup'], res['wave']
    rel = float((a[0] - b[0]).abs().max() / a[0].abs().max())
    print('%-8s td %d %s tol %.0e batch %5d: attempts %d / %d  launches %s / %s  status %d / %d  rel diff %.2e   %.3f ms -> %.3f ms per call' % (
        act, td, method, tol, batch, a[1]['n_attempts'], b[1]['n_attempts'], a[1].get('n_launches'), b[1].get('n_launches'), a[1]['status'], b[1]['status'],
        rel, a[2], b[2]), flush=True)
