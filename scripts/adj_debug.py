"""First-contact check of the fused adjoint kernel (GPU): dynamics vs autograd, one segment vs the plane-kernel engine."""
import sys
import time

import torch

import tfdiffeq_amd as T
from tfdiffeq_amd import adjoint as ADJ
from tfdiffeq_amd.models import ODEFunc

torch.manual_seed(0)
dev = 'cuda'


def canon(func):
    return torch.cat([func.fc1.weight.t().reshape(-1), func.fc1.bias, func.fc2.weight.t().reshape(-1), func.fc2.bias,
                      func.fc3.weight.t().reshape(-1), func.fc3.bias]).detach()


def run(batch, dim, hidden, tol=1e-3, time_it=False):
    func = ODEFunc(dim, hidden, non_linearity='tanh').to(dev)
    y = torch.randn(batch, dim, device=dev)
    a = torch.randn(batch, dim, device=dev) / batch
    mlp = func.device_rhs()
    eng = ADJ._FusedAdjointEngine(batch, dim, hidden, tol, tol, 0.9, 10.0, 0.2, 1000, dev)
    f, vy, vp = eng.dynamics(mlp, y, a)
    yr = y.clone().requires_grad_(True)
    fr = func(torch.tensor(0.), yr)
    g = torch.autograd.grad(fr, (yr,) + tuple(func.parameters()), -a)
    ref_p = torch.cat([g[1].t().reshape(-1), g[2], g[3].t().reshape(-1), g[4], g[5].t().reshape(-1), g[6]])
    sc = lambda x: x.abs().max().item()
    print('[dyn %dx%dx%d] f %.2e  vjp_y %.2e (scale %.2e)  vjp_p %.2e (scale %.2e)' % (
        batch, dim, hidden, sc(f - fr.detach()), sc(vy - g[0]), sc(g[0]), sc(vp - ref_p), sc(ref_p)))
    # one backward segment 1 -> 0 against the plane-kernel engine
    theta0 = torch.randn(eng.n_params, device=dev) * 0.01
    adj_t = torch.tensor(0.3, device=dev)
    a_out, t_out, p_out = eng.segment(mlp, y, a, adj_t, theta0, 1.0, 0.0)
    st = eng.stats.as_dict()
    fp = tuple(func.parameters())

    def aug(tt, ya):
        yy, aa = ya[0], ya[1]
        with torch.enable_grad():
            y_ = yy.detach().requires_grad_(True)
            fe = func(tt, y_)
            vj = torch.autograd.grad(fe, (y_,) + fp, -aa)
        vp_ = torch.cat([vj[1].t().reshape(-1), vj[2], vj[3].t().reshape(-1), vj[4], vj[5].t().reshape(-1), vj[6]])
        return (fe.detach(), vj[0], torch.zeros_like(ya[2]), vp_)
    with torch.no_grad():
        ref = T.odeint(aug, (y, a, adj_t, theta0), torch.tensor([1.0, 0.0]), rtol=tol, atol=tol, method='dopri5',
                       options={'max_num_steps': 1000})
    rs = T.odeint.last_stats
    print('   fused: attempts %d accepted %d dt %.6g | planes: %s' % (st['n_attempts'], st['n_accepted'], st['dt'], rs))
    print('   adj_y %.2e (scale %.2e)  adj_t %.2e  adj_params %.2e (scale %.2e)' % (
        sc(a_out - ref[1][1]), sc(ref[1][1]), sc(t_out - ref[2][1]), sc(p_out - ref[3][1]), sc(ref[3][1])))
    if time_it:
        for _ in range(2):
            eng.segment(mlp, y, a, adj_t, theta0, 1.0, 0.0)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 5
        for _ in range(n):
            eng.segment(mlp, y, a, adj_t, theta0, 1.0, 0.0)
        torch.cuda.synchronize()
        fused = (time.time() - t0) / n
        with torch.no_grad():
            T.odeint(aug, (y, a, adj_t, theta0), torch.tensor([1.0, 0.0]), rtol=tol, atol=tol, method='dopri5')
            torch.cuda.synchronize()
            t0 = time.time()
            T.odeint(aug, (y, a, adj_t, theta0), torch.tensor([1.0, 0.0]), rtol=tol, atol=tol, method='dopri5')
            torch.cuda.synchronize()
            planes = time.time() - t0
        print('   segment time: fused %.3f ms (%d attempts)  plane kernels + autograd %.1f ms' % (fused * 1e3, st['n_attempts'], planes * 1e3))
    eng.close()


if __name__ == '__main__':
    run(8, 4, 16)
    run(100, 10, 16)
    run(300, 64, 128)
    run(32768, 64, 128, time_it=True)
