"""GPU tests (-m gpu) of the linear right-hand side under odeint_adjoint (round 4; VERDICT r2 / r3: "no rhs.Linear adjoint - config 4's
training analogue").  Reference behaviour: tfdiffeq/adjoint.py:69-105 - the augmented dynamics are f and its three vector-Jacobian
products; for f = y W + b those are (y W + b, -a W^T, 0, -(y^T a), -sum_rows a).  `models.LinearODEFunc` gets them from the MFMA
kernels (mi_ode_eval_rhs, mi_ode_outer_reduce) instead of autograd over rocBLAS.  Checked:
  * mi_ode_outer_reduce against torch (float64: 1e-13 of the largest entry - another summation order; float32: 2e-5 ~ sqrt(batch) ulp);
  * gradients against the SAME backward solve with autograd dynamics (the generic path): same attempts, values to 1e-10;
  * gradients against float64 autograd through the oracle restatement of the reference's Dopri5 path.
"""
import copy
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


@pytest.mark.parametrize('dtype,band', [(torch.float64, 1e-13), (torch.float32, 2e-5)])
@pytest.mark.parametrize('batch,dim', [(1, 3), (5, 16), (1000, 33), (4097, 64), (70001, 128), (300, 100)])
def test_outer_reduce_against_torch(dtype, band, batch, dim):
    from tfdiffeq_amd import _native as N
    lib = N.load()
    g = torch.Generator().manual_seed(batch + dim)
    y = torch.randn(batch, dim, generator=g, dtype=dtype).to(dev())
    a = torch.randn(batch, dim, generator=g, dtype=dtype).to(dev())
    code = N.dtype_code(dtype)
    ws = torch.empty(int(lib.mi_ode_outer_workspace_bytes(code, batch, dim)), dtype=torch.uint8, device=dev())
    for scale, with_b in ((-1.0, True), (0.5, False)):
        w_out = torch.full((dim, dim), float('nan'), dtype=dtype, device=dev())
        b_out = torch.full((dim,), float('nan'), dtype=dtype, device=dev())
        N.check(lib.mi_ode_outer_reduce(code, batch, dim, y.data_ptr(), a.data_ptr(), scale, w_out.data_ptr(),
                                        b_out.data_ptr() if with_b else None, ws.data_ptr(), N.stream_ptr(dev())), 'mi_ode_outer_reduce')
        ref_w = scale * (y.double().t() @ a.double())
        ref_b = scale * a.double().sum(0)
        # the bound is on the products' magnitude, not on the (possibly cancelling) sum: sqrt(batch) * |y|max |a|max
        mag = float(batch) ** 0.5 * float(y.abs().max()) * float(a.abs().max())
        assert float((w_out.double() - ref_w).abs().max()) <= band * max(mag, float(ref_w.abs().max()))
        if with_b:
            assert float((b_out.double() - ref_b).abs().max()) <= band * max(float(batch) ** 0.5 * float(a.abs().max()), float(ref_b.abs().max()))
        else:
            assert torch.isnan(b_out).all()                          # untouched
    # deterministic: the same bits twice
    w1 = torch.empty(dim, dim, dtype=dtype, device=dev())
    w2 = torch.empty(dim, dim, dtype=dtype, device=dev())
    for w_ in (w1, w2):
        N.check(lib.mi_ode_outer_reduce(code, batch, dim, y.data_ptr(), a.data_ptr(), 1.0, w_.data_ptr(), None, ws.data_ptr(),
                                        N.stream_ptr(dev())), 'mi_ode_outer_reduce')
    assert torch.equal(w1, w2)


def _grads(func, y0, t, w, fused, **kw):
    from tfdiffeq_amd import adjoint as ADJ
    from tfdiffeq_amd import odeint_adjoint
    ADJ.FUSED = fused
    try:
        for p in func.parameters():
            p.grad = None
        yi = y0.clone().requires_grad_(True)
        sol = odeint_adjoint(func, yi, t, **kw)
        (sol * w).sum().backward()
        return sol.detach(), yi.grad.clone(), [p.grad.clone() for p in func.parameters()], dict(odeint_adjoint.last_backward_stats)
    finally:
        ADJ.FUSED = True


@pytest.mark.parametrize('batch,dim,bias,dtype', [(64, 8, True, torch.float64), (1000, 33, False, torch.float64), (5000, 128, True, torch.float64),
                                                  (300, 16, True, torch.float32)])
def test_linear_odefunc_gradients_against_the_autograd_dynamics(batch, dim, bias, dtype):
    from tfdiffeq_amd import models
    torch.manual_seed(dim)
    func = models.LinearODEFunc(dim, bias=bias, dtype=dtype).to(dev())
    if bias:
        with torch.no_grad():
            func.bias.normal_(0.0, 0.1)
    g = torch.Generator().manual_seed(batch)
    y0 = torch.randn(batch, dim, generator=g, dtype=dtype).to(dev())
    t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64)
    w = torch.randn(3, batch, dim, generator=g, dtype=dtype).to(dev())
    tol = dict(rtol=1e-7, atol=1e-9, method='dopri5') if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-5, method='dopri5')
    sol_f, gy_f, gp_f, st_f = _grads(func, y0, t, w, True, **tol)
    sol_p, gy_p, gp_p, st_p = _grads(func, y0, t, w, False, **tol)
    assert st_f['engine'].startswith('linear right-hand side') and st_p['engine'] == 'plane kernels'
    assert _rel(sol_f, sol_p) < (1e-12 if dtype == torch.float64 else 1e-5)     # (forward: whole-call MFMA kernel vs the callable path)
    # float64: the same algorithm with products summed in another order (MFMA slabs vs rocBLAS); the controller sees differences at
    # 1e-16 only.  float32: sqrt(batch) ulp per product, and the two controllers may then pick different, equally valid step sizes
    band = 1e-10 if dtype == torch.float64 else 2e-4
    assert _rel(gy_f, gy_p) < band
    for a_, b_ in zip(gp_f, gp_p):
        assert _rel(a_, b_) < band
    assert t.grad is None


def test_linear_odefunc_gradients_against_autograd_through_the_restatement():
    """The independent checker: float64 autograd through oracle/ode_torch_cpu.odeint_dopri5 at rtol 1e-10 = the exact gradient of the
    exact flow to 1e-9; the adjoint solve at rtol 1e-8 agrees with it to its own accuracy."""
    from tfdiffeq_amd import models, odeint_adjoint
    from oracle import ode_torch_cpu as TC
    torch.manual_seed(5)
    func = models.LinearODEFunc(12, bias=True).to(dev())
    with torch.no_grad():
        func.bias.normal_(0.0, 0.2)
    cpu = copy.deepcopy(func).cpu()
    y0 = torch.randn(50, 12, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    w = torch.randn(50, 12, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    y64 = y0.clone().requires_grad_(True)
    sol64, _ = TC.odeint_dopri5(lambda t_, y_: cpu(t_, y_), y64, [0.0, 1.5], rtol=1e-10, atol=1e-12)
    (sol64[1] * w).sum().backward()
    yi = y0.to(dev()).requires_grad_(True)
    sol = odeint_adjoint(func, yi, torch.tensor([0.0, 1.5], dtype=torch.float64), rtol=1e-8, atol=1e-10, method='dopri5')
    (sol[1] * w.to(dev())).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('linear right-hand side')
    assert _rel(yi.grad.cpu(), y64.grad) < 1e-6
    for pg, pc in zip(func.parameters(), cpu.parameters()):
        assert _rel(pg.grad.cpu(), pc.grad) < 1e-6


def test_forward_of_the_linear_module_takes_the_whole_call_kernel_and_frozen_parameters_the_generic_path():
    from tfdiffeq_amd import models, odeint, odeint_adjoint
    torch.manual_seed(8)
    func = models.LinearODEFunc(32, bias=True).to(dev())
    y0 = torch.randn(2000, 32, dtype=torch.float64, device=dev())
    t = torch.tensor([0.0, 1.0], dtype=torch.float64)
    out = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-6, atol=1e-9, method='dopri5')
    assert odeint.last_stats.get('n_launches') == 1                 # forward: the whole-call MFMA kernel of config 4
    with torch.no_grad():
        ref = odeint(lambda t_, y_: func(t_, y_), y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
    assert _rel(out.detach(), ref.detach()) < 1e-10
    # an in-place optimizer step is seen by both directions (the descriptor reads the parameter's own storage)
    with torch.no_grad():
        func.weight.mul_(0.5)
    out2 = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-6, atol=1e-9, method='dopri5')
    with torch.no_grad():
        ref2 = odeint(lambda t_, y_: func(t_, y_), y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
    assert _rel(out2.detach(), ref2.detach()) < 1e-10 and _rel(out2.detach(), out.detach()) > 1e-3
    out2[-1].sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('linear right-hand side')
    func.bias.requires_grad_(False)
    out3 = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-6, atol=1e-9, method='dopri5')
    out3[-1].sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'] == 'plane kernels'
