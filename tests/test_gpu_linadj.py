"""GPU tests (-m gpu) of the one-launch backward segment of odeint_adjoint for the linear right-hand side (round 5; include/mi_ode.h
section A''', csrc/mi_ode_linadj.h).  Reference behaviour: tfdiffeq/adjoint.py:117-178 - per output interval, odeint over the tuple
(y, adj_y, adj_t, adj_params) with the dynamics of adjoint.py:69-105, which for f = y W + b are (y W + b, -adj_y W^T, 0, -(y^T adj_y) | -sum_rows adj_y).
Checked:
  * mi_ode_linadj_segment, called through the C ABI, against the numpy ORACLE solving that very tuple system (oracle/ode_numpy.odeint over
    the restated augmented dynamics): the same attempt / accept counts, adj_y(t_end), adj_t(t_end), adj_params(t_end) to 1e-11 (float64);
  * odeint_adjoint end to end against the callable-engine path of round 4 (one Python evaluation per stage): same attempts per interval,
    gradients to 1e-11, one launch per interval; time gradients included;
  * the fall-backs (another adjoint method, solver options) keep the callable engine.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


def _matrix(dim, rng):
    S = rng.standard_normal((dim, dim))
    return -0.4 * np.eye(dim) + 0.6 * (S - S.T) / np.sqrt(dim) + 0.1 * rng.standard_normal((dim, dim)) / np.sqrt(dim)


@pytest.mark.parametrize('batch,dim,bias,t0,t1', [(40, 5, True, 1.0, 0.0), (300, 16, False, 0.7, 0.1), (1000, 33, True, 1.0, 0.25), (2100, 128, True, 0.5, 0.0),
                                                 (700, 100, False, 0.0, 0.8)])
def test_segment_through_the_c_abi_against_the_oracle(batch, dim, bias, t0, t1):
    """One backward interval t0 -> t1 (the last case: increasing time, the sign s = +1) against oracle/ode_numpy.odeint on the tuple state."""
    from oracle import ode_numpy as O
    from tfdiffeq_amd import adjoint as ADJ
    rng = np.random.default_rng(batch + dim)
    W = _matrix(dim, rng)
    b = 0.3 * rng.standard_normal(dim) if bias else None
    y = rng.standard_normal((batch, dim))
    a = rng.standard_normal((batch, dim)) / batch
    g = rng.standard_normal((batch, dim)) / batch                # grad_output at t0: the time gradient's dot product
    adjt = np.array(0.37)
    th = 0.01 * rng.standard_normal(dim * dim + (dim if bias else 0))
    rtol, atol = 1e-7, 1e-10

    def aug(t_, s_):                                             # adjoint.py:69-105 for f = y W + b
        y_, a_ = s_[0], s_[1]
        f = y_ @ W + (b if bias else 0.0)
        vth = -(y_.T @ a_).reshape(-1)
        if bias:
            vth = np.concatenate([vth, -a_.sum(0)])
        return (f, -(a_ @ W.T), np.zeros_like(s_[2]), vth)

    dl = float(((y @ W + (b if bias else 0.0)) * g).sum())       # adjoint.py:134-140
    sol, stats = O.odeint(aug, (y, a, adjt - dl, th), np.array([t0, t1]), rtol=rtol, atol=atol, method='dopri5', return_stats=True)
    f32 = lambda v: float(np.float32(v))  # noqa: E731
    eng = ADJ._LinearAdjointEngine(batch, dim, torch.float64, rtol, atol, f32(0.9), f32(10.0), f32(0.2), 2 ** 31 - 1, str(dev()))
    try:
        tt = lambda x: torch.tensor(x, dtype=torch.float64, device=dev())  # noqa: E731
        a_out, t_out, p_out = eng.segment(tt(W), tt(b) if bias else None, tt(y), tt(a), tt(adjt), tt(th), t0, t1, grad_out=tt(g))
        st = eng.stats.as_dict()
        assert st['n_launches'] == 1 and st['status'] == 0
        assert st['n_attempts'] == stats.n_attempts and st['n_accepted'] == stats.n_accepted
        assert abs(eng.scalars[0] - dl) <= 1e-12 * max(1.0, abs(dl))
        for got, ref in ((a_out, sol[1][1]), (t_out, sol[2][1]), (p_out, sol[3][1])):
            ref = np.asarray(ref)
            assert np.abs(got.cpu().numpy() - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        assert abs(eng.scalars[1] - float(sol[2][1])) <= 1e-12 * max(1.0, abs(float(sol[2][1])))
    finally:
        eng.close()


def _grads(func, y0, t, w, one_launch, **kw):
    from tfdiffeq_amd import adjoint as ADJ
    from tfdiffeq_amd import odeint_adjoint
    ADJ.LINEAR_ONE_LAUNCH = one_launch
    try:
        for p in func.parameters():
            p.grad = None
        yi = y0.clone().requires_grad_(True)
        tt = t.clone().requires_grad_(True)
        sol = odeint_adjoint(func, yi, tt, **kw)
        (sol * w).sum().backward()
        return sol.detach(), yi.grad.clone(), [p.grad.clone() for p in func.parameters()], tt.grad.clone(), dict(odeint_adjoint.last_backward_stats)
    finally:
        ADJ.LINEAR_ONE_LAUNCH = True


@pytest.mark.parametrize('batch,dim,bias,dtype,tpts', [(64, 8, True, torch.float64, [0.0, 1.0]), (1000, 33, False, torch.float64, [0.0, 0.4, 1.0]),
                                                      (5000, 128, True, torch.float64, [0.0, 0.4, 1.0]), (70001, 128, False, torch.float64, [0.0, 1.0]),
                                                      (4100, 100, True, torch.float64, [0.0, 0.3, 0.5, 1.0]), (300, 16, True, torch.float32, [0.0, 0.4, 1.0])])
def test_one_launch_backward_against_the_callable_engine(batch, dim, bias, dtype, tpts):
    from tfdiffeq_amd import models
    torch.manual_seed(dim)
    func = models.LinearODEFunc(dim, bias=bias, dtype=dtype).to(dev())
    if bias:
        with torch.no_grad():
            func.bias.normal_(0.0, 0.1)
    g = torch.Generator().manual_seed(batch)
    y0 = torch.randn(batch, dim, generator=g, dtype=dtype).to(dev())
    t = torch.tensor(tpts, dtype=torch.float64)
    w = torch.randn(len(tpts), batch, dim, generator=g, dtype=dtype).to(dev())
    tol = dict(rtol=1e-7, atol=1e-9, method='dopri5') if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-5, method='dopri5')
    a = _grads(func, y0, t, w, True, **tol)
    b = _grads(func, y0, t, w, False, **tol)
    assert a[4]['engine'].startswith('linear right-hand side: one launch per interval')
    assert b[4]['engine'].startswith('linear right-hand side: augmented dynamics on the MFMA kernels')
    segs = a[4]['segments']
    assert len(segs) == len(tpts) - 1 and all(s_['n_launches'] == 1 for s_ in segs)
    assert segs[-1]['n_attempts'] == b[4]['last_segment']['n_attempts']           # (the callable engine keeps the last interval's statistics)
    # float64: the same algorithm with the parameter component in its power form (1e-15 per step); float32: the parameter component is
    # carried in float64 here and in float32 there - sqrt(batch) ulp per product on that side
    band = 1e-11 if dtype == torch.float64 else 2e-4
    assert _rel(a[1], b[1]) < band
    for x, y in zip(a[2], b[2]):
        assert _rel(x, y) < band
    assert float((a[3] - b[3]).abs().max()) <= band * max(1.0, float(b[3].abs().max()))


def test_other_adjoint_methods_and_options_keep_the_callable_engine():
    from tfdiffeq_amd import models, odeint_adjoint
    torch.manual_seed(1)
    func = models.LinearODEFunc(16, bias=True).to(dev())
    y0 = torch.randn(200, 16, dtype=torch.float64, device=dev())
    t = torch.tensor([0.0, 1.0], dtype=torch.float64)
    for kw in (dict(method='dopri5', adjoint_method='bosh3'), dict(method='dopri5', adjoint_options={'first_step': 0.01})):
        out = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-6, atol=1e-9, **kw)
        out[-1].sum().backward()
        assert odeint_adjoint.last_backward_stats['engine'].startswith('linear right-hand side: augmented dynamics on the MFMA kernels')
    out = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-6, atol=1e-9, method='dopri5', adjoint_options={'max_num_steps': 1000})
    out[-1].sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('linear right-hand side: one launch per interval')


def test_status_bits_become_the_references_assertions():
    """max_num_steps (dopri5.py:85-86) through the one-launch kernel."""
    from tfdiffeq_amd import models, odeint_adjoint
    torch.manual_seed(2)
    func = models.LinearODEFunc(16, bias=False).to(dev())
    y0 = torch.randn(100, 16, dtype=torch.float64, device=dev())
    t = torch.tensor([0.0, 5.0], dtype=torch.float64)
    out = odeint_adjoint(func, y0.clone().requires_grad_(True), t, rtol=1e-9, atol=1e-12, method='dopri5', adjoint_options={'max_num_steps': 3})
    with pytest.raises(AssertionError, match='max_num_steps exceeded'):
        out[-1].sum().backward()
