"""GPU tests of the float64 MLP tile kernels (csrc/mi_ode_mlp64.h, round 6): the ODEFunc network dim -> hidden -> hidden -> dim in
float64 on v_mfma_f64_16x16x4_f64 - whole call in one launch, launch per attempt, fixed grid - against the numpy oracle run on the same
network (same attempt / accept counts, float64 agreement) and against the same network as a Python callable."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _net(d, h, act, seed, td=False):
    from tfdiffeq_amd import rhs
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)   # noqa: E731
    W1, b1 = r(d + (1 if td else 0), h) / np.sqrt(d), 0.1 * r(h)
    W2, b2 = r(h, h) / np.sqrt(h), 0.1 * r(h)
    W3, b3 = r(h, d) / np.sqrt(h), 0.1 * r(d)
    return rhs.MLP(W1.to(dev()), b1.to(dev()), W2.to(dev()), b2.to(dev()), W3.to(dev()), b3.to(dev()), activation=act, time_dependent=td), \
        [x.numpy() for x in (W1, b1, W2, b2, W3, b3)]


def _np_f(ws, act, td):
    W1, b1, W2, b2, W3, b3 = ws
    a = {'tanh': np.tanh, 'relu': lambda x: np.maximum(x, 0), 'softplus': lambda x: np.log1p(np.exp(x))}[act]

    def f(t, y):
        x = np.concatenate([np.full(y.shape[:-1] + (1,), t), y], -1) if td else y
        return a(a(x @ W1 + b1) @ W2 + b2) @ W3 + b3
    return f


@pytest.mark.parametrize('d,h,batch,act,td', [(64, 128, 4096, 'tanh', False), (64, 128, 100, 'relu', False), (64, 128, 33, 'softplus', True),
                                               (6, 16, 48, 'tanh', False), (16, 16, 1000, 'softplus', False), (48, 100, 777, 'tanh', True),
                                               (3, 128, 65, 'relu', False)])
@pytest.mark.parametrize('method', ['dopri5', 'bosh3', 'tsit5', 'dopri8'])
def test_float64_network_against_the_oracle(d, h, batch, act, td, method):
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint
    if method in ('bosh3', 'tsit5', 'dopri8') and (d, h) not in ((64, 128), (6, 16)):
        pytest.skip('covered by dopri5 for this geometry')
    m, ws = _net(d, h, act, 11 * d + h, td)
    g = torch.Generator().manual_seed(5)
    y0 = torch.randn(batch, d, generator=g, dtype=torch.float64)
    t = np.array([0., 0.4, 1.0])
    rtol, atol = (1e-6, 1e-8) if method != 'bosh3' else (1e-4, 1e-6)
    sol = odeint(m, y0.to(dev()), torch.tensor(t), rtol=rtol, atol=atol, method=method)
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1 and st['status'] == 0, st
    if batch > 1000 and method != 'dopri5':
        return
    n = min(batch, 512)
    if n == batch and method != 'tsit5':        # (tsit5: the oracle follows the reference's defective tableau, the product the published one)
        ref, info = O.odeint(_np_f(ws, act, td), y0.numpy(), t, rtol=rtol, atol=atol, method=method, return_info=True) \
            if 'return_info' in O.odeint.__code__.co_varnames else (O.odeint(_np_f(ws, act, td), y0.numpy(), t, rtol=rtol, atol=atol, method=method), None)
        assert float(np.abs(sol.cpu().numpy() - ref).max()) < (1e-7 if method == 'dopri8' else 1e-9), float(np.abs(sol.cpu().numpy() - ref).max())
    # the same network as a Python callable on the callable engine: same attempt sequence
    ref2 = odeint(lambda t_, y: m.forward(t_, y), y0.to(dev()), torch.tensor(t), rtol=rtol, atol=atol, method=method, options={'lower': False})
    rst = dict(odeint.last_stats)
    assert (st['n_attempts'], st['n_accepted']) == (rst['n_attempts'], rst['n_accepted']), (st, rst)
    assert float((sol - ref2).abs().max()) < (1e-7 if method == 'dopri8' else 1e-9)


@pytest.mark.parametrize('fusion', ['step', 'whole'])
def test_float64_schedules_are_bit_identical(fusion):
    from tfdiffeq_amd import odeint
    m, _ = _net(64, 128, 'tanh', 3)
    y0 = torch.randn(300, 64, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev())
    t = torch.tensor([0., 0.5, 1.0])
    a = odeint(m, y0, t, rtol=1e-6, atol=1e-8, method='dopri5', options={'fusion': fusion})
    st = dict(odeint.last_stats)
    b = odeint(m, y0, t, rtol=1e-6, atol=1e-8, method='dopri5')
    assert torch.equal(a, b)
    assert (st['n_launches'] == 1) == (fusion == 'whole')


@pytest.mark.parametrize('method', ['euler', 'rk4'])
def test_float64_network_on_a_fixed_grid(method):
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint
    m, ws = _net(40, 72, 'tanh', 9, td=True)
    y0 = torch.randn(130, 40, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    t = np.linspace(0., 1., 12)
    sol = odeint(m, y0.to(dev()), torch.tensor(t), method=method)
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1
    ref = O.odeint(_np_f(ws, 'tanh', True), y0.numpy(), t, method=method)
    assert float(np.abs(sol.cpu().numpy() - ref).max()) < 1e-11


def test_float64_odeblock_uses_the_tile_kernels_and_trains():
    """models.ODEBlock over a float64 ODEFunc 64-128-128-64 (the reference's dense_odenet.py:41-92 in its tests' dtype): inference in one
    launch on k_persist_mlp64; training: forward on it, backward on the generic adjoint; the weights are re-read on every call."""
    from tfdiffeq_amd import models, odeint, odeint_adjoint
    torch.manual_seed(0)
    blk = models.ODEBlock(models.ODEFunc(64, 128, non_linearity='tanh'), tol=1e-5).to(dev()).double()
    x = torch.randn(4096, 64, dtype=torch.float64, device=dev())
    with torch.no_grad():
        out = blk(x)
        st = dict(odeint.last_stats)
        assert st['n_launches'] == 1 and 'engine' not in st or 'callable' not in str(st.get('engine')), st
        ref = odeint(lambda t, y: blk.odefunc(t, y), x, torch.tensor([0., 1.]), rtol=1e-5, atol=1e-5, method='dopri5', options={'lower': False, 'max_num_steps': 1000})[1]
        assert float((out - ref).abs().max()) < 1e-8
        with torch.no_grad():
            blk.odefunc.fc2.weight.mul_(0.5)
        out2 = blk(x)
        assert float((out2 - out).abs().max()) > 1e-6           # the pack is refreshed: no stale weights
    xs = x[:256].clone().requires_grad_(True)
    blk(xs).pow(2).sum().backward()
    assert xs.grad is not None and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in blk.odefunc.parameters())
    assert odeint_adjoint.last_backward_stats['forward']['n_launches'] == 1
