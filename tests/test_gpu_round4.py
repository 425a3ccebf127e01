"""GPU tests (-m gpu) of the kernels round 4 added to the hot path's neighbours (SURVEY.md 8(f) / VERDICT r3 items 5, 7):
  * `k_fixed_mlp`: euler / rk4 (3/8 rule) on a fixed grid for the ODEFunc MLP (models/dense_odenet.py:41-92 through
    fixed_grid.py:6-42) in ONE launch - against the numpy oracle's fixed-grid solver over a float32 numpy network, and against
    the plane-kernel engine running the same network as a Python callable.
  * plain callables that close over trainable state receive gradients (odeint.py:28-81 under the reference's tape).
"""
import numpy as np
import pytest
import torch

from oracle import ode_numpy as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _np_mlp(Ws, bs, act, time_dependent, dtype=np.float32):
    acts = {'tanh': np.tanh, 'relu': lambda x: np.maximum(x, dtype(0)), 'softplus': lambda x: np.logaddexp(x, dtype(0)).astype(dtype)}
    a = acts[act]
    Ws = [np.asarray(w, dtype=dtype) for w in Ws]
    bs = [np.asarray(b, dtype=dtype) for b in bs]

    def f(t, y):
        h = y
        if time_dependent:
            h = np.concatenate([np.full(y.shape[:-1] + (1,), dtype(t), dtype=dtype), y], axis=-1)
        h = a(h @ Ws[0] + bs[0])
        h = a(h @ Ws[1] + bs[1])
        return (h @ Ws[2] + bs[2]).astype(dtype)
    return f


@pytest.mark.parametrize('method', ['euler', 'rk4'])
@pytest.mark.parametrize('act', ['tanh', 'relu', 'softplus'])
@pytest.mark.parametrize('dim,hidden,td', [(64, 128, False), (6, 16, True), (10, 100, True)])
def test_mlp_fixed_grid_in_one_launch(method, act, dim, hidden, td):
    from tfdiffeq_amd import odeint, rhs
    g = torch.Generator().manual_seed(11 + dim)
    sc = 0.6
    W1 = torch.randn(dim + (1 if td else 0), hidden, generator=g) * (sc / np.sqrt(dim))
    W2 = torch.randn(hidden, hidden, generator=g) * (sc / np.sqrt(hidden))
    W3 = torch.randn(hidden, dim, generator=g) * (sc / np.sqrt(hidden))
    b1, b2, b3 = (torch.randn(n, generator=g) * 0.1 for n in (hidden, hidden, dim))
    mlp = rhs.MLP(W1, b1, W2, b2, W3, b3, activation=act, time_dependent=td)
    batch = 1000 if dim > 10 else 77                                   # (ragged last tile in both cases)
    y0 = torch.randn(batch, dim, generator=g)
    fn = _np_mlp([W1.numpy(), W2.numpy(), W3.numpy()], [b1.numpy(), b2.numpy(), b3.numpy()], act, td)
    for t in (np.linspace(0., 1., 21), -np.linspace(0., 0.5, 6) ** 2):
        tt = torch.tensor(t)
        got = odeint(mlp, y0.to(dev()), tt, method=method)
        st = dict(odeint.last_stats)
        assert st['n_launches'] == 1 and st['status'] == 0, st          # the whole fixed-grid integration: one launch
        assert tuple(got.shape) == (len(t), batch, dim) and got.dtype == torch.float32
        ref = O.odeint(fn, y0.numpy(), t.astype(np.float32), method=method)
        # float32 network, <= 20 steps: matmul summation order (MFMA vs numpy) and the activation's last bits (v_exp_f32 based tanh
        # / softplus, 5e-7 relative) - a-priori 20 steps x 4 evaluations x ~1e-6 = 1e-4; observed ~3e-6
        dev_ = np.abs(got.cpu().numpy() - ref) / (1.0 + np.abs(ref))
        assert dev_.max() < 1e-4, (method, act, dim, dev_.max())
        assert torch.equal(got[0].cpu(), y0)
        # the same network as an opaque Python callable on the plane kernels (torch matmuls)
        gen = odeint(lambda t_, y_: mlp.forward(t_, y_), y0.to(dev()), tt, method=method)
        assert ((got - gen).abs() / (1.0 + gen.abs())).max().item() < 1e-4
    # a grid of its own (step_size) with requested times between grid points (solvers.py:86-115), and eps (fixed_grid.py:7)
    tt = torch.tensor([0., 0.33, 0.7])
    for opts in ({'step_size': 0.1}, {'eps': 1e-3}):
        got = odeint(mlp, y0.to(dev()), tt, method=method, options=opts)
        st = dict(odeint.last_stats)
        assert st['n_launches'] == 1, st
        gen = odeint(lambda t_, y_: mlp.forward(t_, y_), y0.to(dev()), tt, method=method, options=opts)
        assert ((got - gen).abs() / (1.0 + gen.abs())).max().item() < 1e-4, opts


def test_odefunc_module_on_a_fixed_grid_takes_the_fused_kernel():
    """models.ODEFunc (the reference's dense_odenet.ODEFunc) with method='rk4': its device descriptor now has a fixed-grid kernel."""
    from tfdiffeq_amd import models, odeint
    torch.manual_seed(3)
    func = models.ODEFunc(12, 32, non_linearity='tanh').to(dev())
    y0 = torch.randn(200, 12, device=dev())
    t = torch.linspace(0., 1., 11)
    with torch.no_grad():
        a = odeint(func.device_rhs(), y0, t, method='rk4')
        assert odeint.last_stats['n_launches'] == 1
        b = odeint(lambda t_, y_: func(t_, y_), y0, t, method='rk4')
    assert ((a - b).abs() / (1.0 + b.abs())).max().item() < 1e-4


def test_plain_callable_closing_over_a_module_receives_gradients():
    """`odeint(lambda t, y: net(y), y0, t)` with y0 requiring grad used to train nothing - the wrapper module
    had no parameters.  The closure is searched now (closure cells, bound objects, partials, globals): net's parameters and a bare
    grad-requiring tensor get the gradients odeint_adjoint gives a module."""
    import copy
    from tfdiffeq_amd import odeint, odeint_adjoint
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3)).double().to(dev())
    scale = torch.tensor(0.7, dtype=torch.float64, device=dev(), requires_grad=True)
    y0 = torch.randn(5, 3, dtype=torch.float64, device=dev(), requires_grad=True)
    t = torch.tensor([0., 0.5, 1.0], dtype=torch.float64)
    out = odeint(lambda t_, y: scale * net(y), y0, t, rtol=1e-8, atol=1e-10)
    out[-1].pow(2).sum().backward()
    got = [p.grad.clone() for p in net.parameters()] + [scale.grad.clone(), y0.grad.clone()]
    assert all(g is not None and torch.isfinite(g).all() for g in got) and got[0].abs().max() > 0

    class M(torch.nn.Module):                               # the same system as a module: what odeint_adjoint always handled
        def __init__(self):
            super().__init__()
            self.net = copy.deepcopy(net)
            self.scale = torch.nn.Parameter(scale.detach().clone())

        def forward(self, t_, y):
            return self.scale * self.net(y)
    m = M()
    for p in m.parameters():
        p.grad = None
    y1 = y0.detach().clone().requires_grad_(True)
    ref = odeint_adjoint(m, y1, t, rtol=1e-8, atol=1e-10)
    ref[-1].pow(2).sum().backward()
    want = [p.grad for p in m.net.parameters()] + [m.scale.grad, y1.grad]
    for a, b in zip(got, want):
        assert (a - b).abs().max().item() <= 1e-9 * max(1.0, b.abs().max().item())
    # y0 without grad: the trainable closure alone routes the call to the adjoint
    for p in net.parameters():
        p.grad = None
    out = odeint(lambda t_, y: net(y), y0.detach(), t, rtol=1e-6, atol=1e-8)
    out[-1].sum().backward()
    assert all(p.grad is not None and p.grad.abs().max() > 0 for p in net.parameters())


# ---------------------------------------------------------------------------------------------
# the Adams family for matrix right-hand sides of any dim <= 256 in one launch (DETEST C1-C4 took the
# per-step loop and were 3x slower than numpy): a thread per state element, floor(256 / dim) trajectories per workgroup
# ---------------------------------------------------------------------------------------------
def _stable(dim, seed):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((dim, dim))
    return -0.5 * np.eye(dim) + 0.5 * (S - S.T) / np.sqrt(dim)


@pytest.mark.parametrize('method', ['adams', 'explicit_adams', 'fixed_adams'])
@pytest.mark.parametrize('dim,batch', [(3, 1), (10, 1), (51, 1), (51, 9), (128, 300), (7, 2000)])
def test_adams_family_for_matrix_systems_in_one_launch(method, dim, batch):
    from oracle import adams_numpy as OA
    from tfdiffeq_amd import odeint, rhs
    A = _stable(dim, 40 + dim)
    W = A.T.copy()
    rng = np.random.default_rng(dim)
    y0 = rng.standard_normal((batch, dim))
    f = rhs.Linear(torch.tensor(W))
    fn = lambda t_, y: y @ W  # noqa: E731
    t = np.linspace(0., 2.0, 5) if method == 'adams' else np.linspace(0., 1.0, 41)
    tol = dict(rtol=1e-6, atol=1e-8)
    for tt in (t, -t):
        got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        st = dict(odeint.last_stats)
        assert st.get('engine', '').startswith('fused') and st['n_launches'] == 1 and st['status'] == 0, st
        ref, rst = OA.odeint(fn, y0, tt, method=method, return_stats=True, **tol)
        scale = max(1.0, np.abs(np.asarray(ref)).max())
        if method == 'adams':
            n_acc = int(sum(1 for r in rst.trace if r[3] > 0))
            assert (st['n_attempts'], st['n_accepted']) == (len(rst.trace), n_acc), (st, len(rst.trace), n_acc)
            band = 1e-6                    # g is rounded to float32 every step (adams.py:34): see test_gpu_multistep_fused.py
        else:
            band = 1e-8                    # Adams-Bashforth weights of order 12 reach ~1e4 with alternating signs: the dot products' summation
                                           # order (fma chain here, BLAS in numpy / torch) shows up at 1e-10 after 40 steps
        assert np.abs(got.cpu().numpy() - np.asarray(ref)).max() <= band * scale, (method, dim, batch)
        loop = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, options={'force_plane_kernels': True} if method == 'adams'
                      else {'fusion': 'stage'}, **tol)
        assert not str(dict(odeint.last_stats).get('engine', '')).startswith('fused')
        assert float((got - loop).abs().max()) <= band * scale


def test_cubic_matrix_system_and_bias_on_the_cooperative_multistep_kernels():
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(8)
    W = torch.tensor(_stable(6, 3).T.copy())
    y0 = torch.tensor(0.5 * rng.standard_normal((40, 6)), device=dev())
    for f in (rhs.CubicLinear(W), rhs.Linear(W, torch.tensor(0.1 * rng.standard_normal(6)))):
        for method in ('adams', 'fixed_adams'):
            # (fixed grid: steps inside the stability region of the order-12 Adams pair - at dt = 0.025 the reference's explicit solver
            # blows up on this oscillatory system and the implicit one does not converge, on any engine)
            t = torch.tensor(np.linspace(0., 1., 6) if method == 'adams' else np.linspace(0., 0.2, 41))
            a = odeint(f, y0, t, method=method, rtol=1e-6, atol=1e-8)
            assert dict(odeint.last_stats).get('engine', '').startswith('fused'), odeint.last_stats
            b = odeint(lambda t_, y: f.forward(t_, y), y0, t, method=method, rtol=1e-6, atol=1e-8)
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
