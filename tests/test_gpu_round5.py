"""GPU tests (-m gpu) added in round 5 outside the linear adjoint (tests/test_gpu_linadj.py): the advisor's round-4 findings."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


class _Neg(torch.nn.Module):
    """f = -y: its backward saves no tensor - the saved-tensor heuristic of the device-controlled engine sees nothing."""

    def forward(self, t, y):
        return -y


class _Shift(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.b = torch.nn.Parameter(torch.full((3,), 0.25, dtype=torch.float64))

    def forward(self, t, y):
        return y + self.b


@pytest.mark.parametrize('make', [_Neg, _Shift])
def test_backward_dynamics_with_autograd_inside_are_never_recorded(make):
    """(advisor, round 4, medium) odeint_adjoint's backward solve runs on the device-controlled engine, whose default records an attempt as a
    hipGraph once enough attempts remain - with torch.autograd.grad inside, that aborts the process.  The augmented dynamics are marked
    `_mi_no_capture`; a long backward interval (dozens of attempts) completes eagerly and the gradients are those of the exact flow."""
    from tfdiffeq_amd import odeint, odeint_adjoint
    func = make().to(dev())
    y0 = torch.tensor([[1.0, -2.0, 0.5], [0.3, 0.1, -0.7]], dtype=torch.float64, device=dev(), requires_grad=True)
    t = torch.tensor([0.0, 6.0], dtype=torch.float64)
    out = odeint_adjoint(func, y0, t, rtol=1e-10, atol=1e-12, method='dopri5')
    out[-1].sum().backward()
    st = odeint.last_stats                                        # (the backward odeint call was the last one)
    assert st.get('n_attempts', 0) >= 14                          # more than EAGER_FIRST + MIN_REMAINING: 'auto' would have recorded
    assert 'hipGraph' not in str(st.get('engine', '')) and st.get('replays', 0) == 0
    if make is _Neg:
        assert torch.allclose(y0.grad, torch.full_like(y0, float(torch.exp(torch.tensor(-6.0, dtype=torch.float64)))), rtol=1e-7)
    else:
        e6 = float(torch.exp(torch.tensor(6.0, dtype=torch.float64)))
        assert torch.allclose(y0.grad, torch.full_like(y0, e6), rtol=1e-7)
        assert torch.allclose(func.b.grad, torch.full((3,), 2 * (e6 - 1.0), dtype=torch.float64, device=dev()), rtol=1e-6)


def test_mlp_fixed_grid_with_stage_fusion_falls_through_to_the_step_loop():
    """(advisor, round 4, low) rhs.MLP has a one-launch fixed-grid kernel but no per-stage kernels: options={'fusion': 'stage'} used to raise
    NativeError from mi_ode_create; it takes the per-step loop again, same values as the one-launch kernel."""
    from tfdiffeq_amd import models, odeint
    torch.manual_seed(0)
    f = models.ODEFunc(6, 16, non_linearity='tanh').to(dev())
    r = f.device_rhs()
    y0 = torch.randn(40, 6, device=dev())
    t = torch.linspace(0.0, 1.0, 5)
    with torch.no_grad():
        a = odeint(r, y0, t, method='rk4')
        b = odeint(r, y0, t, method='rk4', options={'fusion': 'stage'})
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_plain_callable_gets_gradients_for_exactly_the_leaves_it_uses():
    """(advisor, round 4, medium) `odeint(lambda t, y: net(y) ...)` under grad mode: the tensors that receive gradients are the leaves of the
    callable's autograd graph - a module it merely names gets None (not zeros: an optimizer with weight decay would move it), a tensor behind
    an attribute chain is found."""
    from tfdiffeq_amd import odeint
    torch.manual_seed(1)
    used = torch.nn.Linear(4, 4).double().to(dev())
    unused = torch.nn.Linear(4, 4).double().to(dev())

    class Box(object):
        pass
    box = Box()
    box.inner = Box()
    box.inner.scale = torch.tensor(0.5, dtype=torch.float64, device=dev(), requires_grad=True)

    def f(t, y):
        _ = unused
        return torch.tanh(used(y)) * box.inner.scale
    y0 = torch.randn(16, 4, dtype=torch.float64, device=dev())
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out = odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), rtol=1e-8, atol=1e-10, method='dopri5')
    out[-1].pow(2).sum().backward()
    assert used.weight.grad is not None and used.bias.grad is not None and box.inner.scale.grad is not None
    assert unused.weight.grad is None and unused.bias.grad is None
    # against autograd through a fine fixed-step integration of the same function (float64, 400 rk4 steps)
    for p in (used.weight, used.bias, box.inner.scale):
        p.grad_ref, p.grad = p.grad.clone(), None
    y = y0.clone()
    h = 1.0 / 400
    for _ in range(400):
        k1 = f(0, y); k2 = f(0, y + 0.5 * h * k1); k3 = f(0, y + 0.5 * h * k2); k4 = f(0, y + h * k3)
        y = y + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    y.pow(2).sum().backward()
    for p in (used.weight, used.bias, box.inner.scale):
        assert float((p.grad - p.grad_ref).abs().max()) <= 1e-6 * max(1.0, float(p.grad.abs().max()))


def test_sixteen_dimensional_user_system_in_one_launch():
    """(round-4 review, item 5) rhs.CustomRowLocal beyond dim 8: a ring of eight coupled damped oscillators (dim 16) as user device code -
    the whole adaptive integration in ONE launch, against the numpy oracle on the same function (float64: same attempt counts, 1e-9)."""
    import numpy as np
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint, rhs
    from tfdiffeq_amd import plugin_examples
    p0, p1, p2 = 1.0, 0.5, 0.05
    r = plugin_examples.oscillator_ring(8, p0, p1, p2)          # (prebuilt by __graft_entry__.build(): no hipcc run on the GPU box)

    def f_np(t, y):
        x, v = y[..., :8], y[..., 8:]
        return np.concatenate([v, -p0 * x + p1 * (np.roll(x, 1, -1) - 2.0 * x + np.roll(x, -1, -1)) - p2 * v], axis=-1)
    rng = np.random.default_rng(3)
    y0 = rng.standard_normal((777, 16))
    t = np.array([0.0, 1.5, 4.0])
    sol = odeint(r, torch.tensor(y0, device=dev()), torch.tensor(t), rtol=1e-7, atol=1e-9, method='dopri5')
    st = dict(odeint.last_stats)
    ref, rs = O.odeint(f_np, y0, t, rtol=1e-7, atol=1e-9, method='dopri5', return_stats=True)
    assert st['n_launches'] == 1 and st['status'] == 0
    assert (st['n_attempts'], st['n_accepted']) == (rs.n_attempts, rs.n_accepted)
    assert np.abs(sol.cpu().numpy() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # ... and a float32 system of dim 32 (two such rings): one launch, inside the float32 a-priori ceiling of its attempt count
    from tests.bands import case_ceiling, observed
    r32 = plugin_examples.oscillator_ring(16, p0, p1, p2)

    def f32_np(t, y):
        x, v = y[..., :16], y[..., 16:]
        return np.concatenate([v, -p0 * x + p1 * (np.roll(x, 1, -1) - 2.0 * x + np.roll(x, -1, -1)) - p2 * v], axis=-1)
    y32 = rng.standard_normal((300, 32)).astype(np.float32)
    sol32 = odeint(r32, torch.tensor(y32, device=dev()), torch.tensor([0.0, 2.0]), rtol=1e-4, atol=1e-6, method='dopri5')
    st32 = dict(odeint.last_stats)
    ref32 = O.odeint(f32_np, y32.astype(np.float64), np.array([0.0, 2.0]), rtol=1e-4, atol=1e-6, method='dopri5')
    assert st32['n_launches'] == 1 and st32['status'] == 0
    assert observed(sol32.cpu().numpy(), ref32) <= max(2e-4, case_ceiling(st32['n_attempts']))     # (the solve's own tolerance is rtol 1e-4)


def test_two_layer_relu_network_of_the_user_runs_on_the_fused_mlp_kernels():
    """(round-4 review, item 5) rhs.from_sequential: nn.Sequential(Linear, ReLU, Linear) as a device descriptor - relu is idempotent, so the
    two-layer network IS the three-layer kernel with an identity middle layer; one launch per call, against the float64 numpy oracle on the
    same function inside the float32 ceiling; the three-layer tanh form maps one to one."""
    import numpy as np
    from oracle import ode_numpy as O
    from tests.bands import case_ceiling, observed
    from tfdiffeq_amd import odeint, rhs
    torch.manual_seed(4)
    net = torch.nn.Sequential(torch.nn.Linear(10, 48), torch.nn.ReLU(), torch.nn.Linear(48, 10)).to(dev())
    with torch.no_grad():
        for m in net:
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(0.5)
    r = rhs.from_sequential(net)
    assert r.activation == 'relu' and r.hidden == 48 and r.dim == 10
    y0 = torch.randn(999, 10, device=dev())
    t = torch.tensor([0.0, 0.5, 1.0])
    sol = odeint(r, y0, t, rtol=1e-5, atol=1e-6, method='dopri5')
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1 and st['status'] == 0
    W1, b1 = net[0].weight.detach().cpu().double().numpy().T, net[0].bias.detach().cpu().double().numpy()
    W2, b2 = net[2].weight.detach().cpu().double().numpy().T, net[2].bias.detach().cpu().double().numpy()
    ref = O.odeint(lambda t_, y: np.maximum(y @ W1 + b1, 0.0) @ W2 + b2, y0.cpu().double().numpy(), t.numpy().astype(np.float64), rtol=1e-5, atol=1e-6,
                   method='dopri5')
    assert observed(sol.cpu().numpy(), ref) <= max(5e-5, case_ceiling(st['n_attempts']))          # (rtol 1e-5 of the solve itself)
    # the Python callable of the same network takes the device-controlled engine: same function
    with torch.no_grad():
        sol_py = odeint(lambda t_, y: net(y), y0, t, rtol=1e-5, atol=1e-6, method='dopri5')
    assert observed(sol.cpu().numpy(), sol_py.cpu().numpy()) <= 5e-5
    net3 = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 6)).to(dev())
    r3 = rhs.from_sequential(net3)
    y3 = torch.randn(100, 6, device=dev())
    with torch.no_grad():
        a = odeint(r3, y3, t, rtol=1e-5, atol=1e-6, method='dopri5')
        assert dict(odeint.last_stats)['n_launches'] == 1
        b = odeint(lambda t_, y: net3(y), y3, t, rtol=1e-5, atol=1e-6, method='dopri5')
    assert observed(a.cpu().numpy(), b.cpu().numpy()) <= 5e-5
    with pytest.raises(ValueError):
        rhs.from_sequential(torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4)))


# ---------------------------------------------------------------------------------------------
# the Adams family for the ODEFunc network in one launch (round-4 review, item 7: "Adams kernels for the MLP family", and float64 /
# widths beyond the tile kernels' box): a thread per state element, the three layers through LDS (csrc/mi_ode_stage_rowlocal.h: RhsMlpCoop)
# ---------------------------------------------------------------------------------------------
def _np_mlp(Ws, bs, act, time_dependent, dtype):
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


@pytest.mark.parametrize('method', ['adams', 'explicit_adams', 'fixed_adams'])
@pytest.mark.parametrize('dim,hidden,batch,act,td,dtype', [
    (2, 50, 1, 'tanh', False, np.float64), (8, 16, 300, 'softplus', True, np.float64), (64, 128, 40, 'tanh', False, np.float64),
    (100, 200, 7, 'relu', False, np.float64), (16, 32, 1000, 'tanh', True, np.float32)])
def test_adams_family_for_the_mlp_in_one_launch(method, dim, hidden, batch, act, td, dtype):
    """'adams' / 'explicit_adams' / 'fixed_adams' on rhs.MLP: one launch, float64 and widths the tile kernels do not take included;
    against the numpy restatement of the reference's solvers over a numpy network (attempt counts exact for 'adams'), and against the
    host loop of the same solver over the same network as a Python callable."""
    from oracle import adams_numpy as OA
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(1000 + dim + hidden)
    sc = 0.7
    Ws = [sc * rng.standard_normal((dim + (1 if td else 0), hidden)) / np.sqrt(dim), sc * rng.standard_normal((hidden, hidden)) / np.sqrt(hidden),
          sc * rng.standard_normal((hidden, dim)) / np.sqrt(hidden)]
    bs = [0.1 * rng.standard_normal(hidden), 0.1 * rng.standard_normal(hidden), 0.1 * rng.standard_normal(dim)]
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    f = rhs.MLP(*[torch.tensor(v, dtype=tdt) for pair in zip(Ws, bs) for v in pair], activation=act, time_dependent=td)
    fn = _np_mlp(Ws, bs, act, td, dtype)
    y0 = rng.standard_normal((batch, dim)).astype(dtype)
    f64 = dtype == np.float64
    t = np.linspace(0., 1.5, 4) if method == 'adams' else np.linspace(0., 0.5, 26)
    tol = dict(rtol=1e-6, atol=1e-8) if f64 else dict(rtol=1e-4, atol=1e-6)
    for tt in (t, -t):
        got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        st = dict(odeint.last_stats)
        assert st.get('engine', '').startswith('fused') and st['n_launches'] == 1 and st['status'] == 0, st
        ref, rst = OA.odeint(fn, y0, tt, method=method, return_stats=True, **tol)
        scale = max(1.0, np.abs(np.asarray(ref)).max())
        if method == 'adams' and f64:
            n_acc = int(sum(1 for r in rst.trace if r[3] > 0))
            assert (st['n_attempts'], st['n_accepted']) == (len(rst.trace), n_acc), (st, len(rst.trace), n_acc)
        band = (1e-6 if method == 'adams' else 1e-8) if f64 else 2e-4
        assert np.abs(got.cpu().numpy().astype(np.float64) - np.asarray(ref, dtype=np.float64)).max() <= band * scale, (method, dim, hidden, batch)
        loop = odeint(lambda t_, y: f.forward(t_, y), torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        assert not str(dict(odeint.last_stats).get('engine', '')).startswith('fused')
        assert float((got - loop).abs().max()) <= band * scale


def test_mlp_beyond_the_multistep_kernels_takes_the_host_loop():
    from tfdiffeq_amd import odeint, rhs
    g = torch.Generator().manual_seed(3)
    d, h = 4, 300                                                            # hidden > 256: no one-launch multistep kernel
    f = rhs.MLP(torch.randn(d, h, generator=g, dtype=torch.float64) / 2, None, torch.randn(h, h, generator=g, dtype=torch.float64) / 17, None,
                torch.randn(h, d, generator=g, dtype=torch.float64) / 17, None)
    y0 = torch.randn(5, d, generator=g, dtype=torch.float64).to(dev())
    t = torch.linspace(0., 0.2, 9)
    a = odeint(f, y0, t, method='explicit_adams')
    assert not str(dict(odeint.last_stats).get('engine', '')).startswith('fused')
    b = odeint(lambda t_, y: f.forward(t_, y), y0, t, method='explicit_adams')
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# the adaptive Runge-Kutta solvers for the ODEFunc network OUTSIDE the tile kernels' box (float64, dim > 64, hidden > 128) in one launch:
# the cooperative right-hand side under the whole-call row-local kernel (round-4 review, item 7)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('method', ['dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun'])
@pytest.mark.parametrize('dim,hidden,batch,act,td,dtype', [
    (2, 50, 1, 'tanh', False, np.float64), (8, 16, 300, 'softplus', True, np.float64), (64, 128, 40, 'tanh', False, np.float64),
    (100, 200, 7, 'relu', False, np.float64), (80, 32, 1000, 'tanh', True, np.float32)])
def test_mlp_outside_the_tile_kernels_runs_in_one_launch(method, dim, hidden, batch, act, td, dtype):
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(2000 + dim + hidden)
    sc = 0.7
    Ws = [sc * rng.standard_normal((dim + (1 if td else 0), hidden)) / np.sqrt(dim), sc * rng.standard_normal((hidden, hidden)) / np.sqrt(hidden),
          sc * rng.standard_normal((hidden, dim)) / np.sqrt(hidden)]
    bs = [0.1 * rng.standard_normal(hidden), 0.1 * rng.standard_normal(hidden), 0.1 * rng.standard_normal(dim)]
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    f = rhs.MLP(*[torch.tensor(v, dtype=tdt) for pair in zip(Ws, bs) for v in pair], activation=act, time_dependent=td)
    fn = _np_mlp(Ws, bs, act, td, dtype)
    y0 = rng.standard_normal((batch, dim)).astype(dtype)
    f64 = dtype == np.float64
    tol = dict(rtol=1e-6, atol=1e-8) if f64 else dict(rtol=1e-4, atol=1e-6)
    horizon = 0.1 if method == 'adaptive_heun' else 1.0                       # (a second-order method at rtol 1e-6: a short horizon)
    for tt in (np.array([0., 0.4, 1.5]) * horizon, np.array([0., -0.4, -1.5]) * horizon):
        got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        st = dict(odeint.last_stats)
        assert st['n_launches'] == 1 and st['status'] == 0 and not str(st.get('engine', '')).startswith('device-controlled'), st
        # (tsit5: the published tableau on both sides - the reference's own is defective)
        ref, rst = O.odeint(fn, y0, tt, method=method, return_stats=True, options={'tsit5_fixed': True} if method == 'tsit5' else None, **tol)
        scale = max(1.0, np.abs(np.asarray(ref)).max())
        if f64:
            assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted), (st, rst.n_attempts, rst.n_accepted)
        assert np.abs(got.cpu().numpy().astype(np.float64) - np.asarray(ref, dtype=np.float64)).max() <= (1e-9 if f64 else 2e-4) * scale
        call = odeint(lambda t_, y: f.forward(t_, y), torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        assert str(dict(odeint.last_stats).get('engine', '')).startswith('device-controlled')
        assert float((got - call).abs().max()) <= (1e-9 if f64 else 2e-4) * scale


def test_mlp_float64_module_trains_and_evaluates_through_the_cooperative_kernel():
    """models.ODEBlock over a float64 ODEFunc: evaluation in one launch (it used to be a Python callable on the device-controlled engine) -
    also for a batch far beyond a co-resident grid (the plane-streaming whole-call kernel) and for dopri8; a single output time is served
    without a solve; nothing warns."""
    from tfdiffeq_amd import models, odeint
    torch.manual_seed(5)
    blk = models.ODEBlock(models.ODEFunc(6, 24, non_linearity='tanh'), tol=1e-6).to(dev()).double()
    x = torch.randn(50, 6, dtype=torch.float64, device=dev())
    with torch.no_grad():
        out = blk(x)
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1 and st['status'] == 0, st
    ref = odeint(lambda t_, y: blk.odefunc(t_, y), x, torch.tensor([0., 1.]), rtol=1e-6, atol=1e-6, method='dopri5')[1]
    assert float((out - ref).abs().max()) < 1e-9
    f = blk.odefunc.device_rhs()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        big = torch.randn(50000, 6, dtype=torch.float64, device=dev())        # 42 trajectories per workgroup: 1191 workgroups - the state streams through planes
        o2 = odeint(f, big, torch.tensor([0., 0.1]), rtol=1e-5, atol=1e-7, method='dopri5')
        st2 = dict(odeint.last_stats)
        assert st2['n_launches'] == 1 and st2['status'] == 0, st2
        o2c = odeint(lambda t_, y: f.forward(t_, y), big, torch.tensor([0., 0.1]), rtol=1e-5, atol=1e-7, method='dopri5')
        assert (st2['n_attempts'], st2['n_accepted']) == (odeint.last_stats['n_attempts'], odeint.last_stats['n_accepted'])
        assert float((o2 - o2c).abs().max()) < 1e-10
        o3 = odeint(f, x, torch.tensor([0., 1.]), rtol=1e-6, atol=1e-6, method='dopri8')
        assert dict(odeint.last_stats)['n_launches'] == 1
        assert float((o3[1] - out).abs().max()) < 1e-4                        # (another method at the same tolerance)
    assert sum('runs as a Python callable' in str(m.message) for m in w) == 0, [str(m.message) for m in w]
    assert torch.equal(odeint(f, x, torch.tensor([0.3]), method='dopri5')[0], x)
    # (round 6: a float64 network of this size is on the MFMA tile kernels - csrc/mi_ode_mlp64.h - at any batch size)
    huge = torch.randn(100000, 6, dtype=torch.float64, device=dev())
    odeint(f, huge, torch.tensor([0., 0.1]), rtol=1e-5, atol=1e-7, method='dopri5')
    assert dict(odeint.last_stats)['n_launches'] == 1
    # a network OUTSIDE the tile kernels' box (hidden 200) is on the cooperative kernel while an evaluation stays under ~50 M
    # multiply-adds; beyond that the callable engine (rocBLAS products) is the faster route: chosen silently
    wide = models.ODEFunc(6, 200, non_linearity='tanh').to(dev()).double().device_rhs()
    odeint(wide, x, torch.tensor([0., 0.1]), rtol=1e-5, atol=1e-7, method='dopri5')
    assert 'cooperative' in str(dict(odeint.last_stats).get('engine', ''))
    odeint(wide, huge, torch.tensor([0., 0.1]), rtol=1e-5, atol=1e-7, method='dopri5')
    assert str(dict(odeint.last_stats).get('engine', '')).startswith('device-controlled')


# ---------------------------------------------------------------------------------------------
# user device code for ONE state element (rhs.CustomCoop): systems of dimension up to 256 in one launch - a thread per element,
# the trajectory's state shared through LDS (round-4 review, item 5: "user code beyond dim 8 ... a conv stencil, a user's own dense layer")
# ---------------------------------------------------------------------------------------------
def _coop_cases():
    from tfdiffeq_amd import plugin_examples as PE
    rng = np.random.default_rng(77)
    n = 100
    ring = PE.reaction_diffusion_ring(n, 0.8, 0.05)              # (prebuilt by __graft_entry__.build(): no compilation on the GPU box)

    def ring_np(t, y):
        return 0.8 * (np.roll(y, -1, axis=-1) - 2 * y + np.roll(y, 1, axis=-1)) - 0.05 * y * y * y
    d = 48
    W = (0.5 * rng.standard_normal((d, d)) / np.sqrt(d))
    b = 0.1 * rng.standard_normal(d)
    dense = PE.swish_layer(torch.tensor(W), torch.tensor(b), 0.3, 0.2)   # a user's own dense layer, own pointwise function, forcing in t

    def dense_np(t, y):
        z = y @ W + b
        return z / (1 + np.exp(-z)) - 0.3 * y + 0.2 * np.sin(t)
    return [('ring', ring, ring_np, n), ('dense', dense, dense_np, d)]


@pytest.mark.parametrize('method', ['dopri5', 'tsit5', 'bosh3', 'dopri8', 'adaptive_heun', 'adams', 'explicit_adams'])
def test_custom_coop_systems_run_in_one_launch(method):
    from oracle import adams_numpy as OA
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint
    for name, f, fn, dim in _coop_cases():
        for batch in (1, 37):
            rng = np.random.default_rng(dim + batch)
            y0 = rng.standard_normal((batch, dim))
            if method == 'explicit_adams':
                t = np.linspace(0., 0.3, 31)
            elif method in ('adaptive_heun', 'bosh3'):
                t = np.array([0., 0.05, 0.12])
            else:
                t = np.array([0., 0.4, 1.2])
            tol = dict(rtol=1e-6, atol=1e-8)
            got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(t), method=method, **tol)
            st = dict(odeint.last_stats)
            assert st['n_launches'] == 1 and st['status'] == 0, (name, method, st)
            if method in ('adams', 'explicit_adams'):
                ref, rst = OA.odeint(fn, y0, t, method=method, return_stats=True, **tol)
                band = 1e-6 if method == 'adams' else 1e-8
            else:
                ref, rst = O.odeint(fn, y0, t, method=method, return_stats=True, options={'tsit5_fixed': True} if method == 'tsit5' else None, **tol)
                assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted), (name, method, st, rst.n_attempts, rst.n_accepted)
                band = 1e-9
            scale = max(1.0, np.abs(np.asarray(ref)).max())
            assert np.abs(got.cpu().numpy() - np.asarray(ref)).max() <= band * scale, (name, method, batch)


def test_custom_coop_limits_are_loud():
    from tfdiffeq_amd import odeint, rhs
    with pytest.raises(ValueError):
        rhs.CustomCoop(257, "k = y[i];")
    f = rhs.CustomCoop(128, "k = -y[i];")
    y0 = torch.randn(4000, 128, dtype=torch.float64, device=dev())           # 2 trajectories per workgroup: 2000 workgroups are not co-resident -
    big = odeint(f, y0, torch.tensor([0., 1.]), method='dopri5', rtol=1e-8, atol=1e-10)      # the plane-streaming whole-call kernel, still one launch
    assert dict(odeint.last_stats)['n_launches'] == 1 and float((big[-1] - y0 * np.exp(-1.0)).abs().max()) < 1e-6
    with pytest.raises(NotImplementedError):                                  # midpoint / heun: no kernel, a Python callable is needed
        odeint(f, y0[:8], torch.tensor([0., 0.5, 1.]), method='midpoint')
    g = rhs.CustomCoop(128, "k = -y[i];", torch_fn=lambda t, y: -y)
    out = odeint(g, y0[:8], torch.tensor([0., 0.5, 1.]), method='rk4')
    ref = odeint(g, y0[:8], torch.tensor([0., 0.5, 1.]), method='dopri5', rtol=1e-9, atol=1e-11)
    assert float((out - ref).abs().max()) < 5e-3                           # (two RK4 steps of 0.5 on y' = -y: 3e-4 |y| each)
    assert float((ref[-1] - y0[:8] * np.exp(-1.0)).abs().max()) < 1e-8


@pytest.mark.parametrize('method', ['euler', 'rk4'])
def test_fixed_grid_methods_on_the_cooperative_kernels(method):
    """euler / rk4 (3/8 rule) in one launch for rhs.CustomCoop and for a float64 network outside the tile kernels' box - any batch size
    (trajectories never interact on a fixed grid) - against the numpy restatement of fixed_grid.py / rk_common.py:73-81."""
    from oracle import ode_numpy as O
    from tfdiffeq_amd import odeint, rhs
    cases = [(f, fn, dim) for _, f, fn, dim in _coop_cases()]
    rng = np.random.default_rng(11)
    d, hd = 6, 40
    Ws = [0.7 * rng.standard_normal((d, hd)) / np.sqrt(d), 0.7 * rng.standard_normal((hd, hd)) / np.sqrt(hd), 0.7 * rng.standard_normal((hd, d)) / np.sqrt(hd)]
    bs = [0.1 * rng.standard_normal(hd), 0.1 * rng.standard_normal(hd), 0.1 * rng.standard_normal(d)]
    cases.append((rhs.MLP(*[torch.tensor(v) for pair in zip(Ws, bs) for v in pair], activation='tanh'), _np_mlp(Ws, bs, 'tanh', False, np.float64), d))
    for f, fn, dim in cases:
        for batch in (3, 5000):                                               # 5000 x dim 100: 2500 workgroups - fine on a fixed grid
            y0 = rng.standard_normal((batch, dim))
            for t in (np.linspace(0., 0.5, 11), -np.linspace(0., 0.05, 11)):      # (backwards the diffusion ring is anti-diffusion: a short horizon)
                got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(t), method=method)
                st = dict(odeint.last_stats)
                assert st['n_launches'] == 1 and st['status'] == 0, st
                ref = np.asarray(O.odeint(fn, y0, t, method=method))
                assert np.isfinite(ref).all() and np.abs(got.cpu().numpy() - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
