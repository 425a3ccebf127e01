"""GPU tests of tuple states on the fused engine (mi_ode_desc.n_segments: components packed into one buffer with a segment
table; per-component error ratios, all <= 1 to accept, python max() for the step size and over the initial-step norms -
odeint.py:28-81, misc.py:183-287, dopri5.py:103-121).  Checkers: the oracle (numpy restatement, tuple-aware), the
reference's own tuple fixture, and the plane-kernel engine running the same tuple."""
import numpy as np
import pytest
import torch

from oracle import ode_numpy as O
from tests.bands import assert_f32
from tests.golden_util import load

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def _lorenz_np(t, y):
    return np.stack([10. * (y[..., 1] - y[..., 0]), y[..., 0] * (28. - y[..., 2]) - y[..., 1], y[..., 0] * y[..., 1] - 8. / 3. * y[..., 2]], axis=-1)


@pytest.mark.parametrize('dtype,method', [(np.float64, 'dopri5'), (np.float32, 'dopri5'), (np.float64, 'bosh3'), (np.float64, 'dopri8')])
def test_tuple_of_lorenz_states_runs_as_one_launch(dtype, method):
    """Three components of different shapes ([300, 3], [5, 7, 3], [1000, 3]); each has its own error ratio.  Same attempt
    sequence as the oracle and the plane-kernel engine."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(7)
    comps = [(np.array([1., 1., 1.]) + s * rng.standard_normal(shape)).astype(dtype)
             for s, shape in ((1e-2, (300, 3)), (3.0, (5, 7, 3)), (1e-1, (1000, 3)))]
    t = np.array([0., 0.15, 0.4])
    f64 = dtype == np.float64
    tol = dict(rtol=1e-7, atol=1e-9) if f64 else dict(rtol=1e-4, atol=1e-6)
    f = rhs.PerComponent(rhs.Lorenz())
    y0 = tuple(torch.tensor(c, device=dev()) for c in comps)
    sol = odeint(f, y0, torch.tensor(t), method=method, **tol)
    st = dict(odeint.last_stats)
    assert st.get('components') == 3 and st['n_launches'] == 1 and st['status'] == 0, st
    planes = odeint(f, y0, torch.tensor(t), method=method, options={'force_plane_kernels': True}, **tol)
    ps = dict(odeint.last_stats)
    assert ps.get('engine') == 'plane kernels'
    ref, rst = O.odeint(lambda t_, ys: tuple(_lorenz_np(t_, y) for y in ys), tuple(comps), t, method=method, return_stats=True, **tol)
    assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted) == (ps['n_attempts'], ps['n_accepted'])
    for got, pl, rf, c in zip(sol, planes, ref, comps):
        assert tuple(got.shape) == (3,) + c.shape and got.dtype == (torch.float64 if f64 else torch.float32)
        if f64:
            scale = np.abs(rf).max()
            assert np.abs(got.cpu().numpy() - rf).max() < 1e-9 * scale
            assert float((got - pl).abs().max()) < 1e-9 * scale
        else:
            comp = 'comp%dx%d' % (c.size // 3, 3)
            assert_f32(got.cpu(), rf, 'tuple_lorenz/%s/%s/fused_vs_oracle' % (method, comp))
            assert_f32(got.cpu(), pl.cpu(), 'tuple_lorenz/%s/%s/fused_vs_planes' % (method, comp))


def test_a_component_with_a_large_error_decides():
    """The accept test is per component: a stiff-ish component (far from the attractor) forces rejections / small steps that
    the same system WITHOUT that component does not need - and a single concatenated tensor (one pooled tolerance) steps
    differently from the tuple (the reference's F3 semantics: tol is one scalar per component)."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(8)
    a = np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((400, 3))
    b = np.array([30., -40., 90.]) + rng.standard_normal((50, 3))
    t = torch.tensor([0., 0.3])
    kw = dict(rtol=1e-6, atol=1e-9, method='dopri5')
    f = rhs.PerComponent(rhs.Lorenz())
    both = odeint(f, (torch.tensor(a, device=dev()), torch.tensor(b, device=dev())), t, **kw)
    n_both = odeint.last_stats['n_attempts']
    alone = odeint(rhs.Lorenz(), torch.tensor(a, device=dev()), t, **kw)
    n_alone = odeint.last_stats['n_attempts']
    assert n_both > n_alone
    ref, rst = O.odeint(lambda t_, ys: tuple(_lorenz_np(t_, y) for y in ys), (a, b), np.array([0., 0.3]), rtol=1e-6, atol=1e-9,
                        method='dopri5', return_stats=True)
    assert n_both == rst.n_attempts
    # (the two pow() implementations move dt by ulps; 20 steps of a chaotic system amplify that)
    assert np.abs(both[0].cpu().numpy() - ref[0]).max() < 1e-4 and np.abs(both[1].cpu().numpy() - ref[1]).max() < 1e-4
    assert float((both[0][1] - alone[1]).abs().max()) < 1e-3           # same trajectories, different step sequences (tolerance 1e-6, chaotic)


def test_reference_tuple_fixture_through_the_fused_engine():
    """tests/golden/run_constant_dopri5_tuple.npz: the reference's own tuple run (tests/problems.py `constant`, two scalar
    components) - here as a one-dimensional CustomRowLocal system lifted with PerComponent: one launch, the reference's step
    trace and values."""
    from tfdiffeq_amd import odeint, rhs
    d, meta = load('run_constant_dopri5_tuple')
    p = meta['rhs_params']
    const = rhs.CustomRowLocal(1, 'const T d_ = y[0] - (p[0] * t + p[1]); k[0] = p[0] + d_ * d_ * d_ * d_ * d_;', params=[p['a'], p['b']],
                               torch_fn=lambda t_, y: p['a'] + (y - (p['a'] * t_ + p['b'])) ** 5)
    y0 = tuple(torch.tensor(np.asarray(d['y0_%d' % i], dtype=np.float64).reshape(1, 1), device=dev()) for i in range(2))
    sol = odeint(rhs.PerComponent(const), y0, torch.tensor(np.asarray(d['t'], dtype=np.float64)), method='dopri5')
    st = dict(odeint.last_stats)
    assert st.get('components') == 2 and st['n_launches'] == 1, st
    trace = d['trace']
    assert st['n_attempts'] == trace.shape[0] and st['n_accepted'] == int(trace[:, 2].sum())
    assert abs(st['dt'] - trace[-1, 3]) <= 1e-9 * abs(trace[-1, 3])
    for i in range(2):
        np.testing.assert_allclose(sol[i].cpu().numpy().reshape(-1), d['y_%d' % i], rtol=1e-9, atol=1e-12)


def test_per_component_tolerances_and_the_pooled_tsit5_ratio():
    """dopri5.py:60-61 accepts one (rtol, atol) pair per component; tsit5 pools all components into ONE mean with scalar
    tolerances (tsit5.py:126-138).  Both on the segmented engine, against the oracle and the plane-kernel engine."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(11)
    comps = [np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((200, 3)), np.array([5., -3., 20.]) + rng.standard_normal((40, 3))]
    y0 = tuple(torch.tensor(c, device=dev()) for c in comps)
    f = rhs.PerComponent(rhs.Lorenz())
    t = np.array([0., 0.2])
    fn = lambda t_, ys: tuple(_lorenz_np(t_, y) for y in ys)          # noqa: E731
    kw = dict(method='dopri5', rtol=[1e-7, 1e-4], atol=[1e-9, 1e-6])
    sol = odeint(f, y0, torch.tensor(t), **kw)
    st = dict(odeint.last_stats)
    assert st.get('components') == 2 and st['n_launches'] == 1
    ref, rst = O.odeint(fn, tuple(comps), t, return_stats=True, **kw)
    assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted)
    uniform = odeint(f, y0, torch.tensor(t), method='dopri5', rtol=1e-4, atol=1e-6)
    assert odeint.last_stats['n_attempts'] < st['n_attempts']          # the tight component costs steps
    for got, rf in zip(sol, ref):
        assert np.abs(got.cpu().numpy() - rf).max() < 1e-5
    del uniform
    sol = odeint(f, y0, torch.tensor(t), method='tsit5', rtol=1e-6, atol=1e-9)
    st = dict(odeint.last_stats)
    assert st.get('components') == 2 and st['n_launches'] == 1
    planes = odeint(f, y0, torch.tensor(t), method='tsit5', rtol=1e-6, atol=1e-9, options={'force_plane_kernels': True})
    ps = dict(odeint.last_stats)
    assert ps.get('engine') == 'plane kernels' and (st['n_attempts'], st['n_accepted']) == (ps['n_attempts'], ps['n_accepted'])
    for got, pl in zip(sol, planes):
        assert float((got - pl).abs().max()) < 1e-6


def test_what_the_segmented_engine_does_not_take_stays_generic():
    from tfdiffeq_amd import odeint, rhs
    y = tuple(torch.randn(10, 3, dtype=torch.float64, device=dev()) for _ in range(2))
    f = rhs.PerComponent(rhs.Lorenz())
    t = torch.tensor([0., 0.05])
    odeint(f, y + tuple(torch.randn(4, 3, dtype=torch.float64, device=dev()) for _ in range(7)), t, method='dopri5')   # 9 components
    assert str(odeint.last_stats.get('engine')).startswith(('plane kernels', 'device-controlled'))
    odeint(rhs.PerComponent(rhs.Linear(torch.eye(16, dtype=torch.float64, device=dev()))),
           tuple(torch.randn(10, 16, dtype=torch.float64, device=dev()) for _ in range(2)), t, method='dopri5')       # not a row-local system
    assert str(odeint.last_stats.get('engine')).startswith(('plane kernels', 'device-controlled'))
    with pytest.raises(TypeError):
        rhs.PerComponent(lambda t_, y_: y_)


@pytest.mark.parametrize('method', ['euler', 'rk4'])
def test_fixed_grid_tuple_state_on_the_one_launch_kernel(method):
    """No norms on a fixed grid: the components share one buffer; bit-exact against the oracle (elementwise arithmetic)."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(9)
    comps = [np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal(shape) for shape in ((40, 3), (3, 5, 3))]
    t = np.linspace(0., 0.2, 21)
    sol = odeint(rhs.PerComponent(rhs.Lorenz()), tuple(torch.tensor(c, device=dev()) for c in comps), torch.tensor(t), method=method)
    st = dict(odeint.last_stats)
    assert st.get('components') == 2 and st['n_launches'] == 1, st
    ref = O.odeint(lambda t_, ys: tuple(_lorenz_np(t_, y) for y in ys), tuple(comps), t, method=method)
    for got, rf, c in zip(sol, ref, comps):
        assert tuple(got.shape) == (21,) + c.shape
        assert np.abs(got.cpu().numpy() - rf).max() < 1e-12


@pytest.mark.parametrize('method', ['dopri5', 'tsit5'])
def test_large_tuple_state_on_the_plane_streaming_kernel(method):
    """More rows than one trajectory per thread keeps co-resident (three components, 100 000 / 50 000 / 20 001 trajectories): the
    workgroups of the plane-streaming whole-call kernel are dealt to the components in proportion to their rows, every component
    keeps its own error ratio.  One launch; the attempt sequence of the oracle and of the plane-kernel engine."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(17)
    comps = [np.array([1., 1., 1.]) + s * rng.standard_normal(shape) for s, shape in ((1e-2, (100000, 3)), (1.0, (50000, 3)), (1e-1, (20001, 3)))]
    t = np.array([0., 0.1, 0.25])
    tol = dict(rtol=1e-6, atol=1e-9)
    f = rhs.PerComponent(rhs.Lorenz())
    y0 = tuple(torch.tensor(c, device=dev()) for c in comps)
    sol = odeint(f, y0, torch.tensor(t), method=method, **tol)
    st = dict(odeint.last_stats)
    assert st.get('components') == 3 and st['n_launches'] == 1 and st['status'] == 0, st
    planes = odeint(f, y0, torch.tensor(t), method=method, options={'force_plane_kernels': True}, **tol)
    ps = dict(odeint.last_stats)
    assert ps.get('engine') == 'plane kernels' and (st['n_attempts'], st['n_accepted']) == (ps['n_attempts'], ps['n_accepted']), (st, ps)
    if method == 'dopri5':
        ref, rst = O.odeint(lambda t_, ys: tuple(_lorenz_np(t_, y) for y in ys), tuple(comps), t, method=method, return_stats=True, **tol)
        assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted)
    for k, (got, pl, c) in enumerate(zip(sol, planes, comps)):
        assert tuple(got.shape) == (3,) + c.shape
        scale = max(1.0, float(pl.abs().max()))
        assert float((got - pl).abs().max()) < 1e-9 * scale, k
        if method == 'dopri5':
            assert np.abs(got.cpu().numpy() - ref[k]).max() < 1e-9 * scale, k
