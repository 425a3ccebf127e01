"""GPU tests of the automatic lowering (tfdiffeq_amd/lower.py): the reference's callers - tests/problems.py, examples/ode_demo.py,
examples/lorenz_attractor.py, the notebook's systems, the 25 DETEST problems - handed to `odeint` as the PLAIN PYTHON CALLABLES the
reference's users write, must run on the fused kernels: one launch per call, the oracle's attempt / accept counts exactly (float64),
values inside the north star's rtol 1e-5 / atol 1e-6 of the reference-generated fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lower_cases as LC                                    # noqa: E402
from golden_util import load                                # noqa: E402

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6                                     # BASELINE.json north_star


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', LC.fixture_names())
def test_reference_fixtures_with_literal_python_callables(name):
    from tfdiffeq_amd import odeint
    d, meta = load(name)
    y0 = torch.tensor(d['y0'], device=dev())
    f = LC.literal_callable(meta, dev(), y0.dtype)
    kw = {k: meta[k] for k in ('rtol', 'atol') if meta[k] is not None}
    if meta['options']:
        kw['options'] = dict(meta['options'])
    sol = odeint(f, y0, torch.as_tensor(d['t']), method=meta['method'], **kw)
    st = dict(odeint.last_stats)
    assert st['lower']['lowered'], st
    assert tuple(sol.shape) == d['y'].shape
    f32 = d['y0'].dtype == np.float32
    ref = torch.tensor(d['y'])
    if f32:
        assert float(((sol.cpu() - ref).abs() / (1 + ref.abs())).max()) < 1e-3
    else:
        assert bool(((sol.cpu() - ref).abs() <= ATOL + RTOL * ref.abs()).all()), float((sol.cpu() - ref).abs().max())
    if meta['method'] in ('dopri5', 'bosh3', 'dopri8', 'adaptive_heun', 'euler', 'rk4', 'explicit_adams', 'fixed_adams', 'adams'):
        assert st.get('n_launches', 1) == 1, st
    if 'trace' in d.files and not f32 and meta['method'] in ('dopri5', 'bosh3', 'dopri8', 'adaptive_heun'):
        assert (st['n_attempts'], st['n_accepted']) == (len(d['trace']), int(d['trace'][:, 2].sum())), (st, len(d['trace']))
    assert st['status'] == 0


def test_the_three_test_problems_of_the_reference():
    """tests/odeint_tests.py:27-98 with tests/problems.py:13-68 as they are written: scalar states for `constant` and `sine`, the
    reshape / A @ y / reshape form for `linear`; every method of the file, rel error against the closed forms below the file's own 1e-3."""
    from scipy.linalg import expm
    from tfdiffeq_amd import odeint
    t_points = torch.tensor(np.linspace(np.float32(1.), np.float32(8.), 10).astype(np.float64))

    def problem(ode):
        if ode == 'constant':
            f = LC.ConstantODE(dev(), as_tensor=True)
            sol = 0.2 * t_points + 3.0
        elif ode == 'sine':
            f = LC.SineODE()
            t = t_points
            sol = (-0.5 * t ** 4 * torch.cos(2 * t) + 0.5 * t ** 3 * torch.sin(2 * t) + 0.25 * t ** 2 * torch.cos(2 * t) - t ** 3 + 2 * t ** 4 +
                   (np.pi - 0.25) * t ** 2)
        else:
            f = LC.LinearODE(dev())
            A = f.A.cpu().numpy()
            sol = torch.tensor(np.stack([expm(A * float(ti)) @ np.ones(10) for ti in t_points]))
            # (the reference's y_exact starts from ones at t = 0, its test from sol[0] at t = 1: the exact flow from there)
            sol = torch.tensor(np.stack([expm(A * (float(ti) - 1.0)) @ sol[0].numpy() for ti in t_points]))
        return f, sol[0].to(dev()), sol

    def rel_error(true, est):
        return float(((true - est.cpu()) / true).abs().max())
    for method in ('dopri5', 'bosh3', 'adaptive_heun', 'dopri8', 'adams'):
        for ode in ('constant', 'linear', 'sine'):
            if ode == 'sine' and method in ('bosh3', 'adaptive_heun'):
                continue                                     # "Sine test never finishes" (odeint_tests.py:53-55)
            f, y0, sol = problem(ode)
            kw = dict(rtol=1e-12, atol=1e-14) if method == 'dopri8' else {}
            y = odeint(f, y0, t_points, method=method, **kw)
            st = odeint.last_stats
            assert st['lower']['lowered'] and st.get('n_launches', 1) == 1, (method, ode, st)
            assert rel_error(sol, y) < 1e-3, (method, ode, rel_error(sol, y))
    for method in ('euler', 'midpoint', 'huen', 'rk4', 'explicit_adams'):
        f, y0, sol = problem('constant')
        y = odeint(f, y0, t_points, method=method)
        assert rel_error(sol, y) < 1e-3, method


@pytest.mark.parametrize('name', [c + i for c in 'ABCDE' for i in '12345'])
def test_detest_callables_in_one_launch(name):
    """tests/DETEST/run.py:25-60: every problem wrapped in the harness' own NFEDiffEq counter, dopri5 at tol 1e-3 and 1e-6 - one launch,
    the reference's NFE / attempts / accepts (tests/golden/fn_detest.npz, captured from the reference) and its y(20)."""
    from tfdiffeq_amd import odeint
    from oracle import detest_problems as DP
    d, meta = load('fn_detest')
    f, y0 = DP.problem(name, torch, like=torch.zeros(1, device=dev(), dtype=torch.float64))
    diffeq = LC.NFEDiffEq(f)
    tgrid = torch.tensor([0., DP.T_END], dtype=torch.float64)
    for row, tol in zip(d[name + '_runs'], meta['tols']):
        diffeq.nfe = 0
        est = odeint(diffeq, y0, tgrid, atol=tol, rtol=tol, method='dopri5')
        st = dict(odeint.last_stats)
        assert st['lower']['lowered'] and st['n_launches'] == 1, st
        ref = d['%s_y20_tol%r' % (name, tol)]
        assert (st['n_attempts'], st['n_accepted']) == (int(row[2]), int(row[3])), (name, tol, st, row)
        assert diffeq.nfe == int(row[1]), (diffeq.nfe, row)          # the harness' counter, credited from the kernel's evaluations
        np.testing.assert_allclose(est[1].cpu().numpy(), ref, rtol=RTOL, atol=ATOL)


NOTEBOOK = ['second_order', 'oscilation', 'jagged_oscilation', 'nonlinear_damping', 'predator_prey', 'limited_predator_prey', 'periodic_sinusodial',
            'linear2d_1', 'linear2d_2', 'linear2d_3', 'linear2d_4', 'linear2d_5', 'parabolic', 'nonlinear_system1', 'nonlinear_predator_prey',
            'spiral_sink', 'jacobian_spiral_sink', 'jacobian_predator_prey', 'spiral_cycle', 'force_pendulum', 'duffing', 'rossler']


@pytest.mark.parametrize('name', NOTEBOOK)
def test_notebook_systems_against_the_cpu_restatement(name):
    """examples/ode_usage.ipynb: the callable as the notebook writes it, its y0 and its 1000 (2500) output times, dopri5 at the default
    tolerances - one launch on the GPU; the torch-CPU restatement of the reference's Dopri5 path (the oracle) runs the SAME callable on
    CPU tensors: same attempts, same accepts, the 1000 outputs inside rtol 1e-5 / atol 1e-6."""
    import reference_systems as RS
    from oracle import ode_torch_cpu as TC
    from tfdiffeq_amd import odeint
    s = RS.systems(dev())[name]
    sol = odeint(s['func'], s['y0'], s['t'])
    st = dict(odeint.last_stats)
    assert st['lower']['lowered'] and st['n_launches'] == 1 and st['status'] == 0, st
    c = RS.systems('cpu')[name]
    ref, rst = TC.odeint_dopri5(c['func'], c['y0'], c['t'])
    assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted), (st, rst.n_attempts, rst.n_accepted)
    assert bool(((sol.cpu() - ref).abs() <= ATOL + RTOL * ref.abs()).all()), float((sol.cpu() - ref).abs().max())


def test_lorenz_attractor_example_output_path():
    """examples/lorenz_attractor.py:40-50 - ONE trajectory, t = range(0, 100, 0.01): 10 000 output times, thousands of dependent attempts
    on one lane (the regime the reference publishes its 47.6 s for).  The first 20 time units (2000 outputs, beyond what the kernel keeps
    of t in LDS) against the oracle's dense output: same attempts, outputs inside the band while the trajectories have not separated."""
    import reference_systems as RS
    from oracle import ode_torch_cpu as TC
    from tfdiffeq_amd import odeint
    s = RS.systems(dev())['lorenz']
    t = s['t'][:2001]
    sol = odeint(s['func'], s['y0'], t)
    st = dict(odeint.last_stats)
    assert st['lower']['lowered'] and st['n_launches'] == 1 and st['status'] == 0, st
    c = RS.systems('cpu')['lorenz']
    ref, rst = TC.odeint_dopri5(c['func'], c['y0'], t)
    assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted)
    head = slice(0, 1001)                                     # t <= 10: the two solutions agree far inside the band (chaos amplifies
    assert bool(((sol.cpu()[head] - ref[head]).abs() <= ATOL + RTOL * ref[head].abs()).all())     # last-bit differences by e^(0.9 t) later)
    assert float((sol.cpu() - ref).abs().max()) < 1e-3
    full = odeint(s['func'], s['y0'], s['t'])                 # the whole published workload: 10 000 outputs, one launch
    st = dict(odeint.last_stats)
    assert tuple(full.shape) == (10000, 3) and st['n_launches'] == 1 and st['status'] == 0 and bool(torch.isfinite(full).all())
    assert torch.equal(full[:2001], sol)                      # (the same arithmetic whatever the number of outputs)


@pytest.mark.parametrize('name', ['ring_100', 'swish_48', 'tdep', 'lorenz_batched', 'demo_spiral_batch', 'demo_net_f64', 'demo_net_f32',
                                  'linear_128_yW', 'linear_24_yW_bias', 'linear_16_Ay', 'linear_32_module', 'mlp_64_128_tanh_f32',
                                  'mlp_16_32_softplus_f64', 'mlp_8_24_relu2_f64'])
@pytest.mark.parametrize('method', ['dopri5', 'rk4'])
def test_lowered_against_the_same_callable_on_the_callable_engine(name, method):
    """Generated code (a trajectory per thread, an element per thread) and the catalogue routes (MFMA tile kernels, cooperative MLP) against
    the very same Python callable evaluated by torch between library kernels (options={'lower': False})."""
    from tfdiffeq_amd import odeint
    f, y0, kind = LC.CASES[name](dev())
    t = torch.tensor([0., 0.3, 0.8] if method == 'dopri5' else np.linspace(0., 0.5, 9))
    kw = dict(rtol=1e-6, atol=1e-8) if method == 'dopri5' else {}
    with torch.no_grad():
        sol = odeint(f, y0, t, method=method, **kw)
        st = dict(odeint.last_stats)
        ref = odeint(f, y0, t, method=method, options={'lower': False}, **kw)
        rst = dict(odeint.last_stats)
    assert st['lower']['lowered'] and st['lower']['kind'] == kind and st.get('n_launches', 1) == 1, st
    assert 'lower' not in rst or not rst['lower'].get('lowered')
    f64 = y0.dtype == torch.float64
    tol = 1e-9 if f64 else 2e-4
    assert float(((sol - ref).abs() / (1 + ref.abs())).max()) < tol, float((sol - ref).abs().max())
    if method == 'dopri5' and f64:
        assert (st['n_attempts'], st['n_accepted']) == (rst['n_attempts'], rst['n_accepted']), (st, rst)


def test_constants_are_fresh_on_every_call():
    """The compiled code depends on the callable's structure only: floats and tensors it closes over are read at every call (a module
    trained in place, a coefficient changed between calls) - and the second call compiles nothing."""
    from tfdiffeq_amd import odeint, lower as L

    f = LC.Decay(dev())
    y0 = torch.ones(7, 3, dtype=torch.float64, device=dev())
    t = torch.tensor([0., 1.])
    a = odeint(f, y0, t)[1]
    n_prog = len(L._PROGRAMS)
    np.testing.assert_allclose(a.cpu().numpy(), np.exp(-0.5 * np.array([1., 2., 3.]))[None].repeat(7, 0), rtol=1e-5)
    f.rate = 0.25
    f.w.mul_(2.0)
    b = odeint(f, y0, t)[1]
    assert len(L._PROGRAMS) == n_prog and odeint.last_stats['lower']['lowered']
    np.testing.assert_allclose(b.cpu().numpy(), np.exp(-0.5 * np.array([1., 2., 3.]))[None].repeat(7, 0), rtol=1e-5)
    f.rate = 1.0
    c = odeint(f, y0, t)[1]
    np.testing.assert_allclose(c.cpu().numpy(), np.exp(-2.0 * np.array([1., 2., 3.]))[None].repeat(7, 0), rtol=1e-5)


def test_callables_outside_the_op_set_say_so_and_still_run():
    from tfdiffeq_amd import odeint
    y0 = torch.ones(5, 3, dtype=torch.float64, device=dev())
    t = torch.tensor([0., 0.5])
    with pytest.warns(UserWarning, match='not lowered onto the fused kernels'):
        sol = odeint(lambda t_, y: -torch.cumsum(y, -1) * 0.1, y0, t)
    st = odeint.last_stats
    assert st['lower'] == {'lowered': False, 'why': 'operation `cumsum` is outside the op set'} and bool(torch.isfinite(sol).all())
    with pytest.raises(ValueError, match='cannot be lowered'):
        odeint(lambda t_, y: -torch.cumsum(y, -1), y0, t, method='dopri5', options={'lower': True})
    log = []

    def impure(t_, y):
        log.append(1)
        return -y
    counter = [0]

    def counting(t_, y):
        counter[0] += 1
        return -y
    sol = odeint(counting, y0, torch.tensor(np.linspace(0., 5., 40)))
    st = odeint.last_stats
    assert 'changed its own Python state' in st['lower']['why']
    # its side effects keep happening at EVERY evaluation: no hipGraph recording for a callable with observable state of its own
    assert counter[0] >= st['nfe'] and not st.get('replays')
    np.testing.assert_allclose(sol[-1].cpu().numpy(), np.exp(-5.0) * np.ones((5, 3)), rtol=1e-5)


def test_reverse_time_and_reshaped_states():
    """A decreasing t (misc.py:318-321) through a lowered callable; a [2, 3, 5] state integrated as ONE system of 30."""
    from tfdiffeq_amd import odeint
    f = LC.SineODE()
    y0 = torch.tensor(2.5, dtype=torch.float64, device=dev())
    fwd = odeint(f, y0, torch.tensor([1., 2., 3.]))
    back = odeint(f, fwd[-1], torch.tensor([3., 2., 1.]))
    assert odeint.last_stats['lower']['lowered']
    np.testing.assert_allclose(back[-1].item(), 2.5, rtol=1e-5)
    f3, y3, _ = LC.CASES['state_2x3x5'](dev())
    sol = odeint(f3, y3, torch.tensor([0., 0.7]))
    st = dict(odeint.last_stats)
    assert st['lower']['lowered'] and st['lower']['dim'] == 30 and st['lower']['batch_axes'] == 0 and tuple(sol.shape) == (2, 2, 3, 5)
    ref = odeint(f3, y3, torch.tensor([0., 0.7]), method='dopri5', options={'lower': False})
    assert float((sol - ref).abs().max()) < 1e-9


def test_ode_demo_trains_with_its_forward_on_a_generated_kernel():
    """This repo's examples/ode_demo.py (= the reference's examples/ode_demo.py:115-129 `net(y ** 3)`): odeint_adjoint's forward pass
    runs the traced network in one launch; the loss falls as before."""
    sys.path.insert(0, os.path.join(LC.ROOT, 'examples'))
    import ode_demo
    from tfdiffeq_amd import odeint, odeint_adjoint
    losses = ode_demo.main(['--niters', '25', '--test_freq', '25', '--data_size', '200'])
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
    f, y0, _ = LC.CASES['demo_net_f64'](dev())
    t = torch.tensor(np.linspace(0., 0.3, 10))
    pred = odeint_adjoint(f, y0, t, method='dopri5')
    fwd = dict(odeint.last_stats)
    assert fwd['lower']['lowered'] and fwd['lower']['kind'] == 'rowlocal' and fwd['n_launches'] == 1, fwd
    pred.abs().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in f.parameters())
    assert 'forward' in odeint_adjoint.last_backward_stats and odeint_adjoint.last_backward_stats['forward']['lowered']


def test_plain_callable_over_trainable_tensors_is_differentiated_without_a_probe():
    """`odeint(lambda t, y: net(y ** 3) * scale, y0, t)` under grad mode, y0 not requiring grad: the tensors the callable closes over are
    read off its trace (no probe evaluation with autograd per call); the ones that require grad make the call an adjoint solve - forward
    on the generated kernel - and receive their gradients; with none of them requiring grad the call is the plain one-launch solve."""
    from tfdiffeq_amd import odeint, odeint_adjoint
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3)).double().to(dev())
    scale = torch.tensor(0.7, dtype=torch.float64, device=dev(), requires_grad=True)
    f = lambda t, y: net(y ** 3) * scale                    # noqa: E731
    y0 = torch.randn(12, 3, dtype=torch.float64, device=dev())
    t = torch.tensor([0., 0.5])
    out = odeint(f, y0, t, rtol=1e-8, atol=1e-10)
    assert out.requires_grad
    out[-1].pow(2).sum().backward()
    assert odeint_adjoint.last_backward_stats['forward']['lowered']
    got = [p.grad.clone() for p in list(net.parameters()) + [scale]]
    for p in list(net.parameters()) + [scale]:
        p.grad = None
    y = y0.clone()
    h = 0.5 / 200
    for _ in range(200):                                    # autograd through a fine RK4 integration of the same callable
        k1 = f(0, y); k2 = f(0, y + 0.5 * h * k1); k3 = f(0, y + 0.5 * h * k2); k4 = f(0, y + h * k3)
        y = y + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
    y.pow(2).sum().backward()
    for g, p in zip(got, list(net.parameters()) + [scale]):
        assert float((g - p.grad).abs().max()) <= 1e-6 * max(1.0, float(p.grad.abs().max()))
    for p in list(net.parameters()) + [scale]:
        p.requires_grad_(False)
    out = odeint(f, y0, t, rtol=1e-8, atol=1e-10)
    assert not out.requires_grad and odeint.last_stats['lower']['lowered'] and odeint.last_stats['n_launches'] == 1


def test_tuple_states_of_the_reference_api_tests():
    """tests/api_tests.py:26-34: `tuple_f = lambda t, y: (f(t, y[0]), f(t, y[1]))`, `tuple_y0 = (y0, y0)` - a tuple state whose components
    follow the same function is traced per component and integrated in ONE launch with one error ratio per component (rhs.PerComponent);
    the reference-generated tuple fixture: its attempt / accept counts exactly.  Components that interact keep their Python loop."""
    from tfdiffeq_amd import odeint
    d, meta = load('run_constant_dopri5_tuple')
    f = LC.ConstantODE(dev())
    tuple_f = lambda t, y: (f(t, y[0]), f(t, y[1]))         # noqa: E731
    y0 = (torch.tensor(d['y0_0'], device=dev()), torch.tensor(d['y0_1'], device=dev()))
    sol = odeint(tuple_f, y0, torch.as_tensor(d['t']), method='dopri5')
    st = dict(odeint.last_stats)
    assert isinstance(sol, tuple) and len(sol) == 2 and st['lower']['lowered'] and st['lower']['components'] == 2 and st['n_launches'] == 1, st
    assert (st['n_attempts'], st['n_accepted']) == (len(d['trace']), int(d['trace'][:, 2].sum()))
    for k in range(2):
        ref = torch.tensor(d['y_%d' % k])
        assert tuple(sol[k].shape) == tuple(ref.shape) and bool(((sol[k].cpu() - ref).abs() <= ATOL + RTOL * ref.abs()).all())
    # components of different batch shapes, a time-dependent factor shared by both
    g = lambda t, y: (torch.cos(t) * y[0] - y[0] ** 3, torch.cos(t) * y[1] - y[1] ** 3)     # noqa: E731
    ya = torch.randn(40, 2, dtype=torch.float64, device=dev(), generator=None)
    yb = torch.randn(7, 2, dtype=torch.float64, device=dev())
    tt = torch.tensor([0., 0.5, 1.5])
    sol = odeint(g, (ya, yb), tt, method='dopri5')
    st = dict(odeint.last_stats)
    ref = odeint(g, (ya, yb), tt, method='dopri5', options={'lower': False})
    rst = dict(odeint.last_stats)
    assert st['lower']['lowered'] and st['n_launches'] == 1 and (st['n_attempts'], st['n_accepted']) == (rst['n_attempts'], rst['n_accepted'])
    assert all(float((a - b).abs().max()) < 1e-10 for a, b in zip(sol, ref))
    # interacting components: not one function per component - the callable engine, and the stats say why
    h = lambda t, y: (y[1], -y[0])                          # noqa: E731
    sol = odeint(h, (ya, ya.clone()), tt, method='dopri5')
    assert 'interact' in odeint.last_stats['lower']['why']
    np.testing.assert_allclose(sol[0][-1].cpu().numpy(), (ya * np.cos(1.5) + ya * np.sin(1.5)).cpu().numpy(), rtol=1e-5, atol=1e-6)
