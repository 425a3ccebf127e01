"""The fixed-grid Adams family ('explicit_adams', 'fixed_adams'; fixed_adams.py:152-212) as ONE launch for the row-local catalogue
systems (csrc/mi_ode_adams.h) - against the fixtures captured from the reference, the numpy oracle and the per-step host loop over
plane kernels."""
import numpy as np
import pytest
import torch

from oracle import adams_numpy as OA
from tests.golden_util import load
from tests.rhs_util import device_rhs

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['run_spiral_b64_explicit_adams', 'run_spiral_b64_fixed_adams'])
def test_reference_fixtures_in_one_launch(name):
    from tfdiffeq_amd import odeint
    d, meta = load(name)
    f = device_rhs(meta['rhs'], meta['rhs_params'])
    kw = {k: meta[k] for k in ('rtol', 'atol') if meta.get(k) is not None}
    if meta.get('options'):
        kw['options'] = dict(meta['options'])
    y0 = torch.tensor(d['y0'], device=dev())
    sol = odeint(f, y0, torch.as_tensor(d['t']), method=meta['method'], **kw)
    st = dict(odeint.last_stats)
    assert st.get('engine', '').startswith('fused multistep') and st['n_launches'] == 1 and st['status'] == 0, st
    assert np.abs(sol.cpu().numpy() - d['y']).max() <= 1e-12 * max(1.0, np.abs(d['y']).max())
    # the same call on the per-step loop over plane kernels (the path every other right-hand side takes)
    loop = odeint(f, y0, torch.as_tensor(d['t']), method=meta['method'], **dict(kw, options=dict(kw.get('options', {}), fusion='stage')))
    assert dict(odeint.last_stats).get('engine', '') != 'fused multistep kernel (one launch)'
    assert float((sol - loop).abs().max()) <= 1e-12 * max(1.0, float(loop.abs().max()))


def _lv_np(t, y):
    u, v = y[..., 0], y[..., 1]
    return np.stack([1.5 * u - 1.0 * u * v, -3.0 * v + 1.0 * u * v], axis=-1)


def _lorenz_np(t, y):
    return np.stack([10. * (y[..., 1] - y[..., 0]), y[..., 0] * (28. - y[..., 2]) - y[..., 1], y[..., 0] * y[..., 1] - 8. / 3. * y[..., 2]], axis=-1)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('method', ['explicit_adams', 'fixed_adams'])
@pytest.mark.parametrize('problem,batch', [('lv', 1), ('lv', 5000), ('lorenz', 300), ('lorenz', 70000)])
def test_against_the_numpy_oracle(problem, batch, method, dtype):
    """Single and many workgroups (the implicit solver's convergence test is ONE decision for the whole batch: it crosses
    workgroups through the grid hand-off), both dtypes, forward and reversed time."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(5)
    if problem == 'lv':
        f, fn, y0 = rhs.LotkaVolterra(1.5, 1.0, 3.0, 1.0), _lv_np, 1.0 + 0.5 * rng.uniform(size=(batch, 2))
        t = np.linspace(0., 1.0, 41)
    else:
        f, fn, y0 = rhs.Lorenz(), _lorenz_np, np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((batch, 3))
        t = np.linspace(0., 0.1, 41)
    y0 = y0.astype(dtype)
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == np.float64 else dict(rtol=1e-4, atol=1e-6)
    n_check = min(batch, 512)                                    # (the oracle is a Python loop: a slice of the batch is enough
    for tt in (t, -t):                                           #  - unless the convergence decisions differ, which the counts show)
        fo = fn if tt[-1] > 0 else (lambda t_, y_: -fn(-t_, y_))
        sol = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method=method, **tol)
        st = dict(odeint.last_stats)
        assert st.get('engine', '').startswith('fused multistep') and st['n_launches'] == 1 and st['status'] == 0, st
        solver = OA.FixedAdams(lambda t_, ys: (fo(t_, ys[0]),), (y0,), implicit=(method == 'fixed_adams'), **tol)
        ref = solver.integrate(np.abs(tt).astype(dtype))[0]
        assert st['n_rejected'] == solver.n_not_converged, (st, solver.n_not_converged)
        band = 1e-11 if dtype == np.float64 else 2e-5
        got = sol.cpu().numpy()
        assert np.abs(got[:, :n_check] - ref[:, :n_check]).max() <= band * max(1.0, np.abs(ref).max()), (problem, batch, method, dtype)


def test_the_corrector_that_does_not_converge_is_reported_like_the_reference(capfd):
    """max_iters = 1 with a tight tolerance: the reference prints a warning per step and drops its oldest history entry
    (fixed_adams.py:197-200); the kernel counts the steps (stats.n_rejected), the solver prints the same warnings."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(6)
    y0 = 1.0 + 0.5 * rng.uniform(size=(300, 2))
    t = np.linspace(0., 1.0, 21)
    opts = {'max_iters': 1}
    sol = odeint(rhs.LotkaVolterra(1.5, 1.0, 3.0, 1.0), torch.tensor(y0, device=dev()), torch.tensor(t), method='fixed_adams',
                 rtol=1e-13, atol=1e-15, options=opts)
    st = dict(odeint.last_stats)
    solver = OA.FixedAdams(lambda t_, ys: (_lv_np(t_, ys[0]),), (y0,), implicit=True, rtol=1e-13, atol=1e-15, max_iters=1)
    ref = solver.integrate(t)[0]
    assert solver.n_not_converged > 0 and st['n_rejected'] == solver.n_not_converged, (st, solver.n_not_converged)
    assert capfd.readouterr().err.count('Functional iteration did not converge') >= st['n_rejected']
    assert np.abs(sol.cpu().numpy() - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


# ---------------------------------------------------------------------------------------------
# 'adams': the variable-step, variable-order solver (adams.py:66-211) in one launch (csrc/mi_ode_adams_vc.h)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('problem,batch', [('lv', 1), ('lv', 5000), ('lorenz', 300), ('lorenz', 70000), ('spiral', 64)])
def test_variable_order_adams_in_one_launch(problem, batch, dtype):
    """Against the numpy oracle (same attempt / accept sequence: the orders, the float32 g vector, the predictor-valued state are the
    reference's) and against the per-step host loop over plane kernels; one and many workgroups; both dtypes; reversed time."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(9)
    if problem == 'lv':
        f, fn, y0 = rhs.LotkaVolterra(1.5, 1.0, 3.0, 1.0), _lv_np, 1.0 + 0.5 * rng.uniform(size=(batch, 2))
        t = np.linspace(0., 2.0, 9)
    elif problem == 'lorenz':
        f, fn, y0 = rhs.Lorenz(), _lorenz_np, np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((batch, 3))
        t = np.linspace(0., 0.5, 6)
    else:
        Am = np.array([[-0.1, 2.0], [-2.0, -0.1]])
        f, fn, y0 = rhs.CubicLinear(torch.tensor(Am)), (lambda t_, y: (y ** 3) @ Am), np.tile(np.array([[2., 0.]]), (batch, 1))
        t = np.linspace(0., 5.0, 20) if dtype == np.float64 else np.linspace(0., 1.0, 5)   # (float32: before rounding forks the sequence)
    y0 = y0.astype(dtype)
    tol = dict(rtol=1e-6, atol=1e-8) if dtype == np.float64 else dict(rtol=1e-4, atol=1e-6)
    # g is rounded to float32 every step (adams.py:34): a last-bit difference in dt (the error norm's summation order) can flip a bit of
    # g, i.e. perturb the step by 1e-8 relative - the band is that, not the float64 roundoff
    band = 1e-6 if dtype == np.float64 else 2e-4
    for tt in (t, -t) if problem != 'spiral' else (t,):
        got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method='adams', **tol)
        st = dict(odeint.last_stats)
        assert st.get('engine', '').startswith('fused variable-order Adams') and st['n_launches'] == 1 and st['status'] == 0, st
        scale = max(1.0, float(got.abs().max()))
        if dtype == np.float64:                                            # (the oracle reverses f itself, misc.py:318-321)
            ref, rst = OA.odeint(fn, y0, tt, method='adams', return_stats=True, **tol)
            n_acc = int(sum(1 for r in rst.trace if r[3] > 0))
            if problem == 'spiral':        # ~280 attempts: a sequence that forks on a last-bit difference of one error ratio (64 equal rows
                # summed in another order) - the slack the reference-fixture test allows the method (test_gpu_parity.py)
                assert abs(st['n_attempts'] - len(rst.trace)) <= max(2, len(rst.trace) // 20) and abs(st['n_accepted'] - n_acc) <= max(2, n_acc // 20)
                assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * scale
            else:
                assert (st['n_attempts'], st['n_accepted']) == (len(rst.trace), n_acc), (st, len(rst.trace), n_acc)
                assert np.abs(got.cpu().numpy() - ref).max() <= band * scale
        # float32: the reference's scheme (state advanced with the PREDICTOR, g in float32) amplifies rounding - numpy's float32 mean
        # and the kernels' float64 accumulation of the error ratio part ways after a dozen steps, by more than the tolerance (the
        # float64 oracle itself is 2e-2 off the true solution at rtol 1e-4 on this problem).  The product's two engines agree.
        loop = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method='adams', options={'force_plane_kernels': True}, **tol)
        sl = dict(odeint.last_stats)
        assert sl.get('engine') == 'plane kernels'
        if problem == 'spiral':
            assert abs(sl['n_attempts'] - st['n_attempts']) <= max(2, st['n_attempts'] // 20)
            assert float((got - loop).abs().max()) <= 1e-4 * scale
        else:
            assert (sl['n_attempts'], sl['n_accepted']) == (st['n_attempts'], st['n_accepted']), (sl, st)
            assert float((got - loop).abs().max()) <= band * scale


def test_variable_order_adams_output_grids():
    """One requested time (solution = [y0], the solver still runs before_integrate) and 200 of them (every output is a state that
    landed exactly on its time: the steps are clipped, adams.py:136-137) - fused kernel against the per-step loop."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(4)
    y0 = torch.tensor(1.0 + 0.5 * rng.uniform(size=(100, 2)), device=dev())
    one = odeint(rhs.LotkaVolterra(), y0, torch.tensor([0.5]), method='adams')
    assert one.shape == (1, 100, 2) and torch.equal(one[0], y0) and dict(odeint.last_stats).get('engine', '').startswith('fused variable-order')
    tm = torch.tensor(np.linspace(0., 1., 200))
    a = odeint(rhs.LotkaVolterra(), y0, tm, method='adams', rtol=1e-6, atol=1e-8)
    sa = dict(odeint.last_stats)
    b = odeint(rhs.LotkaVolterra(), y0, tm, method='adams', rtol=1e-6, atol=1e-8, options={'force_plane_kernels': True})
    sb = dict(odeint.last_stats)
    assert sa['n_launches'] == 1 and sa['n_attempts'] == sb['n_attempts'] >= 199 and sb.get('engine') == 'plane kernels'
    assert float((a - b).abs().max()) <= 1e-12


@pytest.mark.parametrize('max_order', [2, 4, 7])
def test_variable_order_adams_max_order_option(max_order):
    """options={'max_order': n} (adams.py:87-90) caps the order selection of the one-launch kernel exactly as it caps the per-step loop's."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(2)
    y0 = torch.tensor(1.0 + 0.5 * rng.uniform(size=(500, 2)), device=dev())
    t = torch.tensor(np.linspace(0., 2., 7))
    a = odeint(rhs.LotkaVolterra(), y0, t, method='adams', rtol=1e-6, atol=1e-8, options={'max_order': max_order})
    sa = dict(odeint.last_stats)
    b = odeint(rhs.LotkaVolterra(), y0, t, method='adams', rtol=1e-6, atol=1e-8, options={'max_order': max_order, 'force_plane_kernels': True})
    sb = dict(odeint.last_stats)
    assert sa.get('engine', '').startswith('fused variable-order') and sb.get('engine') == 'plane kernels'
    assert (sa['n_attempts'], sa['n_accepted']) == (sb['n_attempts'], sb['n_accepted']), (sa, sb)
    assert float((a - b).abs().max()) <= 1e-9


# ---------------------------------------------------------------------------------------------
# round 4: the one-launch 'adams' kernel DIRECTLY on the reference's own fixtures, and float32 against the
# float32 oracle over the horizon before rounding forks the two
# ---------------------------------------------------------------------------------------------
def _fixture_rhs(meta):
    """The fixture's right-hand side as device code for the row-local kernels (tests/problems.py:13-34 for the scalar problems:
    written the way the reference's callable evaluates them)."""
    from tfdiffeq_amd import rhs
    if meta['rhs'] == 'cubic_linear':
        return rhs.CubicLinear(torch.tensor(meta['rhs_params']['W'], dtype=torch.float64)), None
    if meta['rhs'] == 'constant':
        a, b = meta['rhs_params'].get('a', 0.2), meta['rhs_params'].get('b', 3.0)
        return rhs.CustomRowLocal(1, 'k[0] = p[0] + pow(y[0] - (p[0] * t + p[1]), (T)5);', params=[a, b]), (1, 1)
    if meta['rhs'] == 'sine':
        return rhs.CustomRowLocal(1, 'k[0] = 2 * y[0] / t + pow(t, (T)4) * sin(2 * t) - pow(t, (T)2) + 4 * pow(t, (T)3);'), (1, 1)
    raise KeyError(meta['rhs'])


@pytest.mark.parametrize('name', ['run_constant_adams', 'run_sine_adams', 'run_spiral_b64_adams'])
def test_reference_adams_fixtures_on_the_one_launch_kernel(name):
    """adams.py:130-211 as captured from the reference (trace columns prev_t, next_t, order, accepted), on k_adams_vc_rowlocal.
    constant: the attempt and accept counts of the reference exactly, values to 1e-9.  The method rounds its g vector to float32
    every step (adams.py:34) and advances with the predictor, so a last-bit difference anywhere becomes another step sequence:
    spiral (209 attempts of 64 identical rows; one error ratio summed in another order) and sine (pow / sin of the device library
    against numpy's: 145 / 136 attempts / accepted against the reference's 143 / 135) keep the +-5 % allowance the plane-engine
    test gives this method - with the values inside the band the reference's own test uses for it (1e-4, odeint_tests.py:86-92)."""
    from tfdiffeq_amd import odeint
    d, meta = load(name)
    f, shape = _fixture_rhs(meta)
    kw = {k: meta[k] for k in ('rtol', 'atol') if meta.get(k) is not None}
    y0 = torch.tensor(d['y0'], device=dev())
    if shape is not None:
        y0 = y0.reshape(shape)
    sol = odeint(f, y0, torch.as_tensor(d['t']), method='adams', **kw)
    st = dict(odeint.last_stats)
    assert st.get('engine', '').startswith('fused variable-order Adams') and st['n_launches'] == 1 and st['status'] == 0, st
    got = sol.cpu().numpy().reshape(d['y'].shape)
    ref_att, ref_acc = len(d['trace']), int(d['trace'][:, 3].sum())
    scale = max(1.0, np.abs(d['y']).max())
    if name != 'run_constant_adams':
        assert abs(st['n_attempts'] - ref_att) <= max(2, ref_att // 20) and abs(st['n_accepted'] - ref_acc) <= max(2, ref_acc // 20), (st, ref_att, ref_acc)
        assert np.abs(got - d['y']).max() <= (1e-5 if name == 'run_spiral_b64_adams' else 1e-4) * scale
    else:
        assert (st['n_attempts'], st['n_accepted']) == (ref_att, ref_acc), (st, ref_att, ref_acc)
        assert np.abs(got - d['y']).max() <= 1e-9 * scale, np.abs(got - d['y']).max() / scale


@pytest.mark.parametrize('problem', ['lv', 'lorenz'])
def test_float32_adams_against_the_float32_oracle_before_the_fork(problem):
    """float32 `adams`: numpy's float32 mean and the kernel's float64 accumulation of the error ratio part ways after about a dozen
    attempts (the state advances with the PREDICTOR and g is rounded to float32, adams.py:34,210 - rounding differences are
    amplified).  Up to there the two must agree: the longest horizon the float32 oracle covers in <= 12 attempts, counts exact,
    values to a few float32 ulps."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(12)
    if problem == 'lv':
        f, fn, y0 = rhs.LotkaVolterra(1.5, 1.0, 3.0, 1.0), _lv_np, (1.0 + 0.5 * rng.uniform(size=(300, 2))).astype(np.float32)
        horizons = (0.4, 0.2, 0.1, 0.05, 0.02, 0.01)
    else:
        f, fn, y0 = rhs.Lorenz(), _lorenz_np, (np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((300, 3))).astype(np.float32)
        horizons = (0.1, 0.05, 0.02, 0.01, 0.005, 0.002)
    tol = dict(rtol=1e-4, atol=1e-6)
    chosen = None
    for h in horizons:
        tt = np.array([0., h])
        ref, rst = OA.odeint(fn, y0, tt, method='adams', return_stats=True, **tol)
        if len(rst.trace) <= 12:
            chosen = (tt, ref, rst)
            break
    assert chosen is not None, 'no horizon with <= 12 oracle attempts'
    tt, ref, rst = chosen
    got = odeint(f, torch.tensor(y0, device=dev()), torch.tensor(tt), method='adams', **tol)
    st = dict(odeint.last_stats)
    assert st.get('engine', '').startswith('fused variable-order Adams'), st
    n_acc = int(sum(1 for r in rst.trace if r[3] > 0))
    assert (st['n_attempts'], st['n_accepted']) == (len(rst.trace), n_acc), (st, len(rst.trace), n_acc)
    assert np.abs(got.cpu().numpy() - np.asarray(ref)).max() <= 2e-6 * max(1.0, np.abs(np.asarray(ref)).max())
