"""Pin the oracle (oracle/ode_numpy.py) against the golden fixtures captured from the
reference's own solver files (tests/golden/make_golden.py).  CPU only.

Bar: fp64 values to <= 1e-13 relative, and the
accept/reject sequence reproduced exactly.
"""
import numpy as np
import pytest

from oracle import ode_numpy as O
from oracle.rhs_numpy import make_rhs
from tests.golden_util import load, run_cases, mlp_weights


def _close(a, b, rel, what=''):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    err = float(np.max(np.abs(a - b))) / scale if b.size else 0.0
    assert err <= rel, '%s: max scaled err %.3e > %.1e' % (what, err, rel)


def test_tableau_constants_match_reference_modules():
    d, _ = load('fn_tableaus')
    for name, tb in (('dopri5', O.DOPRI5), ('tsit5', O.TSIT5_REF), ('bosh3', O.BOSH3)):
        alpha, beta, c_sol, c_err = O.tableau_arrays(tb)
        assert np.array_equal(alpha, d[name + '_alpha'])
        assert np.array_equal(beta, d[name + '_beta'])
        assert np.array_equal(c_sol, d[name + '_c_sol'])
        assert np.array_equal(c_err, d[name + '_c_error'])
    assert np.array_equal(np.asarray(O.DOPRI5_C_MID), d['dopri5_c_mid'])
    assert np.array_equal(np.asarray(O.BOSH3_C_MID), d['bosh3_c_mid'])
    # the quirks themselves
    assert O.BOSH3.alpha[0] == 5.0 and O.BOSH3.beta[1][1] == 7.5
    assert abs(sum(O.TSIT5_REF.c_error) - 0.9697) < 1e-3
    assert abs(sum(O.TSIT5_FIXED.c_error)) < 1e-14


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_function_vectors(dtype):
    d, meta = load('fn_rkstep_' + dtype)
    npdt = np.dtype(dtype)
    tol = 1e-13 if dtype == 'float64' else 2e-6
    f_ = make_rhs('tdep', dtype=npdt)
    f = lambda t, ys: (f_(t, ys[0]),)  # noqa: E731
    y0, t0, dt = d['y0'], float(d['t0']), float(d['dt'])
    f0 = f_(npdt.type(t0), y0)
    _close(f0, d['f0'], tol, 'f0')
    for name, tb in (('dopri5', O.DOPRI5), ('tsit5', O.TSIT5_REF), ('bosh3', O.BOSH3)):
        y1, f1, err, k = O.runge_kutta_step(f, (y0,), (f0,), t0, dt, tb)
        assert y1[0].dtype == npdt
        _close(y1[0], d[name + '_y1'], tol, name + ' y1')
        _close(f1[0], d[name + '_f1'], tol, name + ' f1')
        _close(err[0], d[name + '_err'], tol, name + ' err')
        _close(np.stack(k[0]), d[name + '_k'], tol, name + ' k')
        ratio = O.compute_error_ratio(err, [meta['ratio_rtol']], [meta['ratio_atol']], (y0,), y1)
        _close(ratio[0], d[name + '_ratio'], 1e-12 if dtype == 'float64' else 1e-5, name + ' ratio')
    y1, f1, err, k = O.runge_kutta_step(f, (y0,), (f0,), t0, dt, O.DOPRI5)
    coeff = O.interp_fit_mid((y0,), y1, k, dt, O.DOPRI5_C_MID)
    _close(np.stack([c[0] for c in coeff]), d['dopri5_interp_coeff'], tol * 10, 'dopri5 interp coeff')
    for j, te in enumerate(d['interp_eval_times']):
        _close(O.interp_evaluate(coeff, t0, t0 + dt, te)[0], d['dopri5_interp_eval%d' % j], tol * 10, 'interp eval')
    y1, f1, err, k = O.runge_kutta_step(f, (y0,), (f0,), t0, dt, O.BOSH3)
    coeff = O.interp_fit_mid((y0,), y1, k, dt, O.BOSH3_C_MID)
    _close(np.stack([c[0] for c in coeff]), d['bosh3_interp_coeff'], tol * 100, 'bosh3 interp coeff')
    _close(O.interp_evaluate(coeff, t0, t0 + dt, t0 + 0.3 * dt)[0], d['bosh3_interp_eval1'], tol * 100, 'bosh3 eval')
    if dtype == 'float64':
        y1, f1, err, k = O.runge_kutta_step(f, (y0,), (f0,), t0, dt, O.TSIT5_REF)
        _close(O.interp_eval_tsit5(np.float64(t0), np.float64(t0 + dt), k, np.float64(t0 + 0.3 * dt))[0],
               d['tsit5_interp_eval1'], tol, 'tsit5 dense output (reference, from f0)')
    dy = O.rk4_alt_step(f, npdt.type(t0), npdt.type(dt), (y0,))
    _close(dy[0], d['rk4_dy'], tol, 'rk4 dy')
    for order in (4, 2):
        h, _ = O.select_initial_step(f, t0, (y0,), order, meta['init_rtol'], meta['init_atol'], f0=(f0,))
        _close(h, d['init_step_order%d' % order], 1e-12 if dtype == 'float64' else 1e-5, 'init step')
    z = lambda t, ys: (ys[0] * 0.0,)  # noqa: E731
    h, _ = O.select_initial_step(z, 0.0, (y0,), 4, meta['init_rtol'], meta['init_atol'])
    _close(h, d['init_step_zero_f'], 1e-12 if dtype == 'float64' else 1e-5, 'init step zero f')
    h, _ = O.select_initial_step(f, t0, (y0 * 0,), 4, meta['init_rtol'], meta['init_atol'])
    _close(h, d['init_step_zero_y'], 1e-12 if dtype == 'float64' else 1e-5, 'init step zero y')


def test_step_controllers():
    d, meta = load('fn_step_controller')
    for order in (5, 3):
        for dt_name in ('float64', 'float32'):
            got = [O.optimal_step_size(np.float64(meta['last_step']), (np.dtype(dt_name).type(r),), meta['safety'],
                                       meta['ifactor'], meta['dfactor'], order) for r in d['ratios']]
            np.testing.assert_allclose(got, d['misc_order%d_%s' % (order, dt_name)], rtol=1e-14, atol=0)
    got = [O.optimal_step_size_tsit5(np.float64(meta['last_step']), np.float64(r), meta['safety'], meta['ifactor'],
                                     meta['dfactor'], 5) for r in d['ratios']]
    np.testing.assert_allclose(got, d['tsit5_order5_float64'], rtol=1e-14, atol=0)
    # F4: the exponent is float64(float32(1/order))
    assert float(np.float64(np.float32(1. / 5))) == 0.20000000298023224


def _run_oracle(name):
    d, meta = load(name)
    dtype = d['y0_0'].dtype if meta['tuple_state'] else d['y0'].dtype
    weights = mlp_weights() if meta['rhs'] == 'mlp_tanh' else None
    f = make_rhs(meta['rhs'], meta['rhs_params'], dtype=dtype, weights=weights)
    kw = {}
    if meta['rtol'] is not None:
        kw['rtol'] = meta['rtol']
    if meta['atol'] is not None:
        kw['atol'] = meta['atol']
    if meta['options'] is not None:
        kw['options'] = meta['options']
    if meta['tuple_state']:
        y0 = (d['y0_0'], d['y0_1'])
        func = lambda t, ys: tuple(f(t, y_) for y_ in ys)  # noqa: E731
    else:
        y0, func = d['y0'], f
    if meta['method'] in ('adams', 'fixed_adams', 'explicit_adams'):
        from oracle import adams_numpy as AD
        sol, stats = AD.odeint(func, y0, d['t'], method=meta['method'], return_stats=True, **kw)
    else:
        sol, stats = O.odeint(func, y0, d['t'], method=meta['method'], return_stats=True,
                              max_attempts=meta['max_attempts'], **kw)
    return d, meta, sol, stats


@pytest.mark.parametrize('name', run_cases())
def test_whole_runs_match_reference(name):
    d, meta, sol, stats = _run_oracle(name)
    f32 = (d['y0_0'] if meta['tuple_state'] else d['y0']).dtype == np.float32
    vtol = 1e-5 if f32 else 1e-12
    if meta['rtol'] is not None and meta['rtol'] <= 1e-9:
        vtol = 1e-9                       # step sizes differ in the roundoff regime (below), so do the last digits
    if 'trace' in d.files and meta['method'] == 'adams':
        tr = np.asarray(stats.trace, dtype=np.float64).reshape(-1, 4)
        ref = d['trace']                                  # (prev_t, next_t, order, accepted)
        assert tr.shape == ref.shape, 'attempt count %d vs reference %d' % (len(tr), len(ref))
        assert np.array_equal(tr[:, 2:], ref[:, 2:]), 'order / accept sequence differs'
        np.testing.assert_allclose(tr[:, :2], ref[:, :2], rtol=1e-9, atol=0)
    elif 'trace' in d.files:
        tr = np.asarray(stats.trace, dtype=np.float64).reshape(-1, 4)
        ref = d['trace']
        assert tr.shape == ref.shape, 'attempt count %d vs reference %d' % (len(tr), len(ref))
        assert np.array_equal(tr[:, 2], ref[:, 2]), 'accept/reject sequence differs'
        # at tolerances near roundoff (dopri8 runs: rtol 1e-12 / 1e-9) the error estimate itself is rounding noise,
        # so dt only agrees loosely (libm-vs-SVML pow, 0-d array vs scalar paths); the accept sequence is still exact
        noise = meta['rtol'] is not None and meta['rtol'] <= 1e-9
        np.testing.assert_allclose(tr[:, [0, 1, 3]], ref[:, [0, 1, 3]], rtol=1e-2 if noise else (1e-4 if f32 else 1e-9), atol=0)
    # the generator counts RHS calls per tuple component; the oracle counts calls of the tuple func
    assert stats.nfe * (2 if meta['tuple_state'] else 1) == int(d['nfe'])
    if meta['max_attempts'] is not None:
        _close(stats.state_y[0], d['y_after_attempts'], vtol, 'state after K attempts')
        _close(stats.state_t, d['t_after_attempts'], 1e-13, 't after K attempts')
    elif meta['tuple_state']:
        _close(sol[0], d['y_0'], vtol, 'y_0')
        _close(sol[1], d['y_1'], vtol, 'y_1')
    else:
        assert sol.shape == d['y'].shape and sol.dtype == d['y'].dtype
        _close(sol, d['y'], vtol, 'solution')


def test_anchor_values_from_survey():
    """SURVEY.md 8(c) anchors: LV rk4 1000 steps and the NFE formula of the spiral."""
    d, meta, sol, stats = _run_oracle('run_lv_rk4_1000')
    assert stats.nfe == 4000
    np.testing.assert_allclose(sol[-1], [1.0263447325842516, 0.9096910992758575], rtol=1e-13)
    d, meta, sol, stats = _run_oracle('run_sine_dopri5')
    assert stats.nfe == 2 + 6 * stats.n_attempts == 266 and stats.n_accepted == 40


def test_reference_unit_test_bars_hold_for_the_oracle():
    """tests/odeint_tests.py: rel_error(sol, y) < 1e-4 against the closed forms (tests/problems.py)."""
    t = np.linspace(1., 8., 10).astype(np.float32)
    t64 = t.astype(np.float64)
    exact_c = 0.2 * t64 + 3.0
    exact_s = (-0.5 * t64 ** 4 * np.cos(2 * t64) + 0.5 * t64 ** 3 * np.sin(2 * t64) + 0.25 * t64 ** 2 * np.cos(2 * t64)
               - t64 ** 3 + 2 * t64 ** 4 + (np.pi - 0.25) * t64 ** 2)
    for m in ('euler', 'rk4', 'dopri5', 'bosh3'):
        y = O.odeint(make_rhs('constant'), np.float64(exact_c[0]), t, method=m)
        assert np.max(np.abs((exact_c - y) / exact_c)) < 1e-4
    y = O.odeint(make_rhs('sine'), np.float64(exact_s[0]), t, method='dopri5')
    assert np.max(np.abs((exact_s - y) / exact_s)) < 1e-4
    # backwards in time
    y = O.odeint(make_rhs('constant'), np.float64(exact_c[-1]), t[::-1].copy(), method='dopri5')
    assert np.max(np.abs((exact_c[::-1] - y) / exact_c[::-1])) < 1e-4


def test_oracle_error_behaviour():
    f = make_rhs('constant')
    with pytest.raises(ValueError):
        O.odeint(f, np.float64(3.2), np.array([1., 2.]), options={'first_step': 0.1})
    with pytest.raises(KeyError):
        O.odeint(f, np.float64(3.2), np.array([1., 2.]), method='nope')
    with pytest.raises(AssertionError):
        O.odeint(f, np.float64(3.2), np.array([1., 3., 2.]), method='dopri5')
    with pytest.raises(AssertionError):
        O.odeint(lambda t, y: y * y, np.float64(1.0), np.array([0., 2.0]), method='dopri5',
                 options={'max_num_steps': 5})


def test_fixed_tsit5_extension_is_accurate():
    """The oracle's corrected-Tsit5 extension (no reference counterpart) against scipy DOP853."""
    from scipy.integrate import solve_ivp
    A = np.array([[-0.1, 2.0], [-2.0, -0.1]])
    f = lambda t, y: (y ** 3) @ A  # noqa: E731
    y0 = np.array([[2., 0.]])
    t = np.linspace(0., 5., 6)
    y, stats = O.odeint(f, y0, t, rtol=1e-8, atol=1e-10, method='tsit5', options={'tsit5_fixed': True},
                        return_stats=True)
    ref = solve_ivp(lambda t_, y_: f(t_, y_.reshape(1, 2)).ravel(), (0, 5), y0.ravel(), method='DOP853',
                    t_eval=t, rtol=1e-13, atol=1e-13).y.T
    assert np.max(np.abs(y[:, 0, :] - ref)) < 1e-6
    assert stats.n_attempts < 1000      # (the reference-faithful tableau needs ~1e8 attempts here, F6a)


def test_next_solver_function_vectors():
    """dopri8 / adaptive_heun through the same runge_kutta_step restatement."""
    d, _ = load('fn_rkstep_next_float64')
    tn, _ = load('fn_tableaus_next')
    tb8, cm8 = O.load_dopri8()
    a, b, cs, ce = O.tableau_arrays(tb8)
    assert np.array_equal(a, tn['dopri8_alpha']) and np.array_equal(b, tn['dopri8_beta'])
    assert np.array_equal(cs, tn['dopri8_c_sol']) and np.array_equal(ce, tn['dopri8_c_error'])
    a, b, cs, ce = O.tableau_arrays(O.ADAPTIVE_HEUN)
    assert np.array_equal(a, tn['adaptive_heun_alpha']) and np.array_equal(cs, tn['adaptive_heun_c_sol'])
    assert np.array_equal(ce, tn['adaptive_heun_c_error']) and not O.is_fsal_shaped(O.ADAPTIVE_HEUN)
    f_ = make_rhs('tdep')
    f = lambda t, ys: (f_(t, ys[0]),)  # noqa: E731
    y0, t0, dt = d['y0'], float(d['t0']), float(d['dt'])
    f0 = f_(np.float64(t0), y0)
    for name, tb, cm in (('dopri8', tb8, cm8), ('adaptive_heun', O.ADAPTIVE_HEUN, O.ADAPTIVE_HEUN_C_MID)):
        y1, f1, err, k = O.runge_kutta_step(f, (y0,), (f0,), t0, dt, tb)
        _close(y1[0], d[name + '_y1'], 1e-13, name + ' y1')
        _close(f1[0], d[name + '_f1'], 1e-13, name + ' f1')
        _close(err[0], d[name + '_err'], 1e-13, name + ' err')
        _close(np.stack(k[0]), d[name + '_k'], 1e-13, name + ' k')
        coeff = O.interp_fit_mid((y0,), y1, k, dt, cm)
        _close(O.interp_evaluate(coeff, t0, t0 + dt, t0 + 0.3 * dt)[0], d[name + '_interp_eval'], 1e-12, name + ' dense')


# ---------------------------------------------------------------------------------------------
# the torch-CPU eager restatement that bench.py times as `cpu_baseline` (oracle/ode_torch_cpu.py)
# ---------------------------------------------------------------------------------------------
def _torch_rhs(meta):
    import torch
    p = meta['rhs_params']
    if meta['rhs'] == 'linear':
        W = torch.tensor(p['W'], dtype=torch.float64)
        return lambda t, y: y @ W
    if meta['rhs'] == 'cubic_linear':
        W = torch.tensor(p['W'], dtype=torch.float64)
        return lambda t, y: (y ** 3) @ W
    if meta['rhs'] == 'lorenz':
        s, be, r = p['sigma'], p['beta'], p['rho']
        return lambda t, y: torch.stack([s * (y[..., 1] - y[..., 0]), y[..., 0] * (r - y[..., 2]) - y[..., 1],
                                         y[..., 0] * y[..., 1] - be * y[..., 2]], -1)
    a, b, c, d = p['a'], p['b'], p['c'], p['d']
    return lambda t, y: torch.stack([a * y[..., 0] - b * y[..., 0] * y[..., 1], -c * y[..., 1] + d * y[..., 0] * y[..., 1]], -1)


@pytest.mark.parametrize('name', ['run_linear_b48_d16_dopri5', 'run_linear_b48_d16_dopri5_T5', 'run_spiral_b64_dopri5',
                                  'run_lorenz_b64_dopri5', 'run_lv_b32_dopri5'])
def test_torch_cpu_restatement_reproduces_reference_runs(name):
    """Same solution (fp64 <= 1e-12) and the IDENTICAL accept / reject sequence as the reference's own run."""
    import torch
    from oracle import ode_torch_cpu as TC
    d, meta = load(name)
    kw = {}
    if meta['rtol'] is not None:
        kw['rtol'] = meta['rtol']
    if meta['atol'] is not None:
        kw['atol'] = meta['atol']
    sol, st = TC.odeint_dopri5(_torch_rhs(meta), torch.tensor(d['y0']), d['t'], **kw)
    assert np.abs(sol.numpy() - d['y']).max() < 1e-12
    assert st.n_attempts == len(d['trace']) and st.n_accepted == int(d['trace'][:, 2].sum())
    assert st.nfe == 2 + 6 * st.n_attempts


# ---------------------------------------------------------------------------------------------
# DETEST (tests/DETEST/detest.py:9-351): the restated problem set + the oracle against the reference's own results
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', [c + i for c in 'ABCDE' for i in '12345'])
def test_detest_problem_matches_the_reference_run(name):
    """Every problem from t = 0 to 20 with dopri5 at tol 1e-3 and 1e-6 (run.py:25-60): the oracle on the restated problem
    takes the IDENTICAL accept / reject sequence and NFE as the reference did and lands on the same y(20)."""
    from oracle import detest_problems as DP
    d, meta = load('fn_detest')
    f, y0 = DP.problem(name, np)
    np.testing.assert_array_equal(np.asarray(y0), d[name + '_y0'])
    for tol, nfe, att, acc in d[name + '_runs']:
        with np.errstate(all='ignore'):
            sol, st = O.odeint(f, np.asarray(y0), np.array([0., DP.T_END]), rtol=tol, atol=tol, method='dopri5', return_stats=True)
        assert (st.nfe, st.n_attempts, st.n_accepted) == (int(nfe), int(att), int(acc)), (name, tol)
        ref = d['%s_y20_tol%g' % (name, tol)]
        assert np.abs(sol[1] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), (name, tol)
        if name in DP.EXACT and tol == 1e-6:
            assert abs(float(sol[1]) - DP.EXACT[name](DP.T_END)) < 2e-5 * max(1.0, abs(DP.EXACT[name](DP.T_END)))
