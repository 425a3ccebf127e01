"""GPU parity tests added in round 2 (-m gpu), all through the C ABI:
  * the DEVICE step controller driven with the reference's own vectors (mi_ode_controller_update, fn_step_controller.npz);
  * step-sequence checks made exact wherever the reference trace does not sit on the accept threshold;
  * the oracle run at BASELINE.json's FULL sizes and compared inside the north star's band (rtol 1e-5 / atol 1e-6),
    with an independent anchor (scipy DOP853) for the configuration the reference cannot pin (tsit5).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ode_numpy as O
from oracle.rhs_numpy import make_rhs
from tests.bands import assert_f32
from tests.golden_util import RK_ORDER, load, mlp_weights, run_cases, trace_is_decided
from tests.rhs_util import device_rhs

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6          # north-star parity band


def dev():
    return torch.device('cuda:0')


def to_dev(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def assert_band(got, ref, rtol=RTOL, atol=ATOL, what=''):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, '%s shape %s vs %s' % (what, got.shape, ref.shape)
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert not bad.any(), '%s: %d/%d outside band, max abs diff %.3e' % (what, bad.sum(), bad.size, np.abs(got - ref).max())


# ---------------------------------------------------------------------------------------------
# device controller, vector level (misc.py:256-287, tsit5.py:53-62)
# ---------------------------------------------------------------------------------------------
def _controller(params, phase, recs, states):
    from tfdiffeq_amd import _native as N
    lib = N.load()
    n = len(recs)
    a_in = np.ascontiguousarray(np.asarray(recs, dtype=np.float64).reshape(n, 8))
    a_st = np.ascontiguousarray(np.asarray(states, dtype=np.float64).reshape(n, 4))
    out = np.zeros((n, 8))
    p = N.CtrlParams(**params)
    dp = C.POINTER(C.c_double)
    with torch.cuda.device(dev()):
        N.check(lib.mi_ode_controller_update(C.byref(p), phase, n, a_in.ctypes.data_as(dp), a_st.ctypes.data_as(dp),
                                             out.ctypes.data_as(dp), N.stream_ptr(dev())), 'mi_ode_controller_update')
    return out


@pytest.mark.parametrize('variant', ['misc_order5_float64', 'misc_order5_float32', 'misc_order3_float64', 'misc_order3_float32',
                                     'tsit5_order5_float64'])
def test_device_controller_matches_reference_vectors(variant):
    """controller_apply() - the function every kernel and the separate controller launch run - on the ratio grid captured
    from the reference: r == 0, r < 1 (dfactor forced to 1), the ifactor / dfactor clamps, both controllers, both dtypes.
    atol = 1, rtol = 0, N = 1 make the error ratio equal the record's sum, so the device sees exactly the fixture's ratio."""
    from tfdiffeq_amd import _native as N
    d, meta = load('fn_step_controller')
    kind, order, dt_name = variant.split('_')
    params = dict(rtol=0.0, atol=1.0, safety=meta['safety'], ifactor=meta['ifactor'], dfactor=meta['dfactor'],
                  order=int(order[-1]), init_order=int(order[-1]) - 1, controller=N.CTRL_TSIT5 if kind == 'tsit5' else N.CTRL_MISC,
                  dtype=N.F32 if dt_name == 'float32' else N.F64)
    ratios = d['ratios']
    recs = [[0.0, 0.0, r, 0.0, 0.0, 1.0, 0.0, 0.0] for r in ratios]
    states = [[0.5, meta['last_step'], 0.0, 0.0] for _ in ratios]
    out = _controller(params, 2, recs, states)
    seen = ratios.astype(np.float32).astype(np.float64) if dt_name == 'float32' else ratios
    np.testing.assert_array_equal(out[:, 0], seen)                                    # the ratio the controller formed
    np.testing.assert_array_equal(out[:, 1], (seen <= 1.0).astype(np.float64))        # accept test (dopri5.py:108)
    # next step size: everything but pow() is exactly rounded on both sides; the device's pow (ocml) and numpy's (libm)
    # differ by up to two units in the last place on this grid (measured), so that is the bar - not 1e-5, not 1e-12
    ulps = np.abs(out[:, 2] - d[variant]) / np.spacing(np.abs(d[variant]))
    assert ulps.max() <= 2.0, ulps.max()
    exact = (seen == 0) | (out[:, 2] == meta['last_step'] * meta['ifactor']) | (out[:, 2] == meta['last_step'] * meta['dfactor'])
    np.testing.assert_array_equal(out[exact, 2], d[variant][exact])                  # r == 0 and the clamps involve no pow: exact
    # accepted steps advance t1 by exactly dt, rejected ones keep it (dopri5.py:110-116)
    np.testing.assert_array_equal(out[:, 3], np.where(seen <= 1.0, 0.5 + meta['last_step'], 0.5))
    assert (out[:, 5] == 0).all()


def test_device_controller_initial_step_phases_match_the_oracle():
    """The two halves of misc._select_initial_step (misc.py:227-245) as the kernels' controller computes them, against the
    oracle's restatement, including the degenerate branches (d0 or d1 < 1e-5; both <= 1e-15)."""
    from tfdiffeq_amd import _native as N
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(32):
        n = int(rng.integers(1, 5000))
        cases.append((n, rng.uniform(1e-3, 1e3) * n, rng.uniform(1e-3, 1e3) * n, rng.uniform(1e-6, 1e2) * n))
    cases += [(10, 1e-12, 5.0, 1.0), (10, 5.0, 1e-13, 1.0), (7, 3.0, 1e-40, 1e-45), (3, 0.0, 0.0, 0.0)]
    for order in (4, 2):
        params = dict(rtol=1e-6, atol=1e-9, safety=0.9, ifactor=10.0, dfactor=0.2, order=order + 1, init_order=order,
                      controller=N.CTRL_MISC, dtype=N.F64)
        recs0 = [[0, 0, s0, s1, 0, n, 0, 0] for n, s0, s1, _ in cases]
        o0 = _controller(params, 0, recs0, [[0.0, 0.0, 0.0, 0.0]] * len(cases))
        for (n, s0, s1, s2), row in zip(cases, o0):
            d0, d1 = np.sqrt(s0) / n ** 0.5, np.sqrt(s1) / n ** 0.5               # misc.py:170-175, 227-228
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * (d0 / d1)           # :230-233
            # sqrt and the divisions are exactly rounded; N ** 0.5 goes through pow() (misc.py:175), one ulp between libraries
            np.testing.assert_allclose([row[7], row[2], row[6]], [d0, d1, h0], rtol=5e-16, atol=0, err_msg=str((n, s0, s1)))
        recs1 = [[0, 0, s2, 0, 0, n, 0, 0] for n, _, _, s2 in cases]
        o1 = _controller(params, 1, recs1, [[0.0, 0.0, r[6], r[2]] for r in o0])
        for (n, s0, s1, s2), r0, row in zip(cases, o0, o1):
            h0, d1 = r0[6], r0[2]
            d2 = (np.sqrt(s2) / n ** 0.5) / h0                                      # :237
            if d1 <= 1e-15 and d2 <= 1e-15:
                h1 = max(1e-6, h0 * 1e-3)                                           # :239-240
            else:
                h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))               # :242-245
            np.testing.assert_allclose(row[2], min(100 * h0, h1), rtol=2e-15, atol=0)


# ---------------------------------------------------------------------------------------------
# step sequences: exact wherever the reference trace stays clear of the accept threshold
# ---------------------------------------------------------------------------------------------
def _fused_cases():
    out = []
    for n in run_cases():
        d, meta = load(n)
        if meta['max_attempts'] is not None or 'trace' not in d.files or meta['tuple_state'] or d['y0'].dtype != np.float64:
            continue
        if meta['rhs'] not in ('cubic_linear', 'linear', 'lotka_volterra', 'lorenz'):
            continue
        if meta['method'] not in ('dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun'):
            continue
        out.append(n)
    return out


@pytest.mark.parametrize('name', _fused_cases())
def test_fused_engine_step_sequence_is_exactly_the_references(name):
    """Attempt and accept counts and the final step size of the fused engine against the reference trace: equal, not
    'within 5 %', whenever the trace does not pass through ratio == 1 +- 1e-9 (where a last-bit difference of the
    reduction order may legitimately flip one decision)."""
    from tfdiffeq_amd import odeint
    d, meta = load(name)
    tr = d['trace']
    if not trace_is_decided(tr, RK_ORDER[meta['method']], meta['method'] == 'tsit5'):
        pytest.skip('the reference trace touches the accept threshold')
    f = device_rhs(meta['rhs'], meta['rhs_params'])
    kw = {}
    if meta.get('rtol') is not None:
        kw['rtol'] = meta['rtol']
    if meta.get('atol') is not None:
        kw['atol'] = meta['atol']
    opts = dict(meta.get('options') or {})
    if meta['method'] == 'tsit5':
        opts['refcompat'] = True
    for fusion in ('step', 'auto'):
        o = dict(opts, fusion=fusion)
        try:
            sol = odeint(f, to_dev(d['y0']), torch.as_tensor(d['t']), method=meta['method'], options=o, **kw)
        except Exception as e:                                       # a schedule this problem has no kernel for
            if 'no whole-attempt kernel' in str(e) or 'fusion' in str(e):
                continue
            raise
        st = dict(odeint.last_stats)
        assert st['n_attempts'] == len(tr) and st['n_accepted'] == int(tr[:, 2].sum()), (name, fusion, st, len(tr))
        if 'dt' in st:                                               # (the plane-kernel engine keeps dt on the host)
            np.testing.assert_allclose(st['dt'], tr[-1, 3], rtol=1e-7)   # the step size the next attempt would take
        assert_band(sol.cpu(), d['y'], RTOL, ATOL, name)


# ---------------------------------------------------------------------------------------------
# the oracle at BASELINE.json's full sizes
# ---------------------------------------------------------------------------------------------
def _config4(batch=65536, D=128):
    g2 = torch.Generator().manual_seed(2)
    S = torch.randn(D, D, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(D, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(D)
    g3 = torch.Generator().manual_seed(3)
    y0 = torch.randn(batch, D, generator=g3, dtype=torch.float64)
    return A, y0


def test_config4_full_size_against_the_oracle():
    """BASELINE config 4 exactly as bench.py runs it (65536 x 128, Dopri5 fp64, rtol 1e-6, atol 1e-9, t = [0, 1]) against the
    oracle on the SAME full-size input: identical step sequence, solution inside the north star's band (measured: ~1e-13,
    the matmul accumulation order is the only difference)."""
    from tfdiffeq_amd import odeint, rhs
    A, y0 = _config4()
    t = np.array([0., 1.])
    W = A.t().contiguous().numpy()
    ref, st_ref = O.odeint(lambda t_, y: y @ W, y0.numpy(), t, rtol=1e-6, atol=1e-9, method='dopri5', return_stats=True)
    for fusion in ('auto', 'step', 'stage'):
        sol = odeint(rhs.Linear.from_matrix(A), y0.to(dev()), torch.tensor(t), rtol=1e-6, atol=1e-9, method='dopri5',
                     options={'fusion': fusion})
        st = dict(odeint.last_stats)
        assert st['n_attempts'] == st_ref.n_attempts and st['n_accepted'] == st_ref.n_accepted, (fusion, st, vars(st_ref))
        assert_band(sol.cpu(), ref, RTOL, ATOL, 'config 4 ' + fusion)
        assert np.abs(sol.cpu().numpy() - ref).max() < 1e-9, fusion


def test_config2_full_size_against_the_oracle():
    """BASELINE config 2: spiral (examples/ode_demo.py:29-35), batch 4096 x 2, Dopri5 fp64 at odeint's default tolerances,
    t = linspace(0, 25, 10) and T = 2: every output row against the oracle; same attempt / accept counts."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(0)
    y0 = rng.uniform(-2, 2, size=(4096, 2))
    Wm = np.array([[-0.1, 2.0], [-2.0, -0.1]])
    f = rhs.CubicLinear(torch.tensor(Wm))
    for t in (np.linspace(0., 25., 10), np.array([0., 25.])):
        ref, st_ref = O.odeint(make_rhs('cubic_linear', {'W': Wm}), y0, t, method='dopri5', return_stats=True)
        sol = odeint(f, to_dev(y0), torch.tensor(t), method='dopri5')
        st = dict(odeint.last_stats)
        assert_band(sol.cpu(), ref, RTOL, ATOL, 'config 2 T=%d' % len(t))
        assert abs(st['n_attempts'] - st_ref.n_attempts) <= 1 and abs(st['n_accepted'] - st_ref.n_accepted) <= 1, (st, vars(st_ref))


def test_config3_full_size_tsit5_against_the_oracle_and_an_independent_integrator():
    """BASELINE config 3: Lorenz, batch 65536 x 3, Tsit5 rtol 1e-6 atol 1e-9, t = [0, 1].  `tsit5` here is the published
    tableau - the reference's own tsit5 is defective and cannot pin it - so two anchors are used:
    the oracle's corrected variant (same algorithm, numpy) on the FULL batch, and scipy's DOP853 at rtol 1e-12 on a slice
    (a different method, a different code base)."""
    from tfdiffeq_amd import odeint, rhs
    from scipy.integrate import solve_ivp
    rng1 = np.random.default_rng(1)
    y0 = np.array([1., 1., 1.]) + 1e-3 * rng1.standard_normal((65536, 3))
    t = np.array([0., 1.])
    f_np = make_rhs('lorenz', {'sigma': 10., 'beta': 8. / 3., 'rho': 28.})
    sol = odeint(rhs.Lorenz(), to_dev(y0), torch.tensor(t), rtol=1e-6, atol=1e-9, method='tsit5')
    st = dict(odeint.last_stats)
    ref, st_ref = O.odeint(f_np, y0, t, rtol=1e-6, atol=1e-9, method='tsit5', options={'tsit5_fixed': True}, return_stats=True)
    assert st['n_attempts'] == st_ref.n_attempts and st['n_accepted'] == st_ref.n_accepted, (st, vars(st_ref))
    assert_band(sol.cpu(), ref, RTOL, ATOL, 'config 3 vs oracle (corrected tsit5)')
    got = sol[1, :64].cpu().numpy()
    for i in range(64):
        r = solve_ivp(lambda tt, y: f_np(tt, y), (0., 1.), y0[i], method='DOP853', rtol=1e-12, atol=1e-14)
        # global error of a 5th-order pair at rtol 1e-6 over t in [0, 1] on Lorenz (|y| ~ 30): measured 1.0e-4; the band says
            # 'the same trajectory' - 'the same algorithm' is what the oracle check above establishes
        assert np.abs(got[i] - r.y[:, -1]).max() < 3e-4, (i, got[i], r.y[:, -1])


def test_config5_against_the_oracle_on_all_rows():
    """BASELINE config 5 as bench.py runs it: ODEFunc-shaped MLP 64-128-128-64 tanh, batch 32768, fp32, Dopri5 rtol = atol = 1e-3,
    t = [0, 1] - every row against the oracle run in float32 (same algorithm, numpy matmul / tanh; a few seconds)."""
    from tfdiffeq_amd import odeint, rhs
    gm = torch.Generator().manual_seed(4)

    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return (torch.rand(i, o, generator=gm) * 2 - 1) * lim
    W1, W2, W3 = glorot(64, 128), glorot(128, 128), glorot(128, 64)
    b1, b2, b3 = torch.zeros(128), torch.zeros(128), torch.zeros(64)
    y0 = torch.randn(32768, 64, generator=torch.Generator().manual_seed(5))
    mlp = rhs.MLPTanh(W1.to(dev()), b1.to(dev()), W2.to(dev()), b2.to(dev()), W3.to(dev()), b3.to(dev()))
    t = np.array([0., 1.])
    sol = odeint(mlp, y0.to(dev()), torch.tensor(t), rtol=1e-3, atol=1e-3, method='dopri5', options={'max_num_steps': 1000})
    st = dict(odeint.last_stats)
    w = {'W1': W1.numpy(), 'b1': b1.numpy(), 'W2': W2.numpy(), 'b2': b2.numpy(), 'W3': W3.numpy(), 'b3': b3.numpy()}
    ref, st_ref = O.odeint(make_rhs('mlp_tanh', weights=w, dtype=np.float32), y0.numpy(), t, rtol=1e-3, atol=1e-3, method='dopri5',
                           return_stats=True)
    assert (st['n_attempts'], st['n_accepted']) == (st_ref.n_attempts, st_ref.n_accepted), (st, vars(st_ref))
    assert_f32(sol.cpu(), ref, 'config5/oracle_all_rows')


# ---------------------------------------------------------------------------------------------
# gradients are never dropped silently (models/dense_odenet.py:178-186 differentiates through either branch)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('adjoint', [False, True])
def test_odenet_trains_the_ode_function_whatever_the_adjoint_flag(adjoint):
    from tfdiffeq_amd import models
    torch.manual_seed(0)
    net = models.ODENet(input_dim=4, hidden_dim=16, output_dim=2, non_linearity='tanh', adjoint=adjoint).to(dev())
    x = torch.randn(32, 4, device=dev())
    loss = net(x).pow(2).mean()
    loss.backward()
    for name, p in net.odeblock.odefunc.named_parameters():
        assert p.grad is not None and float(p.grad.abs().max()) > 0, name
    assert net.linear_layer.weight.grad is not None


def test_plain_odeint_with_grad_inputs_returns_gradients():
    from tfdiffeq_amd import odeint
    lin = torch.nn.Linear(3, 3).to(dev()).double()

    class F(torch.nn.Module):
        def forward(self, t, y):
            return torch.tanh(lin(y))
    f = F()
    f.lin = lin
    y0 = torch.randn(8, 3, device=dev(), dtype=torch.float64, requires_grad=True)
    with pytest.warns(UserWarning, match='adjoint'):
        out = odeint(f, y0, torch.tensor([0., 0.5]), rtol=1e-6, atol=1e-8)
    out[1].sum().backward()
    assert y0.grad is not None and float(y0.grad.abs().max()) > 0 and lin.weight.grad is not None
    # a plain callable: the reference back-propagates through it too (odeint.py:28-81 under a tape) - here dL/dy0 comes from the
    # adjoint solve of the parameterless system; y' = -y: y(0.5) = y0 e^-0.5, so d sum(y(0.5)) / d y0 = e^-0.5 everywhere
    y1 = torch.randn(8, 3, device=dev(), dtype=torch.float64, requires_grad=True)
    with pytest.warns(UserWarning, match='plain callable'):
        out = odeint(lambda t, y: -y, y1, torch.tensor([0., 0.5]), rtol=1e-9, atol=1e-11)
    out[1].sum().backward()
    assert float((y1.grad - np.exp(-0.5)).abs().max()) < 1e-7


def test_adjoint_of_a_forcing_only_rhs_has_zero_gradients():
    """adjoint.py:83-95 (UnconnectedGradients.ZERO): f does not depend on y or on any parameter."""
    from tfdiffeq_amd import odeint_adjoint

    class Forcing(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.unused = torch.nn.Parameter(torch.ones(2, dtype=torch.float64))

        def forward(self, t, y):
            return torch.ones_like(y) * 0.5
    f = Forcing().to(dev())
    y0 = torch.zeros(4, 2, device=dev(), dtype=torch.float64, requires_grad=True)
    out = odeint_adjoint(f, y0, torch.tensor([0., 1.0]), rtol=1e-6, atol=1e-9)
    out[1].sum().backward()
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), 0.5, atol=1e-9)
    np.testing.assert_allclose(y0.grad.cpu().numpy(), 1.0, atol=1e-9)          # dy(1)/dy0 = I
    assert float(f.unused.grad.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# the one-launch schedule next to somebody else's kernel
# ---------------------------------------------------------------------------------------------
def test_whole_call_kernel_next_to_a_competing_kernel_on_another_stream():
    """The whole-call kernels need all their workgroups resident at once.  A launch cannot promise that: here a long
    GEMM chain on a second stream holds the CUs while odeint is called.  Whatever the outcome - the grid is admitted
    late, or the first hand-off (the residency check, ~5 ms) gives up and the handle drops to one launch per attempt - the
    call must return promptly with the bits of the quiet run."""
    import time
    from tfdiffeq_amd import odeint, rhs
    g2 = torch.Generator().manual_seed(2)
    S = torch.randn(128, 128, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(128, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(128)
    y0 = torch.randn(16384, 128, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(dev())
    yl = to_dev(np.array([1., 1., 1.]) + 1e-3 * np.random.default_rng(1).standard_normal((65536, 3)))
    t = torch.tensor([0., 1.])
    f = rhs.Linear.from_matrix(A)
    quiet = odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
    quiet_l = odeint(rhs.Lorenz(), yl, t, rtol=1e-6, atol=1e-9, method='dopri5')
    assert dict(odeint.last_stats)['n_launches'] == 1
    side = torch.cuda.Stream()
    big = torch.randn(8192, 8192, device=dev())
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        acc = big
        for _ in range(40):                         # ~100 ms of matrix work spread over every CU
            acc = (acc @ big) * 1e-2
    t0 = time.perf_counter()
    busy = odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5')
    st = dict(odeint.last_stats)
    busy_l = odeint(rhs.Lorenz(), yl, t, rtol=1e-6, atol=1e-9, method='dopri5')
    st_l = dict(odeint.last_stats)
    wall = time.perf_counter() - t0
    torch.cuda.synchronize()
    assert st['status'] == 0 and st_l['status'] == 0, (st, st_l)
    assert torch.equal(busy, quiet) and torch.equal(busy_l, quiet_l)
    assert wall < 5.0, 'two odeint calls took %.2f s next to a competing kernel' % wall
    # and the handles still work (and still agree) once the chip is quiet again
    assert torch.equal(odeint(f, y0, t, rtol=1e-6, atol=1e-9, method='dopri5'), quiet)


# ---------------------------------------------------------------------------------------------
# DETEST on the product (tests/DETEST/detest.py, run.py): 25 known problems, Python callables -> plane-kernel engine
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', [c + i for c in 'ABCDE' for i in '12345'])
def test_detest_problem_on_the_plane_kernel_engine(name):
    """Every DETEST problem, t = 0 -> 20, dopri5 at tol 1e-3 and 1e-6, written with torch ops on the MI355X: y(20) inside
    the north star's band of the reference's result, NFE as the reference counted it (a decision may flip where the error
    ratio touches 1: +- one attempt)."""
    from tfdiffeq_amd import odeint
    from oracle import detest_problems as DP
    d, meta = load('fn_detest')
    y_like = torch.zeros(1, device=dev(), dtype=torch.float64)
    f, y0 = DP.problem(name, torch, like=y_like)
    class Counted(object):                     # the reference's own harness counts like this (tests/DETEST/run.py:14-22); evaluations
        def __init__(self):                    # that a recorded hipGraph replays do not run Python and are credited to `nfe` afterwards
            self.nfe = 0

        def __call__(self, t, y):
            self.nfe += 1
            return f(t, y)
    counted = Counted()
    nfe = [0]
    for tol, ref_nfe, att, acc in d[name + '_runs']:
        counted.nfe = 0
        sol = odeint(counted, y0, torch.tensor([0., DP.T_END], dtype=torch.float64), rtol=float(tol), atol=float(tol), method='dopri5')
        st = dict(odeint.last_stats)
        nfe[0] = counted.nfe
        ref = d['%s_y20_tol%g' % (name, tol)]
        got = sol[1].cpu().numpy()
        assert np.abs(got - ref).max() <= ATOL + RTOL * np.abs(ref).max() + 50 * tol * max(1.0, np.abs(ref).max()) * (st['n_attempts'] != att), \
            (name, tol, np.abs(got - ref).max())
        assert abs(nfe[0] - int(ref_nfe)) <= 6 and abs(st['n_attempts'] - int(att)) <= 1, (name, tol, nfe[0], ref_nfe, st)


# ---------------------------------------------------------------------------------------------
# gradients against an oracle: autograd THROUGH the torch-CPU restatement of the reference's solver
# ---------------------------------------------------------------------------------------------
def _mlp_cpu(sizes, seed):
    g = torch.Generator().manual_seed(seed)
    layers = []
    for i, o in zip(sizes[:-1], sizes[1:]):
        lin = torch.nn.Linear(i, o).double()
        with torch.no_grad():
            lin.weight.copy_((torch.rand(o, i, generator=g, dtype=torch.float64) * 2 - 1) * (6.0 / (i + o)) ** 0.5)
            lin.bias.copy_((torch.rand(o, generator=g, dtype=torch.float64) - 0.5) * 0.1)
        layers.append(lin)
    return layers


class _TanhMLP(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)

    def forward(self, t, y):
        h = y
        for i, lin in enumerate(self.layers):
            h = lin(h)
            if i + 1 < len(self.layers):
                h = torch.tanh(h)
        return h


def test_adjoint_gradients_match_autograd_through_the_reference_restatement():
    """The reference differentiates `odeint` by taping its eager ops (any caller under tf.GradientTape) and offers
    `odeint_adjoint` as the O(1)-memory alternative (adjoint.py:35-224); both give dL/dy0, dL/dtheta of the SAME loss up
    to the solver tolerance.  Oracle here: torch autograd through oracle/ode_torch_cpu.odeint_dopri5 - the op-for-op
    restatement of the reference's Dopri5 path, pinned by the golden fixtures - i.e. the taped gradient.  The product's
    adjoint (backward solve on the MI355X) must agree with it inside the adjoint's own tolerance."""
    from tfdiffeq_amd import odeint_adjoint
    from oracle import ode_torch_cpu as TC
    torch.manual_seed(0)
    cpu_net = _TanhMLP(_mlp_cpu([6, 24, 24, 6], 11))
    y0_cpu = torch.randn(40, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(12), requires_grad=True)
    t = [0.0, 0.6, 1.0]
    w = torch.randn(3, 40, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(13))
    sol_cpu, _ = TC.odeint_dopri5(cpu_net, y0_cpu, t, rtol=1e-8, atol=1e-10)
    (sol_cpu * w).sum().backward()
    import copy
    gpu_net = copy.deepcopy(cpu_net).to(dev())
    for p in gpu_net.parameters():
        p.grad = None
    y0_gpu = y0_cpu.detach().to(dev()).requires_grad_(True)
    sol_gpu = odeint_adjoint(gpu_net, y0_gpu, torch.tensor(t, dtype=torch.float64), rtol=1e-8, atol=1e-10, method='dopri5')
    (sol_gpu * w.to(dev())).sum().backward()
    np.testing.assert_allclose(sol_gpu.detach().cpu().numpy(), sol_cpu.detach().numpy(), rtol=1e-7, atol=1e-9)
    scale = float(y0_cpu.grad.abs().max())
    assert float((y0_gpu.grad.cpu() - y0_cpu.grad).abs().max()) < 1e-6 * scale
    for (n_, pc), pg in zip(cpu_net.named_parameters(), gpu_net.parameters()):
        s_ = max(float(pc.grad.abs().max()), 1e-12)
        assert float((pg.grad.cpu() - pc.grad).abs().max()) < 2e-6 * s_, n_


def test_odeblock_forward_and_gradients_against_the_oracle():
    """ODEBlock (dense_odenet.py:95-191: t = [0, 1], rtol = atol = tol, returns y(1)) on the fused MLP kernel against the
    torch-CPU restatement with the same weights (fp32 state: roundoff-limited band), and its parameter gradients (through the
    adjoint) against autograd through the restatement."""
    from tfdiffeq_amd import models
    from oracle import ode_torch_cpu as TC
    torch.manual_seed(3)
    block = models.ODEBlock(models.ODEFunc(16, 32, non_linearity='tanh'), tol=1e-4).to(dev())
    x = torch.randn(256, 16, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        got = block(x.to(dev()))
    import copy
    cpu_func = copy.deepcopy(block.odefunc).cpu()
    ref, _ = TC.odeint_dopri5(lambda t_, y_: cpu_func(t_, y_), x, [0., 1.], rtol=1e-4, atol=1e-4)
    assert_f32(got.cpu(), ref[1].detach(), 'odeblock_forward/fused_vs_restatement')
    # gradients: double precision so that the comparison measures the method, not fp32 roundoff
    blk64 = copy.deepcopy(block).double()
    cpu64 = copy.deepcopy(cpu_func).double()
    x64 = x.double()
    out = blk64(x64.to(dev()))
    out.pow(2).sum().backward()
    ref64, _ = TC.odeint_dopri5(lambda t_, y_: cpu64(t_, y_), x64, [0., 1.], rtol=1e-4, atol=1e-4)
    ref64[1].pow(2).sum().backward()
    for (n_, pg), pc in zip(blk64.odefunc.named_parameters(), cpu64.parameters()):
        s_ = max(float(pc.grad.abs().max()), 1e-12)
        assert float((pg.grad.cpu() - pc.grad).abs().max()) < 5e-3 * s_, (n_, float((pg.grad.cpu() - pc.grad).abs().max()), s_)


# ---------------------------------------------------------------------------------------------
# any dim <= 128 on the MFMA tile kernels (zero padded to the next tile width)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('dim', [3, 5, 17, 48, 100, 127])
def test_linear_rhs_of_any_dim_runs_on_the_padded_tile_kernels(dim, dtype):
    """f = y @ W (+ b) for dims that are not 16 / 32 / 64 / 128 (the reference takes any shape, rk_common.py:49-53): the
    whole-call / whole-attempt / fixed-grid tile kernels run them zero padded.  Against the oracle, against the VALU
    fallback kernel they replace as the default, and schedule against schedule (same bits)."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(100 + dim)
    S = rng.standard_normal((dim, dim))
    A = (-0.5 * np.eye(dim) + 0.5 * (S - S.T) / np.sqrt(dim)).astype(dtype)
    b = (0.1 * rng.standard_normal(dim)).astype(dtype)
    y0 = rng.standard_normal((1000 + dim, dim)).astype(dtype)           # ragged last tile too
    t = np.array([0., 0.3, 1.0])
    f64 = dtype == np.float64
    tol = dict(rtol=1e-6, atol=1e-9) if f64 else dict(rtol=1e-4, atol=1e-6)
    f = rhs.Linear(torch.tensor(A.T.copy()), torch.tensor(b))
    ref = O.odeint(lambda t_, y: y @ A.T + b, y0, t, method='dopri5', **tol)
    outs = {}
    for fusion in ('auto', 'step'):
        outs[fusion] = odeint(f, to_dev(y0), torch.tensor(t), method='dopri5', options={'fusion': fusion}, **tol)
        st = dict(odeint.last_stats)
        assert st['status'] == 0 and (st['n_launches'] == 1) == (fusion == 'auto'), (fusion, st)
    assert torch.equal(outs['auto'], outs['step'])
    valu = odeint(f, to_dev(y0), torch.tensor(t), method='dopri5', options={'linear_variant': 1}, **tol)
    if f64:
        assert np.abs(outs['auto'].cpu().numpy() - ref).max() < 1e-11
        assert float((outs['auto'] - valu).abs().max()) < 1e-11
    else:
        assert_f32(outs['auto'].cpu(), ref, 'padded_dim_%d/fused_vs_oracle' % dim)
        assert float((outs['auto'] - valu).abs().max()) < 1e-4
    # fixed grid (rk4, 3/8 rule) on the padded one-launch kernel vs the oracle
    tg = np.linspace(0., 1., 6)
    got = odeint(f, to_dev(y0), torch.tensor(tg), method='rk4')
    ref4 = O.odeint(lambda t_, y: y @ A.T + b, y0, tg.astype(dtype), method='rk4')
    assert np.abs(got.cpu().numpy() - ref4).max() < (1e-12 if f64 else 2e-5)


# ---------------------------------------------------------------------------------------------
# fixed-grid solvers on a grid of their own (step_size) and with eps: still one launch
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('method', ['rk4', 'euler'])
def test_fixed_grid_with_step_size_runs_in_one_launch_and_interpolates_like_the_reference(method):
    """FixedGridODESolver(step_size=...) (solvers.py:41-71: dead code in the reference, F7 - the oracle restates its intent) takes
    steps on its own uniform grid and interpolates the requested times linearly (solvers.py:106-115).  The one-launch kernels
    walk that grid themselves: row-local family (Lotka-Volterra) and MFMA-linear family, against the oracle and against the
    per-step plane-kernel loop."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(31)
    t = np.array([0., 0.33, 1.0, 1.7, 2.0])
    y0 = 1.0 + 0.5 * rng.uniform(size=(300, 2))
    opts = {'step_size': 0.0625 if method == 'rk4' else 0.03125}
    ref = O.odeint(make_rhs('lotka_volterra', {'a': 1.5, 'b': 1., 'c': 3., 'd': 1.}), y0, t, method=method, options=opts)
    got = odeint(rhs.LotkaVolterra(1.5, 1., 3., 1.), to_dev(y0), torch.tensor(t), method=method, options=opts)
    st = dict(odeint.last_stats)
    assert st['n_launches'] == 1, st
    assert np.abs(got.cpu().numpy() - ref).max() < 1e-13
    lv = rhs.LotkaVolterra(1.5, 1., 3., 1.)
    loop = odeint(lambda t_, y_: lv.forward(t_, y_), to_dev(y0), torch.tensor(t), method=method, options=opts)     # opaque callable: plane path
    assert float((loop - got).abs().max()) < 1e-13
    W = 0.3 * rng.standard_normal((16, 16))
    yl = rng.standard_normal((200, 16))
    ref = O.odeint(lambda t_, y: y @ W, yl, t, method=method, options=opts)
    got = odeint(rhs.Linear(torch.tensor(W)), to_dev(yl), torch.tensor(t), method=method, options=opts)
    assert dict(odeint.last_stats)['n_launches'] == 1
    assert np.abs(got.cpu().numpy() - ref).max() < 1e-12
    with pytest.raises(ValueError, match='exclusive'):                 # solvers.py:49-56: the reference rejects ANY grid_constructor
        odeint(rhs.LotkaVolterra(1.5, 1., 3., 1.), to_dev(y0), torch.tensor(t), method=method, options={'grid_constructor': lambda f, y, tt_: tt_})


# ---------------------------------------------------------------------------------------------
# dopri8 (13 rows) on the MFMA tile kernels of the linear RHS ("widen the MFMA family")
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dim,batch', [(128, 4096), (20, 1000), (64, 333)])
def test_dopri8_on_the_linear_tile_kernels(dim, batch):
    """dopri8.py:12-77 with f = y @ W + b: the whole-call and whole-attempt tile kernels are instantiated for the 13-row
    tableau too (14 stage derivatives of a 16-row tile in registers).  Against the oracle's dopri8, against the plane-kernel
    engine (the path this case took before), and schedule against schedule (same bits).
    Sharp comparison: a regime where every engine takes EXACTLY the same steps (first_step given, tolerance so loose that the
    step factor sits on its 1/ifactor clamp): the 13-stage arithmetic and the dense output then agree to roundoff.  At
    rtol = 1e-10 the error estimate is a 128-term cancellation at 1e-15: its noise moves dt by 1e-4 relative between
    accumulation orders (MFMA / rocBLAS / numpy), and dopri8's dense output is only the 4th-order quartic of interp.py,
    so interpolated values move by ~1e-7 - the plane-kernel engine differs from the oracle by as much."""
    from tfdiffeq_amd import odeint, rhs
    rng = np.random.default_rng(500 + dim)
    S = rng.standard_normal((dim, dim))
    A = -0.5 * np.eye(dim) + 0.5 * (S - S.T) / np.sqrt(dim)
    b = 0.1 * rng.standard_normal(dim)
    y0 = rng.standard_normal((batch, dim))
    f = rhs.Linear(torch.tensor(A.T.copy()), torch.tensor(b))
    fn = lambda t_, y: y @ A.T + b          # noqa: E731
    # (a) identical step sequences by construction: dt = 0.01, then 0.1
    t = np.array([0., 0.005, 0.05, 0.1])
    loose = dict(rtol=1e-3, atol=1e-3)
    ref = O.odeint(fn, y0, t, method='dopri8', options={'first_step': 0.01}, **loose)
    outs = {}
    for fusion in ('auto', 'step'):
        outs[fusion] = odeint(f, to_dev(y0), torch.tensor(t), method='dopri8', options={'fusion': fusion, 'first_step': 0.01}, **loose)
        st = dict(odeint.last_stats)
        assert st['status'] == 0 and st['n_attempts'] == 2 and (st['n_launches'] == 1) == (fusion == 'auto'), (fusion, st)
    assert torch.equal(outs['auto'], outs['step'])
    assert np.abs(outs['auto'].cpu().numpy() - ref).max() < 1e-13
    # (b) a demanding tolerance: same attempt sequence as the oracle and the plane-kernel engine, values within the band above
    t = np.array([0., 0.4, 1.0])
    tol = dict(rtol=1e-10, atol=1e-12)
    ref, rst = O.odeint(fn, y0, t, method='dopri8', return_stats=True, **tol)
    for fusion in ('auto', 'step'):
        outs[fusion] = odeint(f, to_dev(y0), torch.tensor(t), method='dopri8', options={'fusion': fusion}, **tol)
        st = dict(odeint.last_stats)
        assert st['status'] == 0 and (st['n_launches'] == 1) == (fusion == 'auto'), (fusion, st)
        assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted)
    assert torch.equal(outs['auto'], outs['step'])
    planes = odeint(f, to_dev(y0), torch.tensor(t), method='dopri8', options={'force_plane_kernels': True}, **tol)
    assert odeint.last_stats.get('engine') == 'plane kernels'
    assert np.abs(outs['auto'].cpu().numpy() - ref).max() < 1e-6
    assert float((outs['auto'] - planes).abs().max()) < 1e-6
    odeint(f, to_dev(y0), torch.tensor(t), method='dopri8', options={'fusion': 'stage'}, **tol)     # no per-stage kernels for 13 rows:
    assert str(odeint.last_stats.get('engine')).startswith(('plane kernels', 'device-controlled'))                                      # the generic engine takes it


@pytest.mark.gpu
@pytest.mark.parametrize('shape,act', [((4096, 64, 128), 'tanh'), ((777, 10, 48), 'relu'), ((300, 16, 16), 'softplus'), ((1000, 40, 16), 'tanh')])
def test_dopri8_on_the_mlp_tile_kernels(shape, act):
    """dopri8.py:12-77 with the ODEFunc network (dense_odenet.py:41-92): the whole-call and whole-attempt MLP kernels are
    instantiated for the 13-row tableau (VERDICT r02 "missing" 4).  (a) a regime where every engine takes EXACTLY the same steps
    (first_step given, loose tolerance: the step factor sits on its clamp): against the fp32 numpy restatement of the network under
    the oracle's dopri8, to a band set by fp32 roundoff of 13 stages; whole call == launch per attempt bit for bit;
    (b) a tighter tolerance against the plane-kernel engine running the same network as a torch callable."""
    from tfdiffeq_amd import odeint, rhs
    batch, d_, h_ = shape
    g = torch.Generator().manual_seed(77)

    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return (torch.rand(i, o, generator=g) * 2 - 1) * lim
    Ws = [glorot(d_, h_), glorot(h_, h_), glorot(h_, d_)]
    bs = [0.1 * torch.randn(h_, generator=g), 0.1 * torch.randn(h_, generator=g), 0.1 * torch.randn(d_, generator=g)]
    f = rhs.MLP(Ws[0].to(dev()), bs[0].to(dev()), Ws[1].to(dev()), bs[1].to(dev()), Ws[2].to(dev()), bs[2].to(dev()), activation=act)
    y0 = torch.randn(batch, d_, generator=g)
    Wn = [w.numpy() for w in Ws]
    bn = [b.numpy() for b in bs]
    actn = {'tanh': np.tanh, 'relu': lambda x: np.maximum(x, np.float32(0)),
            'softplus': lambda x: np.logaddexp(x, np.float32(0)).astype(np.float32)}[act]

    def fn(t_, y):
        h = actn(y @ Wn[0] + bn[0])
        h = actn(h @ Wn[1] + bn[1])
        return (h @ Wn[2] + bn[2]).astype(np.float32)
    t = np.array([0., 0.005, 0.05, 0.1])
    loose = dict(rtol=1e-2, atol=1e-2)
    ref, rst = O.odeint(fn, y0.numpy(), t, method='dopri8', options={'first_step': 0.01}, return_stats=True, **loose)
    outs = {}
    for fusion in ('auto', 'step'):
        outs[fusion] = odeint(f, y0.to(dev()), torch.tensor(t), method='dopri8', options={'fusion': fusion, 'first_step': 0.01}, **loose)
        st = dict(odeint.last_stats)
        assert st['status'] == 0 and (st['n_launches'] == 1) == (fusion == 'auto'), (fusion, st)
        assert (st['n_attempts'], st['n_accepted']) == (rst.n_attempts, rst.n_accepted), (fusion, st, rst)
    assert torch.equal(outs['auto'], outs['step'])
    assert_f32(outs['auto'].cpu().numpy(), ref, 'dopri8_mlp_%s_%dx%dx%d' % (act, batch, d_, h_))
    tt = torch.tensor([0., 0.4, 1.0])
    tol = dict(rtol=1e-5, atol=1e-6)
    a = odeint(f, y0.to(dev()), tt, method='dopri8', **tol)
    sa = dict(odeint.last_stats)
    assert sa['status'] == 0 and sa['n_launches'] == 1
    b = odeint(f, y0.to(dev()), tt, method='dopri8', options={'force_plane_kernels': True}, **tol)
    sb = dict(odeint.last_stats)
    assert sb.get('engine') == 'plane kernels' and abs(sa['n_attempts'] - sb['n_attempts']) <= 1
    assert (a - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())
    c = odeint(f, y0.to(dev()), -tt, method='dopri8', **tol)                # decreasing times
    dref = odeint(f, y0.to(dev()), -tt, method='dopri8', options={'force_plane_kernels': True}, **tol)
    assert (c - dref).abs().max().item() < 2e-5 * max(1.0, dref.abs().max().item())


# ---------------------------------------------------------------------------------------------
# time-dependent ODEFunc (dense_odenet.py:79-84: fc1 sees concat([t, x])) on the fused MLP kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('act,method', [('tanh', 'dopri5'), ('relu', 'dopri5'), ('softplus', 'tsit5'), ('tanh', 'bosh3')])
def test_time_dependent_odefunc_on_the_fused_kernel(act, method):
    """The stage time only shifts the first layer's bias by t * W1[0, :]: the network stays on the one-launch MFMA kernel.
    Checks: (a) the same solve through the plane-kernel engine with the torch module as a callable (same step sequence),
    (b) the torch-CPU restatement of the reference's Dopri5 path, (c) decreasing times (f <- -f(-t, y), misc.py:318-321),
    (d) the time dependence is real (a t-shifted solve differs)."""
    import copy
    from tfdiffeq_amd import models, odeint, odeint_adjoint
    from oracle import ode_torch_cpu as TC
    torch.manual_seed(41)
    func = models.ODEFunc(10, 48, time_dependent=True, non_linearity=act).to(dev())
    with torch.no_grad():
        func.fc1.weight[:, 0].mul_(3.0)                          # make the time column matter
    d = func.device_rhs()
    assert d is not None and d.time_dependent and d.dim == 10
    y0 = torch.randn(777, 10, generator=torch.Generator().manual_seed(42)).to(dev())
    t = torch.tensor([0.25, 0.9, 1.6])
    kw = dict(rtol=1e-5, atol=1e-6, method=method)
    with torch.no_grad():
        a = odeint(d, y0, t, **kw)
        sa = dict(odeint.last_stats)
        assert sa['status'] == 0 and sa['n_launches'] == 1
        b = odeint(func, y0, t, **kw)                             # python callable: plane kernels
        sb = dict(odeint.last_stats)
        assert abs(sa['n_attempts'] - sb['n_attempts']) <= 1
        assert (a - b).abs().max().item() < 3e-5 * max(1.0, b.abs().max().item())
        if method == 'dopri5':
            cpu = copy.deepcopy(func).cpu()
            ref, _ = TC.odeint_dopri5(lambda t_, y_: cpu(t_, y_), y0.cpu(), [0.25, 0.9, 1.6], rtol=1e-5, atol=1e-6)
            assert (a.cpu() - torch.stack([r.detach() for r in ref])).abs().max().item() < 5e-5 * max(1.0, b.abs().max().item())
        # launch per attempt == whole call, bit for bit
        c = odeint(d, y0, t, options={'fusion': 'step'}, **kw)
        assert torch.equal(a, c)
        # decreasing times
        tr = torch.tensor([1.6, 0.9, 0.25])
        ar = odeint(d, y0, tr, **kw)
        br = odeint(func, y0, tr, **kw)
        assert (ar - br).abs().max().item() < 3e-5 * max(1.0, br.abs().max().item())
        # a shifted time grid gives another answer
        shifted = odeint(d, y0, t + 1.0, **kw)
        assert (shifted[-1] - a[-1]).abs().max().item() > 1e-3
    # ODEBlock: inference on the fused kernel; training on the fused adjoint kernel too since round 4 (adj_t's derivative is
    # dot(w_t, .) of theta's b1 slice, tests/test_gpu_adjoint_fused.py)
    block = models.ODEBlock(func, tol=1e-4, adjoint=True, solver='dopri5')
    with torch.no_grad():
        out = block(y0)
    assert odeint.last_stats.get('n_launches') == 1
    x = y0.clone().requires_grad_(True)
    block(x).pow(2).sum().backward()
    assert odeint_adjoint.last_backward_stats['engine'].startswith('fused')
    assert x.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in func.parameters())
    assert float(func.fc1.weight.grad[:, 0].abs().max()) > 0.0
