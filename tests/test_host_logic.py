"""CPU tests: host logic of the product package, the C-ABI library's symbols and struct layout.
No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import warnings

import numpy as np
import pytest
import torch

import tfdiffeq_amd
from tfdiffeq_amd import _native as N
from tfdiffeq_amd import misc, odeint, rhs
from tfdiffeq_amd.bosh3 import _BOGACKI_SHAMPINE_TABLEAU, BS_C_MID
from tfdiffeq_amd.dopri5 import _DORMAND_PRINCE_SHAMPINE_TABLEAU, DPS_C_MID
from tfdiffeq_amd.tsit5 import _TSITOURAS_TABLEAU, _TSITOURAS_TABLEAU_PUBLISHED
from tests.golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tab_arrays(tb):
    S = len(tb.alpha)
    beta = np.zeros((S, S))
    for i, row in enumerate(tb.beta):
        beta[i, :len(row)] = row
    return np.asarray(tb.alpha, dtype=np.float64), beta, np.asarray(tb.c_sol, dtype=np.float64), \
        np.asarray(tb.c_error, dtype=np.float64)


def test_library_exports_every_declared_symbol_and_layout_matches():
    lib = N.load()                               # raises if a prototype is missing or a struct size differs
    header = open(os.path.join(ROOT, 'include', 'mi_ode.h')).read()
    declared = sorted(set(re.findall(r'^(?:int|int64_t|const char\*|const double\*)\s+(mi_ode_[a-z_0-9]+)\(', header, flags=re.M)))
    assert declared == list(N.EXPORTED_SYMBOLS), (declared, N.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mi_ode_abi_version() == N.ABI_VERSION == 13
    assert lib.mi_ode_sizeof(0) == C.sizeof(N.Desc) and lib.mi_ode_sizeof(1) == C.sizeof(N.Stats)
    assert lib.mi_ode_status_string(N.ST_MAX_STEPS).decode().startswith('max_num_steps exceeded')
    assert lib.mi_ode_status_string(N.ST_DT_UNDERFLOW).decode().startswith('underflow in dt')
    assert lib.mi_ode_status_string(N.ST_NONFINITE).decode().startswith('non-finite values in state')


def test_create_without_device_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    lib = N.load()
    d, h = N.Desc(), C.c_void_p()
    assert lib.mi_ode_create(C.byref(d), C.byref(h)) < 0
    assert 'HIP device' in N.last_error() or 'bad' in N.last_error()


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing somewhere else."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    for m in ('dopri5', 'rk4', 'euler', 'bosh3', 'tsit5'):
        with pytest.raises(N.NativeError):
            odeint(lambda t, y: -y, torch.ones(3, dtype=torch.float64), torch.tensor([0., 1.]), method=m)
        with pytest.raises(N.NativeError):
            odeint(rhs.Lorenz(), torch.ones(4, 3, dtype=torch.float64), torch.tensor([0., 1.]), method=m)


def test_product_does_not_import_the_oracle():
    import sys
    pkg = os.path.join(ROOT, 'tfdiffeq_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in src.replace('oracle/', ''), fn + ' mentions the oracle'
    assert not any(m.startswith('oracle') for m in sys.modules if 'tfdiffeq_amd' in m)


def test_tableaus_equal_the_reference_modules():
    d, _ = load('fn_tableaus')
    for name, tb in (('dopri5', _DORMAND_PRINCE_SHAMPINE_TABLEAU), ('tsit5', _TSITOURAS_TABLEAU),
                     ('bosh3', _BOGACKI_SHAMPINE_TABLEAU)):
        a, b, cs, ce = _tab_arrays(tb)
        assert np.array_equal(a, d[name + '_alpha']) and np.array_equal(b, d[name + '_beta'])
        assert np.array_equal(cs, d[name + '_c_sol']) and np.array_equal(ce, d[name + '_c_error'])
    assert np.array_equal(np.asarray(DPS_C_MID), d['dopri5_c_mid'])
    assert np.array_equal(np.asarray(BS_C_MID), d['bosh3_c_mid'])
    assert abs(sum(_TSITOURAS_TABLEAU_PUBLISHED.c_error)) < 1e-14          # the corrected tsit5 error weights
    assert abs(sum(_TSITOURAS_TABLEAU.c_error) - 0.9697) < 1e-3            # the reference's (F6a)


def test_solver_registry_and_api_errors():
    assert set(tfdiffeq_amd.SOLVERS) >= {'dopri5', 'tsit5', 'bosh3', 'euler', 'rk4'}
    y0, t = torch.ones(3, dtype=torch.float64), torch.tensor([0., 1.])
    with pytest.raises(ValueError):            # odeint.py:72-73
        odeint(lambda t_, y: -y, y0, t, options={'first_step': 0.1})
    with pytest.raises(KeyError):              # odeint.py:77
        odeint(lambda t_, y: -y, y0, t, method='not_a_solver')
    with pytest.raises(TypeError):             # misc.py:323-325
        odeint(lambda t_, y: -y, torch.ones(3, dtype=torch.int64), t)
    with pytest.raises(TypeError):             # misc.py:326-327
        odeint(lambda t_, y: -y, y0, torch.tensor([0, 1]))
    with pytest.raises(AssertionError):        # misc.py:305
        odeint(lambda t_, y: -y, [y0], t)
    with pytest.raises(AssertionError):        # misc.py:158-159 (before any device work)
        odeint(lambda t_, y: -y, y0, torch.tensor([0., 2., 1.]), method='dopri5')
    with warnings.catch_warnings(record=True) as w:     # misc.py:178-181
        warnings.simplefilter('always')
        try:
            odeint(lambda t_, y: -y, y0, t, method='dopri5', options={'bogus_option': 1})
        except N.NativeError:
            pass
        assert any('Unexpected arguments' in str(x.message) for x in w)


def test_check_inputs_reversal_and_tuple_lifting():
    f = rhs.Lorenz()
    y0 = torch.ones(2, 3, dtype=torch.float64)
    tensor_input, func, y0t, t = misc._check_inputs(f, y0, torch.tensor([3., 2., 1.]))
    assert tensor_input and isinstance(y0t, tuple) and torch.equal(t, torch.tensor([-3., -2., -1.]))
    assert func.device_rhs is not None and func.device_rhs.sign == -1.0 and f.sign == 1.0
    out = func(torch.tensor(-3.0, dtype=torch.float64), y0t)          # -f(-t, y)
    assert torch.allclose(out[0], -f.forward(torch.tensor(3.0), y0))
    # length-1 t is "decreasing" vacuously, like the reference (misc.py:153-155)
    _, func1, _, t1 = misc._check_inputs(f, y0, torch.tensor([5.]))
    assert float(t1[0]) == -5.0 and func1.device_rhs.sign == -1.0
    # tuple states keep plain callables
    g = lambda t_, ys: tuple(-y for y in ys)  # noqa: E731
    ti, func2, y2, _ = misc._check_inputs(g, (y0, y0), torch.tensor([0., 1.]))
    assert not ti and func2 is g and len(y2) == 2


def test_host_controller_matches_reference_vectors():
    """misc._optimal_step_size (host) against the vectors captured from the reference (F4 exponent included)."""
    d, meta = load('fn_step_controller')
    for order in (5, 3):
        for dt_name in ('float64', 'float32'):
            got = [misc._optimal_step_size(np.float64(meta['last_step']), (np.dtype(dt_name).type(r),), meta['safety'],
                                           meta['ifactor'], meta['dfactor'], order) for r in d['ratios']]
            np.testing.assert_allclose(got, d['misc_order%d_%s' % (order, dt_name)], rtol=1e-14, atol=0)


def test_convert_to_tensor_float32_detour():
    assert float(misc._convert_to_tensor(0.9, dtype=np.float64)) == 0.8999999761581421
    assert float(misc._convert_to_tensor(0.2, dtype=np.float64)) == 0.20000000298023224
    assert float(misc._convert_to_tensor(10.0, dtype=np.float64)) == 10.0


def test_device_rhs_python_paths_match_numpy():
    from oracle.rhs_numpy import make_rhs
    rng = np.random.default_rng(0)
    y = rng.standard_normal((5, 3))
    np.testing.assert_allclose(rhs.Lorenz()(None, torch.tensor(y)).numpy(),
                               make_rhs('lorenz', {'sigma': 10., 'beta': 8. / 3., 'rho': 28.})(None, y), rtol=1e-15)
    y2 = rng.standard_normal((5, 2))
    np.testing.assert_allclose(rhs.LotkaVolterra()(None, torch.tensor(y2)).numpy(),
                               make_rhs('lotka_volterra', {'a': 1.5, 'b': 1., 'c': 3., 'd': 1.})(None, y2), rtol=1e-15)
    A = rng.standard_normal((4, 4))
    y4 = rng.standard_normal((6, 4))
    np.testing.assert_allclose(rhs.Linear.from_matrix(torch.tensor(A))(None, torch.tensor(y4)).numpy(), y4 @ A.T, rtol=1e-13)
    np.testing.assert_allclose(rhs.CubicLinear(torch.tensor(A))(None, torch.tensor(y4)).numpy(), (y4 ** 3) @ A, rtol=1e-13)
    rev = rhs.Lorenz().reversed()
    np.testing.assert_allclose(rev(torch.tensor(1.0), torch.tensor(y)).numpy(), -rhs.Lorenz()(None, torch.tensor(y)).numpy())


def test_rhs_plugin_builds_and_exports_its_table():
    """rhs.CustomRowLocal: generated source, hipcc cross-compile, cache hit, table layout (no GPU needed)."""
    import ctypes as C
    import os
    import torch
    from tfdiffeq_amd import _plugin_build, rhs
    from tfdiffeq_amd import plugin_examples
    f = plugin_examples.van_der_pol(5.0)
    src = f.source(torch.float64)
    assert 'static constexpr int D = 2;' in src and 'MI_ODE_PLUGIN_F64' in src and 'k[1] = p[0]' in src
    path = _plugin_build.build(src)
    mtime = os.path.getmtime(path)
    assert _plugin_build.build(src) == path and os.path.getmtime(path) == mtime        # cache hit, no recompile
    lib, table = f._plugin(torch.float64)

    class Table(C.Structure):
        _fields_ = [('abi', C.c_int), ('dtype', C.c_int), ('dim', C.c_int), ('reserved', C.c_int), ('solver_size', C.c_size_t),
                    ('launch_init', C.c_void_p), ('launch_step', C.c_void_p), ('launch_fixed', C.c_void_p), ('persist_fn', C.c_void_p),
                    ('persist_planes_fn', C.c_void_p), ('multistep_fn', C.c_void_p)]        # (plugin ABI 2; ABI 3: the clock probe pointer in FixedArgs)
    tb = Table.from_address(table)
    core = N.load()
    assert tb.abi == 3 and tb.dtype == N.dtype_code(torch.float64) and tb.dim == 2
    assert tb.solver_size == core.mi_ode_sizeof(4)
    assert tb.launch_init and tb.launch_step and tb.launch_fixed and tb.persist_fn and tb.persist_planes_fn and tb.multistep_fn
    assert not lib.mi_ode_plugin_get(N.dtype_code(torch.float32))                      # built for one dtype only
    with pytest.raises(ValueError):
        rhs.CustomRowLocal(rhs.CustomRowLocal.MAX_DIM + 1, "k[0] = 0;")


# ---------------------------------------------------------------------------------------------
# round 2 host logic: tuple packing, adjoint parameter order, eligibility checks (no GPU needed)
# ---------------------------------------------------------------------------------------------
def test_tuple_components_are_packed_on_workgroup_boundaries():
    import torch
    from tfdiffeq_amd import _native as N
    from tfdiffeq_amd import solvers
    comps = (torch.arange(6.).reshape(3, 2), torch.arange(10.).reshape(5, 2) + 100., torch.arange(600.).reshape(300, 2) - 7.)
    packed, rows, offs = solvers._pack_components(comps, 2)
    assert rows == [3, 5, 300]
    assert offs == [0, N.SEGMENT_ALIGN, 2 * N.SEGMENT_ALIGN]
    assert packed.shape == (2 * N.SEGMENT_ALIGN + 2 * N.SEGMENT_ALIGN, 2)          # 300 rows round up to 512
    for c, r, o in zip(comps, rows, offs):
        assert torch.equal(packed[o:o + r], c)
    used = torch.zeros(packed.shape[0], dtype=torch.bool)
    for r, o in zip(rows, offs):
        used[o:o + r] = True
    assert float(packed[~used].abs().max()) == 0.0                                    # padding rows are zero
    # a trailing-shape component ([batch, k, dim]) is flattened row-wise
    packed3, rows3, _ = solvers._pack_components((torch.ones(4, 3, 2), torch.ones(1, 2)), 2)
    assert rows3 == [12, 1] and packed3.shape[0] == 2 * N.SEGMENT_ALIGN


def test_adjoint_parameter_order_round_trip():
    """The fused adjoint kernel returns adj_params as (W1 [in, out], b1, W2, b2, W3, b3); torch.nn.Linear keeps [out, in]."""
    import torch
    from tfdiffeq_amd import adjoint, models
    f = models.ODEFunc(3, 5, non_linearity='tanh')
    with torch.no_grad():
        for p in f.parameters():
            p.copy_(torch.randn_like(p))
    canonical = torch.cat([f.fc1.weight.t().reshape(-1), f.fc1.bias, f.fc2.weight.t().reshape(-1), f.fc2.bias,
                           f.fc3.weight.t().reshape(-1), f.fc3.bias]).detach()
    want = torch.cat([p.reshape(-1) for p in f.parameters()]).detach()
    assert torch.equal(adjoint.canonical_to_module_order(f, canonical), want)


def test_fused_paths_are_not_chosen_for_host_tensors_or_uncovered_functions():
    import torch
    from tfdiffeq_amd import adjoint, models, rhs, solvers
    f = models.ODEFunc(3, 5, non_linearity='tanh')
    cfg = dict(adjoint_method=None, adjoint_options=None, adjoint_rtol=1e-6, adjoint_atol=1e-9)
    wrapped = adjoint._TupleModule(f)
    like = torch.zeros(2, 4, 3)                                                       # CPU tensor: never the fused kernel
    assert adjoint._fused_plan(wrapped, 1, cfg, like, list(f.parameters())) is None
    assert adjoint._fused_plan(f, 1, cfg, like, list(f.parameters())) is None         # not the adjoint's own wrapper
    assert adjoint._fused_plan(wrapped, 2, cfg, like, list(f.parameters())) is None   # tuple state
    assert adjoint._fused_plan(wrapped, 1, dict(cfg, adjoint_method='rk4'), like, list(f.parameters())) is None
    # descriptors: activations outside the kernel's set have none; a time-dependent network has one (forward kernel only)
    td = models.ODEFunc(3, 5, time_dependent=True, non_linearity='tanh')
    d = td.device_rhs()
    assert d.time_dependent and d.dim == 3 and tuple(d.Ws[0].shape) == (4, 5)
    y = torch.randn(6, 3)
    assert torch.allclose(d(torch.tensor(0.7), y), td(torch.tensor(0.7), y).detach(), atol=1e-6)
    assert torch.allclose(d.reversed()(torch.tensor(-0.7), y), -td(torch.tensor(0.7), y).detach(), atol=1e-6)   # misc.py:318-321
    assert models.ODEFunc(3, 5, non_linearity='ELU').device_rhs() is None
    assert models.ODEFunc(3, 5).device_rhs().activation == 'relu'
    # tuple lift: only row-local systems, 2..8 components, device tensors
    lift = rhs.PerComponent(rhs.Lorenz())
    assert solvers._fusable_tuple(lift, (torch.zeros(4, 3), torch.zeros(2, 3))) is None          # host tensors
    assert solvers._fusable_tuple(rhs.Lorenz(), (torch.zeros(4, 3), torch.zeros(2, 3))) is None  # not lifted
    dense = rhs.PerComponent(rhs.Linear.from_matrix(torch.eye(4)))                               # not row-local: a plain callable
    assert solvers._fusable_tuple(dense, (torch.zeros(4, 4), torch.zeros(2, 4))) is None
    import pytest
    with pytest.raises(TypeError):
        rhs.PerComponent(lambda t, y: y)                                                         # not a DeviceRHS


def test_adams_scalars_match_the_oracle():
    """g (float32) and beta of adams.py:29-63 as tfdiffeq_amd/adams.py hands them to the plane kernels (and as csrc/mi_ode_adams_vc.h
    recomputes them on the device) against the oracle's g_and_explicit_phi, on random histories of every order."""
    import collections
    from oracle import adams_numpy as OA
    from tfdiffeq_amd import adams
    rng = np.random.default_rng(3)
    for order in range(1, 13):
        for _ in range(5):
            steps = rng.uniform(0.01, 0.2, size=order + 1)
            prev_t = collections.deque([np.float64(v) for v in (np.cumsum(steps)[:-1][::-1])], maxlen=13)   # newest first
            next_t = np.float64(np.cumsum(steps)[-1])
            phi = collections.deque([(np.full((2,), float(j + 1)),) for j in range(order)], maxlen=order)
            g_ref, ephi_ref = OA.g_and_explicit_phi(prev_t, next_t, phi, order)
            g, beta = adams.g_and_beta(prev_t, next_t, order)
            assert g.dtype == np.float32 and np.array_equal(g, g_ref)
            for j in range(order):                                   # explicit_phi_j = beta_j * phi_j
                assert np.array_equal(np.float64(beta[j]) * phi[j][0], ephi_ref[j][0]), (order, j)


def test_time_reversal_uses_the_callables_own_reversal_when_it_has_one():
    """misc.py:311-321: decreasing t -> t <- -t, func <- -func(-t, y).  A callable with `_mi_time_reversed()` (the linear system's augmented
    dynamics: the sign is a scale factor of its kernels) is asked for that function instead of being wrapped in a negation."""
    class Dyn(object):
        def __init__(self, sign=1.0):
            self.sign = sign

        def _mi_time_reversed(self):
            return Dyn(-self.sign)

        def __call__(self, t, y):
            return tuple(self.sign * (v + t) for v in y)
    y0 = (torch.ones(3), torch.zeros(2))
    plain = lambda t, y: tuple(v + t for v in y)  # noqa: E731
    for t in ([0.0, 1.0, 2.0], [2.0, 1.0, 0.5]):
        _, f_nat, _, t_nat = misc._check_inputs(Dyn(), y0, torch.tensor(t))
        _, f_wrp, _, t_wrp = misc._check_inputs(plain, y0, torch.tensor(t))
        assert torch.equal(t_nat, t_wrp) and bool((t_nat[1:] > t_nat[:-1]).all())
        decreasing = t[0] > t[-1]
        assert isinstance(f_nat, Dyn) and f_nat.sign == (-1.0 if decreasing else 1.0)
        assert isinstance(f_wrp, misc._ReverseFunc) == decreasing
        tau = torch.tensor(-1.25 if decreasing else 0.75)
        # the two formulations are the same function of (tau, y) when the dynamics do not depend on t - which is what a callable
        # promises by offering _mi_time_reversed() with a sign alone
        a = f_nat(torch.tensor(0.0), y0)
        b = tuple(-(v) for v in y0) if decreasing else y0
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert all(torch.equal(u, v) for u, v in zip(f_wrp(tau, y0), (tuple(-(v - tau) for v in y0) if decreasing else tuple(v + tau for v in y0))))


def test_recorded_attempt_key_follows_the_users_callable_through_the_wrappers():
    from tfdiffeq_amd import graph_step

    class F(object):
        def __call__(self, t, y):
            return y
    f = F()
    base, chain = graph_step._user_callable(misc._ReverseFunc(misc._TupleFunc(f)))
    assert base is f and chain == ('_ReverseFunc', '_TupleFunc', 'F')
    base2, chain2 = graph_step._user_callable(misc._TupleFunc(f))
    assert base2 is f and chain2 == ('_TupleFunc', 'F') and chain2 != chain          # reversed time records an attempt of its own
    m = torch.nn.Linear(2, 2)
    assert graph_step._user_callable(m)[0] is m                                      # (nn.Module: no `base` attribute to follow)


def test_linear_odefunc_descriptor_reads_the_parameters_own_storage():
    from tfdiffeq_amd import models
    f = models.LinearODEFunc(5, bias=True)
    d1 = f.device_rhs()
    assert d1 is f.device_rhs() and d1.dim == 5 and d1.kind == N.RHS_LINEAR
    assert d1.W.data_ptr() == f.weight.data_ptr() and d1.b.data_ptr() == f.bias.data_ptr()   # in-place optimizer steps stay visible
    with torch.no_grad():
        f.weight.mul_(0.5)
    assert f.device_rhs() is d1
    f.weight = torch.nn.Parameter(f.weight.detach().clone())                                # a rebound parameter: a new descriptor
    assert f.device_rhs() is not d1
    y = torch.randn(4, 5, dtype=torch.float64)
    assert torch.allclose(f(torch.tensor(0.0), y), y @ f.weight + f.bias)
    assert models.LinearODEFunc(3, bias=False).bias is None


def test_committed_bench_line_carries_the_contract():
    """The bench.py line committed under profiles/ (what DESIGN.md section 5 quotes) has every field of the bench contract, the metric
    BASELINE.json names, and figures that are consistent with each other."""
    import json
    line = open(os.path.join(ROOT, 'profiles', 'r05_bench_config4.json')).read().strip().split('\n')[-1]
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['unit'] == base.get('unit', d['unit']) and d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(d['value'] - 65536 * 128 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']          # whole-job throughput = elements / wall
    assert r['avg_launch_ms'] <= d['ms_per_step']                                                   # the kernel fits inside the step
    assert abs(r['achieved'] - r['algorithmic_flops_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e12) <= 1e-9 * r['achieved']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and d['parity_max_abs_diff'] < 1e-12
    # round 5: the transport that actually carried the controller records, the hand-off cost and the board's own telemetry ride along
    cfg = d['config']
    assert cfg['records_transport'] == 'single rank' and cfg['rccl_ranks'] == 0 and cfg['handoff_us'] > 0
    smi = cfg['smi']['sustained_loop']['cards'][0]
    assert smi['power_w']['max'] < smi['power_cap_w']['min'] and 2000 < smi['sclk_mhz']['max'] <= 2400


def test_bench_refuses_more_ranks_than_devices_with_one_line():
    """`bench.py --gpus N` with fewer than N devices visible ends at once with exit status 3 and ONE line that says why (round-4 review, item 4c:
    the first 8-GPU contact must not end as eight stack traces or a hang).  No JSON line is printed."""
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('two or more devices visible')
    env = dict(os.environ)
    env.pop('BENCH_SHARE_GPU', None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 3
    assert res.stdout.strip() == ''
    lines = [ln for ln in res.stderr.strip().split('\n') if ln.startswith('bench.py:')]
    assert len(lines) == 1 and '--gpus 2' in lines[0] and 'device(s) visible' in lines[0]


def test_a_users_own_time_reversed_attribute_is_not_taken_for_the_native_hook():
    """(advisor, round 4) only the private `_mi_time_reversed` is the hook; a user's callable that happens to have a `time_reversed` attribute is
    wrapped like any other."""
    class Dyn(object):
        def time_reversed(self):
            raise AssertionError('not ours to call')

        def __call__(self, t, y):
            return tuple(v + t for v in y)
    _, f, _, _ = misc._check_inputs(Dyn(), (torch.ones(2),), torch.tensor([1.0, 0.0]))
    assert isinstance(f, misc._ReverseFunc)


def test_trainable_leaves_of_a_plain_callable_come_from_its_autograd_graph():
    """(advisor, rounds 4 and 5) `odeint(lambda t, y: net(y), ...)`: the tensors the adjoint differentiates with respect to are the
    grad-requiring LEAVES one probe evaluation depends on (a tensor reached through an attribute chain or a container is found) IN UNION with
    what the callable can name (a branch the probe did not take keeps its gradient; a module merely named gets a zero gradient).  Nothing is
    cached between calls: a rebound network is seen as it is now."""
    import importlib
    OD = importlib.import_module('tfdiffeq_amd.odeint')      # (the package exports the FUNCTION under the same name)
    torch.manual_seed(0)
    used, unused = torch.nn.Linear(3, 3).double(), torch.nn.Linear(3, 3).double()
    holder = {'deep': [torch.randn(3, dtype=torch.float64, requires_grad=True)]}
    frozen = torch.randn(3, dtype=torch.float64)
    calls = []

    def make():
        def f(t, y):
            calls.append(1)
            _ = unused                                     # named, never evaluated
            return used(y) + holder['deep'][0] * t + frozen
        return f
    y0 = torch.randn(4, 3, dtype=torch.float64)
    leaves = OD._graph_leaves(make(), y0, torch.tensor([0.0, 1.0]))
    assert [id(x) for x in leaves[:3]] == [id(x) for x in leaves[:3]] and {id(x) for x in leaves[:3]} == {id(used.weight), id(used.bias), id(holder['deep'][0])}
    assert {id(x) for x in leaves[3:]} == {id(unused.weight), id(unused.bias)}           # named only: differentiated too (zero gradient)
    n = len(calls)
    again = OD._graph_leaves(make(), y0, torch.tensor([0.0, 1.0]))
    assert [id(x) for x in again] == [id(x) for x in leaves] and len(calls) == n + 1    # probed again: no cache to go stale
    assert OD._wants_grad(make(), y0) is True and OD._wants_grad(lambda t, y: y * 2.0, y0) is False
    # the advisor's two reproductions: a network reached through a rebound name, and a branch the probe at t[0] does not take
    box = {'net': torch.nn.Linear(3, 3).double()}
    f_box = lambda t, y: box['net'](y)                     # noqa: E731
    first = OD._graph_leaves(f_box, y0, torch.tensor([0.0, 1.0]))
    box['net'] = torch.nn.Linear(3, 3).double()
    second = OD._graph_leaves(f_box, y0, torch.tensor([0.0, 1.0]))
    assert {id(x) for x in second} == {id(box['net'].weight), id(box['net'].bias)} and not ({id(x) for x in first} & {id(x) for x in second})
    nA, nB = torch.nn.Linear(3, 3).double(), torch.nn.Linear(3, 3).double()
    both = OD._graph_leaves(lambda t, y: nA(y) if t < 0.5 else nB(y), y0, torch.tensor([0.0, 1.0]))
    assert {id(x) for x in both} == {id(p) for p in list(nA.parameters()) + list(nB.parameters())}
    # a tuple state and a callable object
    class Obj(object):
        def __init__(self):
            self.w = torch.ones(3, dtype=torch.float64, requires_grad=True)

        def __call__(self, t, y):
            return (y[0] * self.w, y[1])
    o = Obj()
    lv = OD._graph_leaves(o, (y0, y0.clone()), torch.tensor([0.0, 1.0]))
    assert len(lv) == 1 and lv[0] is o.w


def test_cooperative_right_hand_sides_host_logic():
    """Round 5, no GPU needed: which kernel family an ODEFunc-shaped network is offered (tile kernels / cooperative kernel / callable), and
    the translation unit rhs.CustomCoop generates (a cooperative plugin: one state element per thread)."""
    from tfdiffeq_amd import rhs
    mk = lambda d, h, dt: rhs.MLP(torch.zeros(d, h, dtype=dt), None, torch.zeros(h, h, dtype=dt), None, torch.zeros(h, d, dtype=dt), None)   # noqa: E731
    small32 = mk(64, 128, torch.float32)
    y = torch.zeros(10, 64)
    assert small32.supports(y) and not small32.supports_coop(y) and small32.supports_multistep(y)         # the MFMA tile kernels' box
    f64 = mk(64, 128, torch.float64)
    y64 = torch.zeros(1000, 64, dtype=torch.float64)
    assert f64.supports(y64) and not f64.coop_in_box(y64) and f64.multistep_fused          # (round 6: float64 tile kernels)
    big = torch.zeros(4096, 64, dtype=torch.float64)                                                     # 1.3e8 multiply-adds per evaluation
    assert f64.supports(big)
    wide = mk(100, 300, torch.float32)
    assert not wide.supports(torch.zeros(5, 100)) and not wide.coop_in_box(torch.zeros(5, 100)) and not wide.multistep_fused
    with pytest.raises(ValueError):
        rhs.CustomCoop(300, 'k = y[i];')
    with pytest.raises(ValueError):
        rhs.CustomCoop(8, 'k = y[i];', tensors=[torch.zeros(1)] * 4)
    c = rhs.CustomCoop(100, 'k = p[0] * (y[(i + 1) % DIM] - y[i]);', params=[0.5])
    src = c.source(torch.float64)
    for needle in ('static constexpr int DIM = 100;', 'static constexpr bool kCoop = true;', 'MI_ODE_DEFINE_COOP_PLUGIN(mi::RhsUserCoop)',
                   'MI_ODE_PLUGIN_F64', 'k = p[0] * (y[(i + 1) % DIM] - y[i]);', 'return 256 / DIM;'):
        assert needle in src, needle
    assert c.supports(torch.zeros(3, 100, dtype=torch.float64)) and c.multistep_fused and c.fixed_grid_fused and not c.row_local
    with pytest.raises(NotImplementedError):
        c(torch.tensor(0.0), torch.zeros(3, 100))                        # no torch_fn: only the kernels can evaluate it


def test_plan_names_the_engine_of_the_five_baseline_configurations():
    """odeint.plan (round-5 review, item 9): the engine a call WILL take and the predicate that chose it, without a GPU - pinned for the
    five BASELINE.json configurations, as device right-hand sides and as the Python callables the reference's users write."""
    import numpy as np
    f64, f32 = torch.float64, torch.float32
    AUTO = {'lower': 'auto'}          # (this module's other tests keep Python callables on the callable engines: tests/conftest.py)
    # 1: Lotka-Volterra 2-D, fixed-step RK4, 1000 steps
    p = odeint.plan(rhs.LotkaVolterra(), torch.ones(1, 2, dtype=f64), method='rk4')
    assert p['engine'] == 'fused' and p['kernel'].startswith('k_fixed_rowlocal<double') and p['launches'] == 'one per call'
    # 2: spiral, batch 4096 x 2, Dopri5 float64
    W = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=f64)
    p = odeint.plan(rhs.CubicLinear(W), torch.ones(4096, 2, dtype=f64), method='dopri5')
    assert p['engine'] == 'fused' and p['kernel'].startswith('k_persist_rowlocal<double, 6') and 'row_local' in p['why']
    p2 = odeint.plan(lambda t, y: torch.matmul(y ** 3, W), torch.ones(4096, 2, dtype=f64), method='dopri5', options=AUTO)
    assert p2['lower'] == {'lowered': True, 'kind': 'rowlocal', 'dim': 2, 'batch_axes': 1} and p2['kernel'] == p['kernel']
    # 3: Lorenz 65536 x 3, Tsit5
    p = odeint.plan(rhs.Lorenz(), torch.ones(65536, 3, dtype=f64), method='tsit5')
    assert p['kernel'].startswith('k_persist_rowlocal<double, 6') and p['state'] == '65536 x 3 float64'
    # 4: linear 65536 x 128, Dopri5 float64 (the headline)
    A = torch.tensor(np.random.RandomState(0).randn(128, 128))
    p = odeint.plan(rhs.Linear(A), torch.ones(65536, 128, dtype=f64), method='dopri5')
    assert p['engine'] == 'fused' and p['kernel'].startswith('k_persist_linear_mfma<double, 128, 6>') and 'MFMA' in p['why']
    p2 = odeint.plan(lambda t, y: y @ A, torch.ones(65536, 128, dtype=f64), method='dopri5', options=AUTO)
    assert p2['lower']['kind'] == 'linear' and p2['kernel'] == p['kernel']
    assert odeint.plan(rhs.Linear(A), torch.ones(65536, 128, dtype=f64), method='dopri5', options={'fusion': 'stage'})['kernel'].startswith('k_stage_linear_mfma')
    # 5: ODENet MLP 64-128-128-64 tanh, 32768 x 64 float32
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Tanh(), torch.nn.Linear(128, 128), torch.nn.Tanh(), torch.nn.Linear(128, 64))
    p = odeint.plan(rhs.from_sequential(net), torch.ones(32768, 64, dtype=f32), method='dopri5')
    assert p['engine'] == 'fused' and p['kernel'].startswith('k_persist_mlp<DP, HP, 0, 6>')
    p2 = odeint.plan(lambda t, y: net(y), torch.ones(32768, 64, dtype=f32), method='dopri5', options=AUTO)
    assert p2['lower']['kind'] == 'mlp' and p2['kernel'] == p['kernel']
    # ... and the ways OUT of the fused engine say why
    p = odeint.plan(rhs.from_sequential(net.double()), torch.ones(32768, 64, dtype=f64), method='dopri5')
    assert p['engine'] == 'fused' and p['kernel'].startswith('k_persist_mlp64<DP, HP, 0, 6>')      # (round 6: the float64 tile kernels)
    wide = torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Tanh(), torch.nn.Linear(300, 300), torch.nn.Tanh(), torch.nn.Linear(300, 64))
    p = odeint.plan(rhs.from_sequential(wide), torch.ones(32768, 64, dtype=f32), method='dopri5')
    assert p['engine'] == 'callable' and 'supports(y0) is False' in p['why']
    p = odeint.plan(lambda t, y: torch.cumsum(y, -1), torch.ones(8, 3, dtype=f64), method='dopri5', options=AUTO)
    assert p['engine'] == 'callable' and p['lower'] == {'lowered': False, 'why': 'operation `cumsum` is outside the op set'}
    p = odeint.plan(rhs.Lorenz(), torch.ones(8, 3, dtype=f64), method='midpoint')
    assert p['engine'] == 'plane kernels' and 'midpoint has no fused kernel' in p['why']
    A256 = torch.eye(256, dtype=f64)
    p = odeint.plan(rhs.Linear(A256), torch.ones(65536, 256, dtype=f64), method='dopri5')              # round 6: W streamed, 256-wide tiles
    assert p['kernel'].startswith('k_persist_linear_mfma<double, 256, 6>') and 'streamed' in p['why']
    p = odeint.plan(rhs.Linear(A256[:200, :200].contiguous()), torch.ones(8, 200, dtype=f32), method='bosh3')
    assert p['kernel'].startswith('k_persist_linear_mfma<float, 256, 3>')
    assert odeint.plan(rhs.Linear(A256), torch.ones(8, 256, dtype=f64), method='dopri5', options={'fusion': 'stage'})['kernel'] == 'k_stage_linear_valu'
    assert odeint.plan(rhs.Linear(A256), torch.ones(8, 256, dtype=f64), method='rk4')['kernel'].startswith('k_fixed_linear_mfma')
    p = odeint.plan(rhs.CubicLinear(torch.eye(64, dtype=f64)), torch.ones(512, 64, dtype=f64), method='dopri5')      # (y ** 3) @ W beyond 2 x 2: where pick_family puts it
    assert p['kernel'] == 'k_stage_linear_valu' and 'no cube' in p['why']
    p = odeint.plan(rhs.Linear(A), torch.ones(64, 128, dtype=f64), method='adaptive_heun')
    assert p['engine'] == 'callable' and '1-row tableau' in p['why']
    p = odeint.plan(lambda t, y: y @ A, torch.ones(64, 128, dtype=f64), method='adaptive_heun', options=AUTO)       # lowered: generated cooperative code
    assert p['engine'] == 'fused' and p['lower']['kind'] == 'coop'
    p = odeint.plan(rhs.Lorenz(), torch.ones(8, 3, dtype=f64), method='adams')
    assert p['kernel'].startswith('k_adams_vc_rowlocal<double')
