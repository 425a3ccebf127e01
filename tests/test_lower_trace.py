"""CPU tests of the tracer / code generator that lowers Python callables onto the fused kernels (tfdiffeq_amd/lower.py).
No GPU: the graph is evaluated in numpy, the generated statements are compiled by g++ as a host function - both against the callable
itself evaluated by torch on the CPU."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lower_cases as LC                                  # noqa: E402
from tfdiffeq_amd import lower as L                       # noqa: E402


def _rows(tr, y0):
    return y0.reshape((-1,) + tr.tail) if tr.nb else y0.reshape((1,) + tr.tail)


@pytest.mark.parametrize('name', sorted(LC.CASES))
def test_graph_evaluates_like_the_callable(name):
    f, y0, kind = LC.CASES[name]('cpu')
    tr = L.trace(f, y0)
    assert L.classify(tr)[0] == kind
    rows = _rows(tr, y0)
    for tt in (0.7, 2.3):
        ref = f(torch.tensor(tt, dtype=y0.dtype), y0).detach().reshape(rows.shape).double().numpy()
        for i in range(min(3, rows.shape[0])):
            got = L.evaluate_row(tr, tt, rows[i].double().numpy())
            tol = 1e-12 if y0.dtype == torch.float64 else 2e-5
            with np.errstate(invalid='ignore'):
                bad = np.abs(got - ref[i]) > tol * (1 + np.abs(ref[i]))
            assert not bad.any(), (name, got, ref[i])


def _host_fn(src, tag):
    d = tempfile.mkdtemp(prefix='lower_host_')
    cpp, so = os.path.join(d, tag + '.cpp'), os.path.join(d, tag + '.so')
    with open(cpp, 'w') as fh:
        fh.write(src)
    res = subprocess.run(['g++', '-O1', '-ffp-contract=off', '-shared', '-fPIC', cpp, '-o', so], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return C.CDLL(so)


ROW_CASES = sorted(n for n, mk in LC.CASES.items() if mk('cpu')[2] == 'rowlocal')


@pytest.mark.parametrize('name', ROW_CASES)
def test_generated_statements_compile_and_agree(name):
    """The one-trajectory-per-thread body, compiled as host C++: elementwise systems agree with torch to the last bits (the
    statements follow the Python expression operation for operation)."""
    f, y0, _ = LC.CASES[name]('cpu')
    tr = L.trace(f, y0)
    lib = _host_fn(L.host_source(tr), name)
    lay = L.Layout(tr)
    f64 = y0.dtype == torch.float64
    ct, npt = (C.c_double, np.float64) if f64 else (C.c_float, np.float32)
    pool = np.zeros(max(lay.size, 1), dtype=npt)
    for idx, off in enumerate(lay.tensor_off):
        e = tr.tensors[idx]
        x = e['t'].detach()
        x = x[(0,) * e['lead']] if e['lead'] else x
        pool[off:off + x.numel()] = x.reshape(-1).to(torch.float64).numpy().astype(npt)
    pool[lay.extra_off:lay.extra_off + max(len(tr.scalars) - 8, 0)] = np.asarray(tr.scalars[8:], dtype=npt)
    ps = (C.c_double * 8)(*(list(tr.scalars[:8]) + [0.0] * (8 - min(len(tr.scalars), 8))))
    fn = lib.rhs_f64 if f64 else lib.rhs_f32
    rows = _rows(tr, y0)
    dim = int(np.prod(tr.tail)) if tr.tail else 1
    tt = 0.7
    ref = f(torch.tensor(tt, dtype=y0.dtype), y0).detach().reshape(rows.shape[0], dim).numpy()
    for i in range(min(4, rows.shape[0])):
        yin = np.ascontiguousarray(rows[i].reshape(-1).numpy())
        out = np.zeros(dim, dtype=npt)
        fn(ct(tt), yin.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), ps, pool.ctypes.data_as(C.c_void_p))
        tol = 1e-13 if f64 else 1e-5
        with np.errstate(invalid='ignore'):
            bad = np.abs(out - ref[i]) > tol * (1 + np.abs(ref[i]))
        assert not bad.any(), (name, out, ref[i])


def test_generated_text_for_the_lorenz_system():
    """Source in, expected body out: the statements are the Python expression in its own order, floats as by-value parameters."""
    s, b, r = 10., 8. / 3., 28.

    def lorenz(t, y):
        return torch.stack([s * (y[1] - y[0]), y[0] * (r - y[2]) - y[1], y[0] * y[1] - b * y[2]])
    tr = L.trace(lorenz, torch.zeros(3, dtype=torch.float64))
    assert tr.scalars == [10.0, 28.0, 8. / 3.]
    assert L.rowlocal_body(tr).splitlines() == [
        'const T v0 = y[1] - y[0];', 'const T v1 = p[0] * v0;', 'const T v2 = p[1] - y[2];', 'const T v3 = y[0] * v2;', 'const T v4 = v3 - y[1];',
        'const T v5 = y[0] * y[1];', 'const T v6 = p[2] * y[2];', 'const T v7 = v5 - v6;', 'k[0] = v1;', 'k[1] = v4;', 'k[2] = v7;']


OPS = {
    'add_sub_mul_div': (lambda t, y: (y + 2) * (y - 1.5) / (y * y + 1), 'const T v0 = y[0] + (T)2;'),
    'rsub_rdiv': (lambda t, y: (1 - y) + 2 / (y * y + 3), 'const T v0 = (T)1 - y[0];'),
    'neg_abs': (lambda t, y: -torch.abs(y), 'fabs(y[0])'),
    'pow_int': (lambda t, y: y ** 2 + y ** 3 + y ** 5, 'pow(y[0], (T)5)'),
    'pow_half': (lambda t, y: (y * y + 1) ** 0.5 + (y * y + 1) ** -1 + (y * y + 1) ** 1.5, 'sqrt('),
    'rpow': (lambda t, y: 2 ** y, 'pow((T)2, y[0])'),
    'sin_cos_exp_tanh': (lambda t, y: torch.sin(y) + torch.cos(y) * torch.exp(-y * y) + torch.tanh(y), 'tanh(y[0])'),
    'methods': (lambda t, y: y.sin() + y.cos().mul(2.5).add(y, alpha=2), 'sin(y[0])'),
    'relu_softplus_sigmoid': (lambda t, y: torch.relu(y) + torch.nn.functional.softplus(y) + torch.sigmoid(y), 'log1p(exp('),
    'where_cmp': (lambda t, y: torch.where(y > 0, y, 0.1 * y), '? y[0] :'),
    'clamp_max': (lambda t, y: torch.clamp(y, -0.5, 0.5) + torch.maximum(y, torch.zeros_like(y)), 'fmax('),
    'time': (lambda t, y: y * torch.cos(t) + t ** 2, 'cos(t)'),
    'index_stack': (lambda t, y: torch.stack([y[..., 1], -y[..., 0], y[..., 2] * y[..., 0]], dim=-1), 'k[0] = y[1];'),
    'slice_cat': (lambda t, y: torch.cat([y[..., 1:], y[..., :1]], dim=-1), 'k[2] = y[0];'),
    'roll_flip': (lambda t, y: torch.roll(y, 1, -1) - torch.flip(y, [-1]), 'y[2] - y[2]'),
    'sum_mean_keepdim': (lambda t, y: y - y.sum(-1, keepdim=True) + y.mean(dim=-1, keepdim=True), '/ (T)3'),
    'unbind_chunk': (lambda t, y: torch.stack([y.unbind(-1)[2], y.chunk(3, -1)[0][..., 0], y[..., 1]], -1), 'k[0] = y[2];'),
    'matmul_const': (lambda t, y: y @ torch.arange(9, dtype=torch.float64).reshape(3, 3), 'cw[3]'),
    'linear_module': (lambda t, y, _l=torch.nn.Linear(3, 3).double(): _l(y), 'cw[9]'),
    'unsqueeze_squeeze': (lambda t, y: (y.unsqueeze(-1) * 2).squeeze(-1), 'y[0] * (T)2'),
    'reshape_view': (lambda t, y: y.reshape(y.shape[0], 3, 1).view(-1, 3) * 3, '(T)3'),
    'const_vector': (lambda t, y: y * torch.tensor([1., 2., 3.], dtype=torch.float64), 'y[1] * cw[1]'),
    'norm': (lambda t, y: y / torch.linalg.norm(y, dim=-1, keepdim=True) if False else y / y.norm(dim=-1, keepdim=True), 'sqrt('),
}


@pytest.mark.parametrize('name', sorted(OPS))
def test_every_traced_operation(name):
    f, fragment = OPS[name]
    y0 = torch.tensor(np.random.RandomState(0).randn(5, 3) * 0.7)
    tr = L.trace(f, y0)
    assert tr.nb == 1 and tr.tail == (3,)
    body = L.rowlocal_body(tr)
    assert fragment in body, body
    ref = f(torch.tensor(0.9, dtype=torch.float64), y0).detach().numpy()
    for i in range(5):
        np.testing.assert_allclose(L.evaluate_row(tr, 0.9, y0[i].numpy()), ref[i], rtol=1e-13, atol=1e-13)
    lib = _host_fn(L.host_source(tr), 'op_' + name)
    lay = L.Layout(tr)
    pool = np.zeros(max(lay.size, 1))
    for idx, off in enumerate(lay.tensor_off):
        x = tr.tensors[idx]['t'].detach()
        pool[off:off + x.numel()] = x.reshape(-1).double().numpy()
    ps = (C.c_double * 8)(*(list(tr.scalars[:8]) + [0.0] * (8 - len(tr.scalars[:8]))))
    for i in range(5):
        yin, out = np.ascontiguousarray(y0[i].numpy()), np.zeros(3)
        lib.rhs_f64(C.c_double(0.9), yin.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), ps, pool.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(out, ref[i], rtol=1e-13, atol=1e-13)


REFUSED = {
    'control_flow': (lambda t, y: y if t > 0.5 else -y, 'data-dependent control flow'),
    'host_read': (lambda t, y: y * float(t), 'reads a traced value on the host'),
    'in_place': (lambda t, y: y.mul_(2), 'in-place'),
    'batch_mixing': (lambda t, y: y - y.mean(0), 'batch ax'),
    'batch_flip': (lambda t, y: torch.flip(y, [0]), 'batch ax'),
    'unknown_op': (lambda t, y: torch.cumsum(y, -1), 'outside the op set'),
    'cast': (lambda t, y: y.float().double(), 'cast'),
    'numpy': (lambda t, y: y * np.sin(t), None),
    'wrong_shape': (lambda t, y: y[..., 0], 'result of shape'),
}


@pytest.mark.parametrize('name', sorted(REFUSED))
def test_refusals_say_why(name):
    f, fragment = REFUSED[name]
    y0 = torch.zeros(70, 200, dtype=torch.float64)          # (too large to be ONE system: the batch axis cannot be given up)
    try:
        L.trace(f, y0)
    except L.TraceError as e:
        assert fragment is None or fragment in str(e), str(e)
    except Exception:
        assert fragment is None                             # the callable itself failed on the proxies: odeint reports "tracing failed"
    else:
        raise AssertionError('traced')


def test_batch_axes_are_given_up_only_for_small_systems():
    f = lambda t, y: torch.stack([y[1], -y[0]])              # noqa: E731  (indexes the FIRST axis)
    tr = L.trace(f, torch.zeros(2, dtype=torch.float64))
    assert tr.nb == 0 and tr.tail == (2,)
    tr = L.trace(lambda t, y: torch.stack([y[1], -y[0]]), torch.zeros(2, 5, dtype=torch.float64))   # a [2, 5] state as ONE system of 10
    assert tr.nb == 0 and tr.tail == (2, 5)
    with pytest.raises(L.TraceError):
        L.trace(lambda t, y: torch.stack([y[1], -y[0]]), torch.zeros(2, 5000, dtype=torch.float64))


def test_constants_are_read_on_every_call_and_code_depends_on_structure_only():
    class F(object):
        def __init__(self):
            self.a = 2.0
            self.w = torch.tensor([1., 2.], dtype=torch.float64)

        def __call__(self, t, y):
            return self.a * y * self.w
    f = F()
    y0 = torch.zeros(4, 2, dtype=torch.float64)
    t1 = L.trace(f, y0)
    f.a = 3.5
    f.w.mul_(2)
    t2 = L.trace(f, y0)
    assert t1.key() == t2.key() and t1.scalars == [2.0] and t2.scalars == [3.5]
    np.testing.assert_allclose(L.evaluate_row(t2, 0.0, np.array([1., 1.])), [7., 14.])
    f.w = torch.tensor([1., 2., 3.], dtype=torch.float64)[:2]
    assert L.trace(f, y0).key() == t1.key()
    g = lambda t, y: 2.0 * y * y                             # noqa: E731
    assert L.trace(g, y0).key() != t1.key()


def test_python_state_of_the_callable():
    class Counting(object):
        def __init__(self):
            self.nfe = 0

        def __call__(self, t, y):
            self.nfe += 1
            return -y
    c = Counting()
    low_err = None
    y0 = torch.zeros(3, dtype=torch.float64)
    before = L.fingerprint(c)
    L.trace(c, y0)
    assert L._restore_nfe(c, before, L.fingerprint(c)) and c.nfe == 0      # an integer `nfe` is put back: the probe is not an evaluation
    log = [0]

    def impure(t, y):
        log[0] += 1
        return -y
    before = L.fingerprint(impure)
    L.trace(impure, y0)
    assert not L._restore_nfe(impure, before, L.fingerprint(impure))
    assert low_err is None


def test_programs_bind_fresh_constants_into_persistent_buffers():
    """Program.bind on the CPU device: the catalogue routes get [in, out] copies that are REFRESHED in place (same storage every call)."""
    A = torch.tensor(np.random.RandomState(0).randn(6, 6))
    f = lambda t, y: (A @ y[..., None])[..., 0]              # noqa: E731
    y0 = torch.zeros(9, 6, dtype=torch.float64)
    tr = L.trace(f, y0)
    prog = L.program_for(tr)
    assert prog.kind == 'linear'
    r1 = prog.bind(tr, 'cpu')
    np.testing.assert_allclose(r1.W.numpy(), A.t().numpy())
    ptr = r1.W.data_ptr()
    A.mul_(3)
    tr2 = L.trace(f, y0)
    r2 = L.program_for(tr2).bind(tr2, 'cpu')
    assert r2 is r1 and r2.W.data_ptr() == ptr
    np.testing.assert_allclose(r2.W.numpy(), A.t().numpy())
    net = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.ReLU(), torch.nn.Linear(9, 5)).double()
    tr = L.trace(lambda t, y: net(y), torch.zeros(4, 5, dtype=torch.float64))
    prog = L.program_for(tr)
    assert prog.kind == 'mlp'
    m = prog.bind(tr, 'cpu')
    assert m.activation == 'relu' and m.hidden == 9 and torch.equal(m.Ws[1], torch.eye(9, dtype=torch.float64))
    y = torch.randn(4, 5, dtype=torch.float64)
    np.testing.assert_allclose(m.forward(0.0, y).numpy(), net(y).detach().numpy(), rtol=1e-13, atol=1e-13)


def test_generated_sources_are_stable_text():
    """The plugin cache is keyed by source text: tracing the same callable twice must give the same bytes."""
    for name in ('nb_lorenz', 'demo_net_f64', 'detest_C5', 'ring_100', 'swish_48'):
        f, y0, _ = LC.CASES[name]('cpu')
        a = L.sources_for(f, y0)
        f, y0, _ = LC.CASES[name]('cpu')
        assert a == L.sources_for(f, y0) and len(a) == 1
        assert 'MI_ODE_DEFINE_' in a[0]


def test_compiled_callables_trace_once_and_still_read_tensors_fresh():
    class F(object):
        def __init__(self):
            self.a = 2.0
            self.w = torch.tensor([1., 2.], dtype=torch.float64)
            self.calls = []

        def __call__(self, t, y):
            return self.a * y * self.w
    f = F()
    y0 = torch.zeros(4, 2, dtype=torch.float64)
    c = L.compile(f, y0)
    n_nodes = len(c._traces[next(iter(c._traces))].nodes)
    low = L.lower(c, y0)
    assert low.kind == 'rowlocal' and low.rhs.params == [2.0]
    f.w.mul_(3.0)                                           # tensors: by reference
    f.a = 5.0                                               # Python numbers: frozen at compile time (the documented promise)
    low = L.lower(c, y0)
    assert low.rhs.params == [2.0] and torch.equal(low.rhs.pool[:2], torch.tensor([3., 6.], dtype=torch.float64))
    assert len(c._traces) == 1 and len(c._traces[next(iter(c._traces))].nodes) == n_nodes
    L.lower(c, torch.zeros(4, 2, dtype=torch.float32))      # another dtype: traced afresh (and then sees a = 5)
    assert len(c._traces) == 2
    np.testing.assert_allclose(c(0.0, torch.ones(1, 2, dtype=torch.float64)).numpy(), [[15., 30.]])


def test_generated_cooperative_code_is_refused_where_rocblas_is_faster():
    """A network with an elementwise pre-op is not the catalogue's MLP shape: small batches get generated cooperative code, a batch whose
    evaluation exceeds ~5e7 multiply-adds stays on the callable engine (and says why)."""
    net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Tanh(), torch.nn.Linear(128, 128), torch.nn.Tanh(), torch.nn.Linear(128, 64)).double()
    f = lambda t, y: net(y ** 3)                            # noqa: E731
    small = L.lower(f, torch.zeros(100, 64, dtype=torch.float64))
    assert small.kind == 'coop' and small.program.macs == 64 * 128 + 128 * 128 + 128 * 64
    with pytest.raises(L.TraceError, match='rocBLAS'):
        L.lower(f, torch.zeros(32768, 64, dtype=torch.float64))


def test_trace_cache_is_reused_only_while_nothing_the_callable_names_has_changed():
    """lower() re-traces a callable only when something it can NAME changed (numbers by value, tensors by identity); a constant the walk
    cannot explain (a float computed inside the callable) makes the callable uncacheable - it is then simply traced on every call."""
    class F(object):
        def __init__(self):
            self.rate = 0.5
            self.w = torch.tensor([1., 2., 3.], dtype=torch.float64)
            self.net = torch.nn.Linear(3, 3).double()

        def __call__(self, t, y):
            return self.rate * y * self.w + self.net(y)
    f = F()
    y0 = torch.zeros(4, 3, dtype=torch.float64)
    L._TRACES.clear()
    st = L.trace_cache_stats

    def delta(fn):
        before = dict(st)
        out = fn()
        return out, {k: st[k] - before[k] for k in st}
    low, d = delta(lambda: L.lower(f, y0))
    assert d == {'hits': 0, 'misses': 1, 'uncacheable': 0} and low.rhs.params == [0.5]
    low, d = delta(lambda: L.lower(f, y0))
    assert d['hits'] == 1
    f.rate = 0.75                                           # a number changed: by VALUE
    low, d = delta(lambda: L.lower(f, y0))
    assert d['misses'] == 1 and low.rhs.params == [0.75]
    f.w.mul_(2.0)                                           # a tensor updated in place: no re-trace needed - tensors are re-read on every call
    with torch.no_grad():
        f.net.weight.add_(1.0)
    low, d = delta(lambda: L.lower(f, y0))
    assert d['hits'] == 1
    np.testing.assert_allclose(L.evaluate_row(low.trace, 0.0, np.ones(3)), f(0.0, torch.ones(1, 3, dtype=torch.float64)).detach().numpy()[0], rtol=1e-14)
    f.w = torch.tensor([5., 5., 5.], dtype=torch.float64)   # a tensor REBOUND: identity changed
    low, d = delta(lambda: L.lower(f, y0))
    assert d['misses'] == 1 and float(low.rhs.pool[low.program.layout.tensor_off[0]]) == 5.0
    f.net = torch.nn.Linear(3, 3).double()                  # a module replaced
    _, d = delta(lambda: L.lower(f, y0))
    assert d['misses'] == 1
    _, d = delta(lambda: L.lower(f, torch.zeros(9, 3, dtype=torch.float64)))       # another batch shape: its own entry
    assert d['misses'] == 1
    # a lambda re-created on every call: the same code over the same cells
    a = 2.0
    mk = lambda: (lambda t, y: a * y)                       # noqa: E731
    L.lower(mk(), y0)
    _, d = delta(lambda: L.lower(mk(), y0))
    assert d['hits'] == 1
    # a float computed inside the callable is not something the walk can see: never cached
    g = lambda t, y: (-f.rate) * y                          # noqa: E731
    L.lower(g, y0)
    _, d = delta(lambda: L.lower(g, y0))
    assert d == {'hits': 0, 'misses': 0, 'uncacheable': 1}
    f.rate = 0.1
    assert L.lower(g, y0).rhs.params == [-0.1]
