"""GPU tests (-m gpu) of the path every reference test and example takes: `odeint(func, y0, t)` with a PYTHON callable
(/root/reference/tests/odeint_tests.py:30-77, examples/ode_demo.py:39,169).  Round 4: the attempt loop's scalars live on the device
(libmi_ode family C: mi_ode_opq_finish / mi_ode_opq_commit), attempts run eagerly first and as one hipGraph replay each once enough
of them remain (tfdiffeq_amd/graph_step.DeviceControlledRK).  Checked here:
  * the three schedules (graph False / True / 'auto') produce IDENTICAL bits (they launch the same kernels);
  * the device controller against the host controller of rounds 1-3 (`graph='host'`: the reference's loop, one synchronisation per
    attempt): same attempt / accept counts, values to 1e-12 (the error sums are folded in a different order);
  * the numpy oracle and closed forms; several output times per step and steps spanning several output times; reversed time; tuple
    states with per-component ratios; float32; every adaptive tableau;
  * the reference's assertions (max_num_steps, non-finite state) surface as AssertionError;
  * a right-hand side that cannot be recorded (host synchronisation inside f) falls back to eager evaluation with the same result,
    one that uses autograd is never recorded, and Python-side evaluation counters are credited with the replayed evaluations.
"""
import warnings

import numpy as np
import pytest
import torch

from oracle import ode_numpy as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def lorenz(t, y):
    x, yy, z = y[..., 0], y[..., 1], y[..., 2]
    return torch.stack([10.0 * (yy - x), x * (28.0 - z) - yy, x * yy - (8.0 / 3.0) * z], dim=-1)


def lorenz_np(t, y):
    x, yy, z = y[..., 0], y[..., 1], y[..., 2]
    return np.stack([10.0 * (yy - x), x * (28.0 - z) - yy, x * yy - (8.0 / 3.0) * z], axis=-1)


def forced(t, y):
    return torch.stack([y[..., 1], 0.7 * torch.cos(2.0 * t) - y[..., 0], -0.1 * y[..., 2]], dim=-1)


def forced_np(t, y):
    return np.stack([y[..., 1], 0.7 * np.cos(2.0 * t) - y[..., 0], -0.1 * y[..., 2]], axis=-1)


def y0_lorenz(n, seed=5, dtype=torch.float64):
    rng = np.random.default_rng(seed)
    return torch.tensor(np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((n, 3)), dtype=dtype, device=dev())


@pytest.mark.parametrize('method', ['dopri5', 'tsit5', 'bosh3', 'dopri8', 'adaptive_heun'])
def test_schedules_agree_bit_for_bit_and_with_the_host_controller(method):
    from tfdiffeq_amd import odeint
    y0 = y0_lorenz(300)
    scale = 0.05 if method in ('bosh3', 'adaptive_heun') else 1.0
    for f, t in ((lorenz, np.linspace(0., 1.5, 6) * scale), (forced, np.array([0., 0.3, 0.31, 0.32, 2.5]) * (4 * scale)),
                 (lorenz, -np.linspace(0., 0.8, 3) * scale)):
        tt = torch.tensor(t)
        res, st = {}, {}
        for g in (False, True, 'auto', 'host'):
            res[g] = odeint(f, y0, tt, method=method, rtol=1e-6, atol=1e-9, options={'graph': g})
            st[g] = dict(odeint.last_stats)
        assert 'device-controlled' in st[False]['engine'] and 'hipGraph' in st[True]['engine'], st
        assert st['host']['engine'] == 'plane kernels'
        for g in (True, 'auto'):
            assert torch.equal(res[False], res[g]), (method, g)
            assert st[g]['n_attempts'] == st[False]['n_attempts'] and st[g]['n_accepted'] == st[False]['n_accepted']
        assert st[False]['n_attempts'] == st['host']['n_attempts'] and st[False]['n_accepted'] == st['host']['n_accepted'], (st[False], st['host'])
        err = (res[False] - res['host']).abs().max().item() / max(1.0, res['host'].abs().max().item())
        assert err < 1e-12, (method, err)
        assert torch.equal(res[False][0], y0)


def test_default_is_the_device_controlled_engine_and_it_records_long_solves():
    from tfdiffeq_amd import odeint
    y0 = y0_lorenz(64)
    out = odeint(lorenz, y0, torch.tensor([0., 3.0]), rtol=1e-7, atol=1e-9)             # ~150 attempts: recorded after the first few
    st = dict(odeint.last_stats)
    assert 'hipGraph replay' in st['engine'], st
    assert st['replays'] > 0 and st['n_polls'] < st['n_attempts'] // 4, st                # read back once per chunk, not per attempt
    ref, rs = O.odeint(lorenz_np, y0.cpu().numpy(), np.array([0., 3.0]), rtol=1e-7, atol=1e-9, method='dopri5', return_stats=True)
    assert st['n_attempts'] == rs.n_attempts and st['n_accepted'] == rs.n_accepted, (st, rs)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-6 * np.abs(ref).max()               # Lorenz amplifies rounding: e^{0.9 t}
    short = odeint(lorenz, y0, torch.tensor([0., 0.05]), rtol=1e-7, atol=1e-9)            # a handful of attempts: never recorded
    assert 'one Python evaluation per stage' in odeint.last_stats['engine'], odeint.last_stats
    assert short.shape == (2, 64, 3)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_against_the_oracle_with_outputs_inside_and_across_steps(dtype):
    from tfdiffeq_amd import odeint
    rng = np.random.default_rng(9)
    y0 = rng.standard_normal((200, 3))
    # clustered output times (several inside one step) followed by a long gap (many steps per output)
    t = np.array([0., 1e-4, 2e-4, 3e-4, 0.5, 0.5001, 4.0])
    tol = (1e-7, 1e-9) if dtype == torch.float64 else (1e-4, 1e-6)
    got = odeint(forced, torch.tensor(y0, dtype=dtype, device=dev()), torch.tensor(t), rtol=tol[0], atol=tol[1], method='dopri5',
                 options={'graph': True})
    st = dict(odeint.last_stats)
    ref, rs = O.odeint(forced_np, y0.astype(np.float64 if dtype == torch.float64 else np.float32), t, rtol=tol[0], atol=tol[1],
                       method='dopri5', return_stats=True)
    band = 1e-9 if dtype == torch.float64 else 2e-4
    assert np.abs(got.double().cpu().numpy() - ref).max() < band * (1 + np.abs(ref).max())
    if dtype == torch.float64:
        assert st['n_attempts'] == rs.n_attempts and st['n_accepted'] == rs.n_accepted, (st, rs)


def test_tuple_state_per_component_ratios_and_float32():
    from tfdiffeq_amd import odeint
    ya, yb = y0_lorenz(120, dtype=torch.float32), torch.linspace(0.5, 2.0, 14, device=dev()).reshape(7, 2)
    ft = lambda t_, ys: (lorenz(t_, ys[0]), -ys[1] * torch.cos(t_))  # noqa: E731
    t = torch.tensor(np.linspace(0., 1.0, 5))
    outs = {}
    for g in (False, True, 'host'):
        outs[g] = odeint(ft, (ya, yb), t, method='dopri5', rtol=[1e-5, 1e-4], atol=[1e-7, 1e-6], options={'graph': g})
        st = dict(odeint.last_stats)
        outs[g] = (outs[g], st['n_attempts'], st['n_accepted'])
    assert torch.equal(outs[False][0][0], outs[True][0][0]) and torch.equal(outs[False][0][1], outs[True][0][1])
    assert outs[False][1:] == outs['host'][1:], (outs[False][1:], outs['host'][1:])
    for a, b in zip(outs[False][0], outs['host'][0]):
        assert (a - b).abs().max().item() < 2e-5 * (1 + b.abs().max().item())
    exact = yb.double() * torch.exp(-torch.sin(t.to(dev())))[:, None, None]
    assert (outs[True][0][1].double() - exact).abs().max().item() < 1e-3


def test_the_reference_assertions_surface():
    from tfdiffeq_amd import odeint
    y0 = y0_lorenz(16)
    with pytest.raises(AssertionError, match='max_num_steps exceeded'):
        odeint(lorenz, y0, torch.tensor([0., 5.0]), method='dopri5', rtol=1e-9, atol=1e-11, options={'max_num_steps': 7})
    bad = y0.clone()
    bad[3, 1] = float('nan')
    with pytest.raises(AssertionError, match='non-finite'):      # dopri5.py:99-100 (with the automatic first step a NaN state trips
        odeint(lorenz, bad, torch.tensor([0., 1.0]), method='dopri5', options={'first_step': 0.01})   # the dt assertion of :98 first)
    with pytest.raises(AssertionError, match='underflow in dt'):
        odeint(lorenz, bad, torch.tensor([0., 1.0]), method='dopri5')


_SYNCING = r"""
import sys, warnings
import numpy as np, torch
sys.path.insert(0, %r)
from tfdiffeq_amd import odeint
def lorenz(t, y):
    x, yy, z = y[..., 0], y[..., 1], y[..., 2]
    return torch.stack([10.0 * (yy - x), x * (28.0 - z) - yy, x * yy - (8.0 / 3.0) * z], dim=-1)
def syncing(t_, y):                                      # a host synchronisation inside f: cannot be stream-captured
    if float(t_) < -1.0:
        return y
    return lorenz(t_, y)
rng = np.random.default_rng(5)
y0 = torch.tensor(np.array([1., 1., 1.]) + 1e-2 * rng.standard_normal((32, 3)), device='cuda:0')
t = torch.tensor([0., 2.0])
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    a = odeint(syncing, y0, t, method='dopri5', options={'graph': True})
st = dict(odeint.last_stats)
assert 'one Python evaluation per stage' in st['engine'], st
assert any('hipGraph' in str(x.message) for x in w), [str(x.message) for x in w]
b = odeint(lorenz, y0, t, method='dopri5', options={'graph': False})
assert torch.equal(a, b)
c = odeint(lorenz, y0, t, method='dopri5', options={'graph': True})      # and the device still records afterwards
assert torch.equal(a, c) and 'hipGraph' in odeint.last_stats['engine']
print('SYNCING_OK')
"""


def test_unrecordable_and_autograd_right_hand_sides():
    """(the failed recording runs in a process of its own: a capture that aborts must not take the test session with it)"""
    import os
    import subprocess
    import sys
    from tfdiffeq_amd import odeint
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, '-c', _SYNCING % root], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'SYNCING_OK' in res.stdout, (res.stdout[-2000:], res.stderr[-3000:])
    y0 = y0_lorenz(32)
    # autograd inside f (what the adjoint's augmented dynamics and Hamiltonian networks do): detected during the eager attempts
    w_ = torch.tensor(0.3, dtype=torch.float64, device=dev())

    def hamiltonian(t_, y):
        with torch.enable_grad():
            q = y.detach().requires_grad_(True)
            h = (0.5 * (q * q).sum() + w_ * torch.cos(q[..., 0]).sum())
            g, = torch.autograd.grad(h, q)
        return torch.stack([g[..., 1], -g[..., 0], -0.1 * g[..., 2]], dim=-1)
    with torch.no_grad():
        b = odeint(hamiltonian, y0, torch.tensor([0., 6.0]), rtol=1e-8, atol=1e-10)
    st = dict(odeint.last_stats)
    assert st['autograd_in_f'] and 'one Python evaluation per stage' in st['engine'], st
    assert torch.isfinite(b).all()


def test_python_evaluation_counters_are_credited():
    """tests/DETEST/run.py:18-21 counts evaluations in Python (`self.nfe += 1`); a replayed evaluation does not run Python."""
    from tfdiffeq_amd import odeint

    class Counted(object):
        def __init__(self):
            self.nfe = 0

        def __call__(self, t, y):
            self.nfe += 1
            return forced(t, y)
    y0 = y0_lorenz(8)
    t = torch.tensor([0., 20.0])
    counts = {}
    for g in ('host', False, True, 'auto'):
        f = Counted()
        odeint(f, y0, t, rtol=1e-8, atol=1e-10, method='dopri5', options={'graph': g})
        counts[g] = (f.nfe, dict(odeint.last_stats)['n_attempts'])
    assert counts['host'][0] == 2 + 6 * counts['host'][1]                 # dopri5.py:71-75 + six stages per attempt
    for g in (False, True, 'auto'):
        assert counts[g] == counts['host'], counts


def test_reference_unit_test_shapes_through_the_default_path():
    """The reference's own problems (tests/problems.py: constant, linear, sine) with Python callables, rel 1e-4 as
    tests/odeint_tests.py:27-31 asserts - now through the device-controlled engine."""
    from tfdiffeq_amd import odeint
    d = dev()
    t = torch.linspace(1., 8., 10, dtype=torch.float64)
    # SineODE (problems.py): dy/dt = 2 y / t + t^4 sin(2t) - t^2 + 4 t^3
    def sine(t_, y):
        return 2 * y / t_ + t_ ** 4 * torch.sin(2 * t_) - t_ ** 2 + 4 * t_ ** 3

    def sine_exact(tt):
        return (-0.5 * tt ** 4 * np.cos(2 * tt) + 0.5 * tt ** 3 * np.sin(2 * tt) + 0.25 * tt ** 2 * np.cos(2 * tt) - tt ** 3 + 2 * tt ** 4 +
                (np.pi - 0.25) * tt ** 2)
    tn = t.numpy()
    y0 = torch.tensor([sine_exact(tn[0])], dtype=torch.float64, device=d)
    for method in ('dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun'):
        got = odeint(sine, y0, t, method=method).cpu().numpy()[:, 0]
        assert 'device-controlled' in odeint.last_stats['engine']
        rel = np.abs(got - sine_exact(tn)).max() / np.abs(sine_exact(tn)).max()
        # (bosh3: the reference's tableau typos, SURVEY F5, cost it accuracy - 7e-4 here exactly as on the host-controlled path)
        assert rel < {'bosh3': 5e-3, 'adaptive_heun': 5e-3}.get(method, 1e-4), (method, rel)
        host = odeint(sine, y0, t, method=method, options={'graph': 'host'}).cpu().numpy()[:, 0]
        assert np.abs(got - host).max() <= 1e-9 * np.abs(host).max(), method


def test_recorded_attempt_reused_across_calls():
    """options={'graph': 'reuse'}: the first call records the attempt, later calls with the same callable / shapes / tolerances only
    reset the controller and replay - identical bits to the default schedule for new initial values, other time grids (more output
    times, reversed), and after an in-place parameter update; a different callable, shape or tolerance records its own."""
    from tfdiffeq_amd import graph_step, odeint
    graph_step.clear_recorded_attempts()

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.lin = torch.nn.Linear(3, 3).double()
            self.nfe = 0

        def forward(self, t, y):
            self.nfe += 1
            return torch.tanh(self.lin(y)) - 0.1 * y * torch.cos(t)
    net = Net().to(dev())
    opts = {'graph': 'reuse'}
    with torch.no_grad():
        cases = [(y0_lorenz(64, 5), torch.tensor([0., 2.0, 4.0], dtype=torch.float64)),
                 (y0_lorenz(64, 6), torch.tensor([0., 1.0, 1.5, 5.0, 5.1], dtype=torch.float64)),
                 (y0_lorenz(64, 7), torch.tensor([3.0, 1.0, 0.0], dtype=torch.float64))]
        for i, (y0, t) in enumerate(cases):
            net.nfe = 0
            got = odeint(net, y0.to(dev()), t, rtol=1e-7, atol=1e-9, method='dopri5', options=dict(opts))
            st = dict(odeint.last_stats)
            nfe_reuse = net.nfe
            if i == 0 or i == 2:                                   # (reversed time wraps f differently: a recording of its own)
                assert 'eager attempts first' in st['engine'], st
            else:
                assert 'recorded by an earlier call' in st['engine'] and st['replays'] >= st['n_attempts'], st
            net.nfe = 0
            ref = odeint(net, y0.to(dev()), t, rtol=1e-7, atol=1e-9, method='dopri5')
            rs = dict(odeint.last_stats)
            assert torch.equal(got, ref)
            assert (st['n_attempts'], st['n_accepted'], st['nfe']) == (rs['n_attempts'], rs['n_accepted'], rs['nfe'])
            assert nfe_reuse == net.nfe == 2 + 6 * st['n_attempts']   # replayed evaluations are credited to the module's counter (dopri5.py:71-75 + six per attempt)
        assert len(graph_step._RECORDED) == 2
        # an optimizer-style in-place update is seen by the replays (the graph reads the parameter's memory)
        y0, t = cases[0]
        net.lin.weight.mul_(0.5)
        got = odeint(net, y0.to(dev()), t, rtol=1e-7, atol=1e-9, method='dopri5', options=dict(opts))
        assert 'recorded by an earlier call' in odeint.last_stats['engine']
        ref = odeint(net, y0.to(dev()), t, rtol=1e-7, atol=1e-9, method='dopri5')
        assert torch.equal(got, ref)
        # other tolerances / shapes / callables do not hit the recording
        odeint(net, y0.to(dev()), t, rtol=1e-6, atol=1e-9, method='dopri5', options=dict(opts))
        assert 'eager attempts first' in odeint.last_stats['engine']
        odeint(net, y0_lorenz(32, 5).to(dev()), t, rtol=1e-7, atol=1e-9, method='dopri5', options=dict(opts))
        assert 'eager attempts first' in odeint.last_stats['engine']
        odeint(lorenz, y0.to(dev()), torch.tensor([0., 0.5], dtype=torch.float64), rtol=1e-7, atol=1e-9, method='dopri5', options=dict(opts))
        assert 'eager attempts first' in odeint.last_stats['engine']
        assert len(graph_step._RECORDED) <= graph_step._RECORDED_MAX
        # more output times than the native handle has held so far (1024): its output-time table moves, the attempt is recorded again -
        # and that recording serves the next call
        for n_t, label in ((1100, 'eager attempts first'), (1100, 'recorded by an earlier call'), (3, 'recorded by an earlier call')):
            tl = torch.linspace(0., 2., n_t, dtype=torch.float64)
            got = odeint(net, y0.to(dev()), tl, rtol=1e-7, atol=1e-9, method='dopri5', options=dict(opts))
            assert label in odeint.last_stats['engine'], (n_t, odeint.last_stats)
            assert torch.equal(got, odeint(net, y0.to(dev()), tl, rtol=1e-7, atol=1e-9, method='dopri5'))
    graph_step.clear_recorded_attempts()
    assert len(graph_step._RECORDED) == 0
