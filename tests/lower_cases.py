"""Callables the tracer (tfdiffeq_amd/lower.py) is tested with - shared by the CPU tests (trace -> numpy evaluation of the graph, generated
statements compiled by g++), the GPU tests (one launch, the oracle's attempt counts) and `__graft_entry__.build()` (which compiles the
generated kernels here so that the GPU box does not run hipcc).  Every entry is `name -> factory(device) -> (func, y0, expected kind)`."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, 'examples'), os.path.join(ROOT, 'tests')):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


class ConstantODE(object):                       # the reference's tests/problems.py:13-21 (a, b: tf.Variable scalars there)
    def __init__(self, device, as_tensor=False):
        self.a = torch.tensor(0.2, dtype=torch.float64, device=device) if as_tensor else 0.2
        self.b = torch.tensor(3.0, dtype=torch.float64, device=device) if as_tensor else 3.0

    def __call__(self, t, y):
        return self.a + (y - (self.a * t + self.b)) ** 5


class SineODE(object):                           # tests/problems.py:28-34
    def __call__(self, t, y):
        return 2 * y / t + t ** 4 * torch.sin(2 * t) - t ** 2 + 4 * t ** 3


class LinearODE(object):                         # tests/problems.py:43-56, written as the reference writes it (reshape, A @ y, reshape)
    def __init__(self, device, dim=10, seed=0):
        self.dim = dim
        U = np.random.RandomState(seed).randn(dim, dim) * 0.1
        A = 2 * U - (U + U.transpose(0, 1))
        self.A = torch.tensor(A, dtype=torch.float64, device=device)

    def __call__(self, t, y):
        y = torch.reshape(y, [self.dim, 1])
        out = torch.matmul(self.A, y)
        return torch.reshape(out, [-1])


class NFEDiffEq(object):                         # tests/DETEST/run.py:14-23
    def __init__(self, diffeq):
        self.diffeq = diffeq
        self.nfe = 0

    def __call__(self, t, y):
        self.nfe += 1
        return self.diffeq(t, y)


def _detest(name):
    def make(device):
        from oracle import detest_problems as DP
        like = torch.zeros(1, device=device, dtype=torch.float64)
        f, y0 = DP.problem(name, torch, like=like)
        return f, y0, ('linear' if name in ('C1', 'C2', 'C3', 'C4') else 'rowlocal')
    return make


def _notebook(name):
    def make(device):
        import reference_systems as RS
        s = RS.systems(device)[name]
        return s['func'], s['y0'], 'rowlocal'
    return make


def _demo_net(dtype):
    def make(device):
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(2, 50), torch.nn.Tanh(), torch.nn.Linear(50, 2)).to(device=device, dtype=dtype)
        for m in net.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.weight, mean=0, std=0.1)
                torch.nn.init.constant_(m.bias, val=0)

        class ODEFunc(torch.nn.Module):              # this repo's examples/ode_demo.py:27-33 = the reference's examples/ode_demo.py:115-129
            def __init__(self):
                super().__init__()
                self.net = net

            def forward(self, t, y):
                return self.net(y ** 3)
        g = torch.Generator().manual_seed(5)
        return ODEFunc(), (torch.rand(20, 1, 2, generator=g, dtype=torch.float64) * 2 - 1).to(device=device, dtype=dtype), 'rowlocal'
    return make


def _mlp(d, h, dtype, act, batch, layers=3):
    def make(device):
        torch.manual_seed(d * 1000 + h)
        acts = {'tanh': torch.nn.Tanh, 'relu': torch.nn.ReLU, 'softplus': torch.nn.Softplus}
        mods = [torch.nn.Linear(d, h), acts[act]()]
        if layers == 3:
            mods += [torch.nn.Linear(h, h), acts[act]()]
        mods += [torch.nn.Linear(h, d)]
        net = torch.nn.Sequential(*mods).to(device=device, dtype=dtype)
        g = torch.Generator().manual_seed(7)
        y0 = torch.randn(batch, d, generator=g, dtype=torch.float64).to(device=device, dtype=dtype)
        return (lambda t, y: net(y)), y0, 'mlp'
    return make


def _linear(d, batch, bias, form):
    def make(device):
        rng = np.random.RandomState(d)
        S = rng.randn(d, d)
        A = torch.tensor(-0.5 * np.eye(d) + 0.5 * (S - S.T) / np.sqrt(d), dtype=torch.float64, device=device)
        b = torch.tensor(0.1 * rng.randn(d), dtype=torch.float64, device=device) if bias else None
        y0 = torch.tensor(rng.randn(batch, d), dtype=torch.float64, device=device)
        if form == 'yW':
            f = (lambda t, y: y @ A) if b is None else (lambda t, y: torch.matmul(y, A) + b)
        elif form == 'Ay':                               # A y for every trajectory, as (A @ y[..., None])[..., 0]
            f = (lambda t, y: (A @ y[..., None])[..., 0]) if b is None else (lambda t, y: (A @ y[..., None])[..., 0] + b)
        else:
            lin = torch.nn.Linear(d, d, bias=bias).to(device=device, dtype=torch.float64)
            f = lambda t, y: lin(y)                      # noqa: E731
        return f, y0, 'linear'
    return make


def _ring(n, batch):
    def make(device):
        D, c = 0.8, 0.05
        g = torch.Generator().manual_seed(n)
        y0 = torch.randn(batch, n, generator=g, dtype=torch.float64).to(device)
        return (lambda t, y: D * (torch.roll(y, -1, -1) - 2 * y + torch.roll(y, 1, -1)) - c * y * y * y), y0, 'coop'
    return make


def _swish(d, batch):
    def make(device):
        g = torch.Generator().manual_seed(d)
        W = (torch.randn(d, d, generator=g, dtype=torch.float64) / math.sqrt(d)).to(device)
        b = (0.1 * torch.randn(d, generator=g, dtype=torch.float64)).to(device)
        y0 = torch.randn(batch, d, generator=g, dtype=torch.float64).to(device)

        def f(t, y):
            z = y @ W + b
            return z * torch.sigmoid(z) - 0.3 * y + 0.2 * torch.sin(t)
        return f, y0, 'coop'
    return make


def _tdep(device):
    y0 = torch.tensor(np.random.RandomState(1).randn(64, 3), dtype=torch.float64, device=device)
    return (lambda t, y: torch.sin(y) * t - 0.5 * y + torch.cos(t)), y0, 'rowlocal'


def _lorenz_batched(device):
    s, b, r = 10., 8. / 3., 28.
    y0 = torch.tensor(np.array([1., 1., 1.]) + 1e-3 * np.random.RandomState(2).randn(4096, 3), dtype=torch.float64, device=device)

    def f(t, y):
        x0, x1, x2 = y[..., 0], y[..., 1], y[..., 2]
        return torch.stack([s * (x1 - x0), x0 * (r - x2) - x1, x0 * x1 - b * x2], dim=-1)
    return f, y0, 'rowlocal'


def _spiral(device):
    import reference_systems as RS
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64, device=device)
    return RS.SpiralLambda(A), torch.tensor([[2., 0.]], dtype=torch.float64, device=device), 'rowlocal'


def _spiral_batch(device):
    import reference_systems as RS
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64, device=device)
    y0 = torch.tensor(np.random.RandomState(4).uniform(-2, 2, (512, 2)), dtype=torch.float64, device=device)
    return RS.SpiralLambda(A), y0, 'rowlocal'


def _lorenz_indexed(device):
    import reference_systems as RS
    return RS.LorenzIndexed(), torch.tensor([1., 1., 1.], dtype=torch.float64, device=device), 'rowlocal'


CASES = {
    'ref_constant': lambda dev: (ConstantODE(dev), torch.tensor(3.2, dtype=torch.float64, device=dev), 'rowlocal'),
    'ref_constant_tensor_params': lambda dev: (ConstantODE(dev, True), torch.tensor(3.2, dtype=torch.float64, device=dev), 'rowlocal'),
    'ref_sine': lambda dev: (SineODE(), torch.tensor(2.5, dtype=torch.float64, device=dev), 'rowlocal'),
    'ref_linear': lambda dev: (LinearODE(dev), torch.ones(10, dtype=torch.float64, device=dev), 'linear'),
    'demo_spiral': _spiral,
    'demo_spiral_batch': _spiral_batch,
    'demo_net_f64': _demo_net(torch.float64),
    'demo_net_f32': _demo_net(torch.float32),
    'lorenz_indexed': _lorenz_indexed,
    'lorenz_batched': _lorenz_batched,
    'tdep': _tdep,
    'mlp_64_128_tanh_f32': _mlp(64, 128, torch.float32, 'tanh', 256),
    'mlp_16_32_softplus_f64': _mlp(16, 32, torch.float64, 'softplus', 64),
    'mlp_8_24_relu2_f64': _mlp(8, 24, torch.float64, 'relu', 64, layers=2),
    'linear_128_yW': _linear(128, 512, False, 'yW'),
    'linear_24_yW_bias': _linear(24, 100, True, 'yW'),
    'linear_16_Ay': _linear(16, 33, False, 'Ay'),
    'linear_32_module': _linear(32, 64, True, 'module'),
    'ring_100': _ring(100, 48),
    'swish_48': _swish(48, 40),
}
for _n in [c + i for c in 'ABCDE' for i in '12345']:
    CASES['detest_' + _n] = _detest(_n)
for _n in ('second_order', 'oscilation', 'jagged_oscilation', 'nonlinear_damping', 'predator_prey', 'limited_predator_prey', 'periodic_sinusodial',
           'linear2d_1', 'linear2d_3', 'parabolic', 'nonlinear_system1', 'nonlinear_predator_prey', 'spiral_sink', 'jacobian_spiral_sink',
           'jacobian_predator_prey', 'spiral_cycle', 'force_pendulum', 'duffing', 'lorenz', 'rossler'):
    CASES['nb_' + _n] = _notebook(_n)


def literal_callable(meta, device, dtype=torch.float64):
    """A golden fixture's right-hand side as the Python callable the reference's tests / examples write (never a DeviceRHS)."""
    import reference_systems as RS
    rhs, p = meta['rhs'], meta['rhs_params']
    if rhs == 'sine':
        return SineODE()
    if rhs == 'constant':
        return ConstantODE(device)
    if rhs == 'cubic_linear':
        return RS.SpiralLambda(torch.tensor(p['W'], dtype=dtype, device=device))
    if rhs == 'lorenz':
        s, b, r = p['sigma'], p['beta'], p['rho']
        return lambda t, y: torch.stack([s * (y[..., 1] - y[..., 0]), y[..., 0] * (r - y[..., 2]) - y[..., 1],
                                         y[..., 0] * y[..., 1] - b * y[..., 2]], dim=-1)
    if rhs == 'lotka_volterra':
        a, b, c, d = p['a'], p['b'], p['c'], p['d']
        return lambda t, y: torch.stack([a * y[..., 0] - b * y[..., 0] * y[..., 1], -c * y[..., 1] + d * y[..., 0] * y[..., 1]], dim=-1)
    if rhs == 'linear':
        W = torch.tensor(p['W'], dtype=dtype, device=device)
        return lambda t, y: torch.matmul(y, W)
    raise KeyError(rhs)


def fixture_names():
    """The reference-generated whole-run fixtures the lowered callables are held to (tensor states; tsit5 needs `refcompat`, the MLP has
    its own tests)."""
    from golden_util import load, run_cases
    out = []
    for n in run_cases():
        _, m = load(n)
        if m['tuple_state'] or m['max_attempts'] is not None or m['rhs'] == 'mlp_tanh' or m['method'] == 'tsit5':
            continue
        if (m['options'] or {}).get('eps') or 'noint' in n:
            continue
        out.append(n)
    return out


class Decay(object):
    def __init__(self, device):
        self.rate = 0.5
        self.w = torch.tensor([1., 2., 3.], dtype=torch.float64, device=device)

    def __call__(self, t, y):
        return -self.rate * y * self.w


def _state_2x3x5(device):
    g = torch.Generator().manual_seed(0)
    M = (torch.randn(5, 5, generator=g, dtype=torch.float64) * 0.3).to(device)
    y3 = torch.randn(2, 3, 5, generator=g, dtype=torch.float64).to(device)
    return (lambda t, y: torch.stack([y[1] @ M, -y[0] @ M.t()]) * torch.cos(t)), y3, 'rowlocal'      # (indexes the first axis: no batch axes)


CASES['decay'] = lambda dev: (Decay(dev), torch.ones(7, 3, dtype=torch.float64, device=dev), 'rowlocal')
CASES['state_2x3x5'] = _state_2x3x5
CASES['minus_y_5x3'] = lambda dev: ((lambda t, y: -y), torch.ones(5, 3, dtype=torch.float64, device=dev), 'rowlocal')


def generated_sources(device='cpu'):
    """Every generated kernel source the cases above compile (CPU only: tracing needs no GPU)."""
    from tfdiffeq_amd import lower
    out = []
    for name, make in CASES.items():
        f, y0, _kind = make(device)
        out.extend(lower.sources_for(f, y0))
    g = lambda t, y: (torch.cos(t) * y[0] - y[0] ** 3, torch.cos(t) * y[1] - y[1] ** 3)     # noqa: E731  (tests/test_gpu_lower.py: tuple states)
    out.extend(lower.sources_for(g, (torch.zeros(40, 2, dtype=torch.float64, device=device), torch.zeros(7, 2, dtype=torch.float64, device=device))))
    c = ConstantODE(device)
    out.extend(lower.sources_for(lambda t, y: (c(t, y[0]), c(t, y[1])), (torch.tensor(3.2, dtype=torch.float64, device=device),) * 2))
    from golden_util import load
    for name in fixture_names():
        d, meta = load(name)
        y0 = torch.tensor(d['y0'], device=device)
        out.extend(lower.sources_for(literal_callable(meta, device, y0.dtype), y0, method=meta['method']))
    return out
