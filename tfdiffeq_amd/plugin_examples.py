"""Example right-hand-side plugins (rhs.CustomRowLocal).  `__graft_entry__.build()` precompiles them so that tests and
first uses on a GPU box hit the in-tree cache (tfdiffeq_amd/_plugins/) instead of running hipcc there."""
import torch

from . import rhs

LORENZ_BODY = """
k[0] = p[0] * (y[1] - y[0]);
k[1] = y[0] * (p[2] - y[2]) - y[1];
k[2] = y[0] * y[1] - p[1] * y[2];
"""


def lorenz(sigma=10.0, beta=8.0 / 3.0, rho=28.0):
    """examples/lorenz_attractor.py:28-37 as user device code: the arithmetic of the built-in rhs.Lorenz, literally."""
    return rhs.CustomRowLocal(3, LORENZ_BODY, params=[sigma, beta, rho],
                              torch_fn=lambda t, y: torch.stack([sigma * (y[..., 1] - y[..., 0]),
                                                                 y[..., 0] * (rho - y[..., 2]) - y[..., 1],
                                                                 y[..., 0] * y[..., 1] - beta * y[..., 2]], dim=-1))


def forced_oscillator(amp=0.7, w=2.0):
    """y'' + y = amp cos(w t): a time-dependent system with parameters."""
    return rhs.CustomRowLocal(2, "k[0] = y[1];\nk[1] = p[0] * cos(p[1] * t) - y[0];", params=[amp, w],
                              torch_fn=lambda t, y: torch.stack([y[..., 1], amp * torch.cos(w * t) - y[..., 0]], dim=-1))


def van_der_pol(mu=5.0):
    return rhs.CustomRowLocal(2, "k[0] = y[1];\nk[1] = p[0] * (1 - y[0] * y[0]) * y[1] - y[0];", params=[mu],
                              torch_fn=lambda t, y: torch.stack([y[..., 1], mu * (1 - y[..., 0] ** 2) * y[..., 1] - y[..., 0]], dim=-1))


def oscillator_ring(n=8, k0=1.0, coupling=0.5, damping=0.05):
    """A ring of n coupled damped oscillators, state (x_0 .. x_{n-1}, v_0 .. v_{n-1}) - dim 2 n: user device code beyond the dim-8 limit of
    rounds 1-4 (n = 8: dim 16 in float64 registers; n = 16: dim 32, float32)."""
    body = "\n".join(["for (int i = 0; i < %d; ++i) k[i] = y[%d + i];" % (n, n),
                      "for (int i = 0; i < %d; ++i) k[%d + i] = -p[0] * y[i] + p[1] * (y[(i + %d) %% %d] - (T)2 * y[i] + y[(i + 1) %% %d]) - p[2] * y[%d + i];"
                      % (n, n, n - 1, n, n, n)])

    def torch_fn(t, y):
        x, v = y[..., :n], y[..., n:]
        return torch.cat([v, -k0 * x + coupling * (torch.roll(x, 1, -1) - 2.0 * x + torch.roll(x, -1, -1)) - damping * v], dim=-1)
    return rhs.CustomRowLocal(2 * n, body, params=[k0, coupling, damping], torch_fn=torch_fn)


def reaction_diffusion_ring(n=100, diffusion=0.8, cubic=0.05):
    """rhs.CustomCoop (round 5): u_t = D (u_{i+1} - 2 u_i + u_{i-1}) - c u^3 on a ring of n cells - a stencil over a state far beyond what
    one thread keeps; a thread per cell, the trajectory's state shared through LDS."""
    return rhs.CustomCoop(n, "k = p[0] * (y[(i + 1) % DIM] - 2 * y[i] + y[(i + DIM - 1) % DIM]) - p[1] * y[i] * y[i] * y[i];", params=[diffusion, cubic],
                          torch_fn=lambda t, y: diffusion * (torch.roll(y, -1, -1) - 2 * y + torch.roll(y, 1, -1)) - cubic * y * y * y)


def swish_layer(W, b, decay=0.3, forcing=0.2):
    """rhs.CustomCoop: a user's own dense layer with their own pointwise function - f = swish(y W + b) - decay y + forcing sin(t);
    W [d, d] and b [d] travel as device arrays (w0, w1)."""
    d = int(W.shape[0])
    body = """
T z = w1[i];
for (int j = 0; j < DIM; ++j) z = fma(y[j], w0[j * DIM + i], z);
k = z / ((T)1 + exp(-z)) - p[0] * y[i] + p[1] * sin(t);
"""

    def torch_fn(t, y):
        z = y @ W.to(y) + b.to(y)
        return z * torch.sigmoid(z) - decay * y + forcing * torch.sin(torch.as_tensor(t, dtype=y.dtype, device=y.device))
    return rhs.CustomCoop(d, body, params=[decay, forcing], tensors=[W, b], torch_fn=torch_fn)


def prebuild(extra_sources=()):
    """Compile every example for both state dtypes (cache hits are free) and drop cache entries that belong to older
    kernel headers (the cache key covers the headers, so those can never be hit again).  extra_sources: more plugin sources to keep in
    the in-tree cache (the kernels tfdiffeq_amd.lower generates for the callables of the tests / examples / bench)."""
    import os
    from . import _plugin_build
    sources = [f.source(dt) for f in (lorenz(), forced_oscillator(), van_der_pol()) for dt in (torch.float64, torch.float32)]
    sources += [oscillator_ring(8).source(torch.float64), oscillator_ring(16).source(torch.float32),
                reaction_diffusion_ring(100).source(torch.float64), swish_layer(torch.zeros(48, 48), torch.zeros(48)).source(torch.float64)]
    sources += [s_ for s_ in dict.fromkeys(extra_sources) if s_ not in sources]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:      # (hipcc runs in subprocesses: ten plugins in the time of the slowest)
        out = list(pool.map(_plugin_build.build, sources))
    keep = set(os.path.basename(p)[:-3] for p in out)
    d = _plugin_build.plugin_dir()
    for name in os.listdir(d):
        stem = name.rsplit('.', 1)[0]
        if name.startswith('rhs_') and stem not in keep:
            try:
                os.remove(os.path.join(d, name))
            except OSError:
                pass
    return out
