"""The systems of the reference's examples as the torch callables a user of this package would write - the same expressions the
reference's notebook / scripts write with tf ops (examples/ode_usage.ipynb, examples/lorenz_attractor.py:20-50,
examples/ode_demo.py:32-35; `tf.unstack` -> `torch.unbind`, `tf.stack` -> `torch.stack`), with the initial states, output grids and -
where the notebook kept a `%%time` output - the wall time it records (Colab CPU).

Every entry is a PLAIN PYTHON CALLABLE: `odeint` traces it (tfdiffeq_amd/lower.py) and runs the whole call in one launch; nothing here
names a device right-hand side.  Used by tests/test_gpu_lower.py, bench.py --config published and `__graft_entry__.build()` (which
precompiles the generated kernels so that the GPU box does not run hipcc).
"""
import math

import numpy as np
import torch


class SecondOrder(object):                       # ode_usage.ipynb cell 15: y'' - 5 y' + 6 y = 0 as a 2-vector
    def __call__(self, t, y):
        u, v = y[0], y[1]
        du_dt = v
        dv_dt = 5 * v - 6 * u
        return torch.stack([du_dt, dv_dt])


class OscilationCurve(object):                   # cell 23
    def __call__(self, t, y):
        return torch.sin(t * t) * y


class JaggedOscilationCurve(object):             # cell 28
    def __call__(self, t, y):
        return torch.sign(torch.sin(t * t)) * y


class NonLinearDamping(object):                  # cell 33
    def __call__(self, t, y):
        return 3. * torch.cos(t) - torch.pow(y, 3)


class PredatorPrey(object):                      # cells 39-42 / README.md:67-82
    def __init__(self, a, b, c, d):
        self.a, self.b, self.c, self.d = a, b, c, d

    def __call__(self, t, y):
        r, f = torch.unbind(y)
        dR_dT = self.a * r - self.b * r * f
        dF_dT = -self.c * f + self.d * r * f
        return torch.stack([dR_dT, dF_dT])


class LimitedPredatorPrey(object):               # cell 46
    def __init__(self, d):
        self.d = d

    def __call__(self, t, y):
        r, f = torch.unbind(y)
        dR_dT = r * (1. - r) - r * f
        dF_dT = -f + self.d * r * f
        return torch.stack([dR_dT, dF_dT])


class PeriodicSinusodial(object):                # cell 53
    def __call__(self, t, v):
        x, y = torch.unbind(v)
        xy = torch.sqrt(x * x + y * y)
        dx_dt = x * (1 - xy) - y
        dy_dt = x + y * (1 - xy)
        return torch.stack([dx_dt, dy_dt])


class LinearODE1(object):                        # cell 68: v' = v @ system, v [1, 2]
    def __init__(self, system):
        self.system = system

    def __call__(self, t, v):
        return torch.matmul(v, self.system)


class ParabolicSystem(object):                   # cell 101
    def __call__(self, t, v):
        x, y = torch.unbind(v)
        return torch.stack([y - x * x, 1 - y])


class NonLinearSystem1(object):                  # cell 107
    def __call__(self, t, v):
        x, y = torch.unbind(v)
        return torch.stack([x * (1 - x), x - y * y])


class NonLinearSystemPredatorPrey(object):       # cell 113
    def __init__(self, A):
        self.A = A

    def __call__(self, t, v):
        x, y = torch.unbind(v)
        dx_dt = x * (1 - x) + self.A * x * y
        dy_dt = y * (1 - y) + x * y
        return torch.stack([dx_dt, dy_dt])


class SpiralSink(object):                        # cell 122
    def __call__(self, t, v):
        x, y = torch.unbind(v)
        dx_dt = y + x * (x * x + y * y)
        dy_dt = -x + y * (x * x + y * y)
        return torch.stack([dx_dt, dy_dt])


class JacobbianSpiralSink(object):               # cell 128: v [1, 2]
    def __init__(self, J):
        self.J = J

    def __call__(self, t, v):
        x, y = v[0, 0], v[0, 1]
        dx_dt = y + x * (x * x + y * y)
        dy_dt = -x + y * (x * x + y * y)
        dv = torch.stack([dx_dt, dy_dt])
        dv = torch.reshape(dv, [1, -1])
        return torch.matmul(dv, self.J)


class JacobbianNonLinearSystemPredatorPrey(object):   # cell 135
    def __init__(self, A):
        self.A = A

    def __call__(self, t, v):
        x, y = torch.unbind(v)
        dx_dt = x * (1 - x) - x * y
        dy_dt = -y + self.A * x * y
        return torch.stack([dx_dt, dy_dt])


class SpiralCycle(object):                       # cell 143
    def __call__(self, t, v):
        x, y = torch.unbind(v)
        sq = x * x + y * y
        return torch.stack([-y + x * (1 - sq), x + y * (1 - sq)])


class ForcePendulum(object):                     # cell 150
    def __init__(self, b, g, F, omega):
        self.b, self.g, self.F, self.omega = b, g, F, omega

    def __call__(self, t, y):
        theta, v = torch.unbind(y)
        dtheta_dt = v
        dv_dt = -self.b * v - self.g * torch.sin(theta) - self.F * torch.cos(self.omega * t) * torch.sin(theta)
        return torch.stack([dtheta_dt, dv_dt])


class DuffingOscilator(object):                  # cell 157
    def __call__(self, t, x):
        y, v = torch.unbind(x)
        return torch.stack([v, y - torch.pow(y, 3)])


class Lorenz(object):                            # cell 163 / examples/lorenz_attractor.py:20-37
    def __init__(self, sigma=10., beta=8 / 3., rho=28.):
        self.sigma, self.beta, self.rho = float(sigma), float(beta), float(rho)

    def __call__(self, t, y):
        x, y, z = torch.unbind(y)
        dx_dt = self.sigma * (y - x)
        dy_dt = x * (self.rho - z) - y
        dz_dt = x * y - self.beta * z
        return torch.stack([dx_dt, dy_dt, dz_dt])


class LorenzIndexed(Lorenz):                     # examples/lorenz_attractor.py:28-37 writes it with y[0], y[1], y[2]
    def __call__(self, t, y):
        dx_dt = self.sigma * (y[1] - y[0])
        dy_dt = y[0] * (self.rho - y[2]) - y[1]
        dz_dt = y[0] * y[1] - self.beta * y[2]
        return torch.stack([dx_dt, dy_dt, dz_dt])


class Rossler(object):                           # cell 169
    def __init__(self, c):
        self.c = float(c)

    def __call__(self, t, v):
        x, y, z = torch.unbind(v)
        dx_dt = -y - z
        dy_dt = x + y / 4
        dz_dt = 1 + z * (x - self.c)
        return torch.stack([dx_dt, dy_dt, dz_dt])


class SpiralLambda(object):                      # examples/ode_demo.py:32-35
    def __init__(self, true_A):
        self.true_A = true_A

    def __call__(self, t, y):
        return torch.matmul(y ** 3, self.true_A)


def linspace(a, b, n=1000):
    """tf.linspace(a, b, n) of float32 arguments, as the notebook calls it (float32 grid values, later cast to float64)."""
    return torch.tensor(np.linspace(np.float32(a), np.float32(b), n, dtype=np.float32).astype(np.float64))


def arange(a, b, step):
    return torch.tensor(np.arange(a, b, step, dtype=np.float64))


def systems(device='cuda', dtype=torch.float64):
    """name -> (callable, y0, t, wall seconds the reference's notebook records or None, source cell)."""
    def ten(v):
        return torch.tensor(v, dtype=dtype, device=device)
    out = {}

    def add(name, f, y0, t, published, cell):
        out[name] = {'func': f, 'y0': ten(y0), 't': t, 'published_s': published, 'source': cell}
    add('second_order', SecondOrder(), [1., 1.], linspace(0., 1.), 0.781, 'ode_usage.ipynb cells 15-17')
    add('oscilation', OscilationCurve(), 1., linspace(0., 8.), 1.55, 'cells 23-25')
    add('jagged_oscilation', JaggedOscilationCurve(), 1., linspace(0., 8.), 6.31, 'cells 28-30')
    add('nonlinear_damping', NonLinearDamping(), 0., linspace(0., 20.), 2.8, 'cells 33-35')
    add('predator_prey', PredatorPrey(1.5, 1, 3, 1), [1., 1.], linspace(0., 10.), 1.71, 'cells 39-42')
    add('limited_predator_prey', LimitedPredatorPrey(2), [1., 1.], linspace(0., 20.), 1.05, 'cells 46-49')
    add('periodic_sinusodial', PeriodicSinusodial(), [1., 1.], linspace(0., 10.), 1.14, 'cells 53-55')
    for k, (sysm, y0, t, pub, cell) in enumerate([
            ([[-1., 0.], [1., -2.]], [[1., 1.]], linspace(0., 10.), 0.981, 'cells 68-71'),
            ([[0., -1.], [1., 0.]], [[1., 1.]], linspace(-np.pi, np.pi), 0.844, 'cells 75-77'),
            ([[-0.1, -1.], [1., -0.1]], [[1., 1.]], linspace(0., 25.), 1.58, 'cells 81-83'),
            ([[0., 1.], [0., -1.]], [[0., -1.]], linspace(0., 5.), 0.807, 'cells 87-89'),
            ([[1., 0.], [1., 0.5]], [[0., 0.1]], linspace(0., 5.), 0.840, 'cells 93-95')]):
        add('linear2d_%d' % (k + 1), LinearODE1(ten(sysm)), y0, t, pub, cell)
    add('parabolic', ParabolicSystem(), [-0.2, 0.2], linspace(0., 25.), 1.15, 'cells 101-103')
    add('nonlinear_system1', NonLinearSystem1(), [0.1, 0.0], linspace(0., 10.), 0.880, 'cells 107-109')
    add('nonlinear_predator_prey', NonLinearSystemPredatorPrey(0.33), [2.0, 1.0], linspace(0., 20.), 1.07, 'cells 113-116')
    add('spiral_sink', SpiralSink(), [0.5, 0.5], linspace(0., 1.), 2.14, 'cells 122-124')
    add('jacobian_spiral_sink', JacobbianSpiralSink(ten([[0., 1.], [-1., 0.]])), [[1., 1.]], linspace(0., 2.), 5.35, 'cells 128-131')
    add('jacobian_predator_prey', JacobbianNonLinearSystemPredatorPrey(5), [5.0, 2.5], linspace(0., 1.), 0.920, 'cells 135-138')
    add('spiral_cycle', SpiralCycle(), [2.0, 2.0], linspace(0., 20.), 1.8, 'cells 143-145')
    add('force_pendulum', ForcePendulum(1., 9.8, 1., math.pi / 2), [math.pi, 0.], linspace(0., 25.), 2.0, 'cells 150-153')
    add('duffing', DuffingOscilator(), [-1., 1.], linspace(0., 25.), 2.29, 'cells 157-159')
    add('lorenz', Lorenz(10., 8. / 3., 28.), [1., 1., 1.], arange(0.0, 100.0, 0.01), 47.6, 'cells 163-166 / lorenz_attractor.py:40-50')
    add('rossler', Rossler(0.01), [1., 1., 1.], arange(0.0, 25.0, 0.01), 7.99, 'cells 169-172')
    return out


PUBLISHED = ('lorenz', 'predator_prey', 'rossler', 'second_order', 'linear2d_1', 'linear2d_2', 'linear2d_3', 'linear2d_4', 'linear2d_5')
