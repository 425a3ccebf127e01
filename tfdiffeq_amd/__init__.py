"""tfdiffeq_amd - MI355X-native explicit Runge-Kutta `odeint` (drop-in for the tfdiffeq hot path).

    from tfdiffeq_amd import odeint
    y = odeint(func, y0, t, method='dopri5')      # y0: torch tensor on the GPU

See DESIGN.md for the path, its boundary and the kernels; INTEGRATION.md for the C ABI.
"""
from .odeint import odeint, SOLVERS
from .adjoint import odeint_adjoint
from .misc import move_to_device
from . import rhs
from .solvers import clear_engine_cache

__all__ = ['odeint', 'odeint_adjoint', 'SOLVERS', 'move_to_device', 'rhs', 'clear_engine_cache']

__version__ = '0.1.0'
