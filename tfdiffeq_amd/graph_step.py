"""hipGraph capture of one adaptive Runge-Kutta attempt for an arbitrary Python right-hand side.

The generic path (any callable f over torch ops, plane kernels for the state arithmetic) issues ~60 small launches per
attempt and is bound by the host's launch rate.  Nothing in an attempt depends on host values except t0 and dt, and
those can live in device memory (`mi_ode_lincomb_dev` reads dt when it runs; the stage times are 0-d tensors computed
on the device).  So the attempt is recorded ONCE - the S evaluations of f, the stage combinations, y1, the error
estimate and its norms - and replayed per attempt with new (t0, dt); the host only reads back the 4-double norm record
and runs the controller.  Opt-in (`options={'graph': True}`): f must be capture-safe (static shapes, no host
synchronisation, no data-dependent Python control flow), which holds for the usual tensor-expression right-hand sides.
"""
import torch

from .misc import _error_norms
from .rk_common import _runge_kutta_step


class GraphedAttempt(object):
    def __init__(self, func, y0, f0, tableau):
        self.func = func
        self.tableau = tableau
        dev = y0[0].device
        self.y0 = tuple(torch.empty_like(y).copy_(y) for y in y0)            # static inputs of the graph
        self.f0 = tuple(torch.empty_like(f).copy_(f) for f in f0)
        self.td = torch.zeros(2, dtype=torch.float64, device=dev)            # [t0, dt]
        self.td_host = torch.zeros(2, dtype=torch.float64).pin_memory()
        self._pending = None                                                 # accepted (y1, f1) not yet copied into the inputs
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                                        # warm-up off the capture (torch.cuda.graphs recipe)
            self.td.fill_(0.0)
            for _ in range(2):
                self._body()
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()

    def _body(self):
        y1, f1, err, k = _runge_kutta_step(self.func, self.y0, self.f0, self.td[0], self.td[1], self.tableau)
        recs = torch.stack([_error_norms(e, a, b) for e, a, b in zip(err, self.y0, y1)])
        return y1, f1, err, k, recs

    def matches(self, y0):
        return len(y0) == len(self.y0) and all(a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
                                               for a, b in zip(y0, self.y0))

    def run(self, y0, f0, t0, dt):
        """One attempt from (y0, f0) at t0 with step dt.  Returns (y0_static, y1, f1, err, k, recs): static tensors that
        the NEXT run() overwrites (y1.. by the replay, y0_static by the deferred state copy)."""
        if self._pending is not None:
            y1p, f1p = self._pending
            if all(a is b for a, b in zip(y0, y1p)):                         # the caller advanced to the accepted state
                for dst, src in zip(self.y0, y1p):
                    dst.copy_(src)
                for dst, src in zip(self.f0, f1p):
                    dst.copy_(src)
            self._pending = None
        elif not all(a is b for a, b in zip(y0, self.y0)):                   # first call / foreign state
            for dst, src in zip(self.y0, y0):
                dst.copy_(src)
            for dst, src in zip(self.f0, f0):
                dst.copy_(src)
        self.td_host[0] = float(t0)
        self.td_host[1] = float(dt)
        self.td.copy_(self.td_host, non_blocking=True)
        self.graph.replay()
        y1, f1, err, k, recs = self.out
        return self.y0, y1, f1, err, k, recs

    def accepted(self, y1, f1):
        """The caller keeps (y1, f1) as its state: they move into the graph's inputs at the start of the next run()
        (not now: the dense output of this step still needs the old y0)."""
        self._pending = (y1, f1)


class GraphedFixedStep(object):
    """One step of a fixed-grid solver (solvers.py:94-97: dy = step_func(...); y1 = y0 + dy) as a hipGraph that updates
    the state in place.  The integration is then `for every grid interval: td <- (t0, dt) (device copy); replay; copy
    the state into solution[i+1]` - no host synchronisation anywhere in the loop."""

    def __init__(self, solver, y0):
        from .misc import _lincomb
        self.solver = solver
        dev = y0[0].device
        self.y = tuple(torch.empty_like(y).copy_(y) for y in y0)
        self.td = torch.zeros(2, dtype=torch.float64, device=dev)

        def body():
            dy = solver.step_func(solver.func, self.td[0], self.td[1], self.y)
            y1 = tuple(_lincomb(y_, [1.0], [dy_], 1.0) for y_, dy_ in zip(self.y, dy))      # solvers.py:95
            for y_, y1_ in zip(self.y, y1):
                y_.copy_(y1_)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()
        for dst, src in zip(self.y, y0):                     # the warm-up advanced the state
            dst.copy_(src)

    def integrate_pairs(self, t0s, dts, outs):
        """t0s / dts: host float64 arrays (left grid points, interval lengths); outs: per component [T, ...] result tensors
        whose row 0 already holds y0."""
        import numpy as np
        pairs_dev = torch.from_numpy(np.ascontiguousarray(np.stack([t0s, dts], axis=1))).to(self.td.device)
        for i in range(pairs_dev.shape[0]):
            self.td.copy_(pairs_dev[i])
            self.graph.replay()
            for out, y_ in zip(outs, self.y):
                out[i + 1].copy_(y_)
