"""Device right-hand-side catalogue.

A `DeviceRHS` is an ordinary `func(t, y)` callable (so it also works through the generic plane-kernel
path and can be compared against any other implementation), that additionally carries the descriptor
(`mi_ode_rhs`, include/mi_ode.h) which lets `odeint` evaluate it INSIDE the fused stage kernels.

Workload definitions follow the reference's examples:
  Linear          f = y @ W (+ b)              BASELINE config 4 (W = A^T for f = A y)
  CubicLinear     f = (y**3) @ W               examples/ode_demo.py:33-35 (spiral)
  LotkaVolterra   examples/ode_usage.ipynb cells 39-42 / README.md:67-82, batched on the last axis
  Lorenz          examples/lorenz_attractor.py:20-37, batched on the last axis
  MLP / MLPTanh   tfdiffeq/models/dense_odenet.py:41-92 (ODEFunc, time independent; tanh, relu or softplus)
"""
import os

import torch

from . import _native as N


class DeviceRHS(object):
    kind = 0
    dim = None

    def __init__(self):
        self.sign = 1.0
        self.nfe = 0          # evaluations through the Python __call__ path only (cf. dense_odenet.py:38,78)
        self._cache = {}

    # -- python path -------------------------------------------------------------------------
    def forward(self, t, y):
        raise NotImplementedError

    def __call__(self, t, y):
        self.nfe += 1
        if self.sign < 0:                       # misc.py:318-321: f <- -f(-t, y)
            return -self.forward(-t, y)
        return self.forward(t, y)

    def reversed(self):
        import copy
        r = copy.copy(self)
        r._cache = {}
        r.sign = -self.sign
        return r

    # -- device path -------------------------------------------------------------------------
    def _dev(self, x, dtype, device):
        """x on `device` in `dtype`, contiguous.  A tensor that already qualifies is used as is (in-place updates
        stay visible to the kernels); otherwise a converted copy is cached until x is modified (x._version)."""
        if x.device == torch.device(device) and x.dtype == dtype and x.is_contiguous():
            return x.detach()
        key = (id(x), dtype, str(device))            # id of the long-lived parameter object itself
        hit = self._cache.get(key)
        if hit is None or hit[0] != x._version or hit[2] is not x:
            hit = (x._version, x.detach().to(device=device, dtype=dtype).contiguous(), x)
            self._cache[key] = hit
        return hit[1]

    def _params_key(self):
        return ()

    def cache_key(self, dtype, device):
        """Identity of the fused engine this RHS needs: kind, sign, by-value scalars, device pointers."""
        r = N.Rhs()
        self.fill(r, dtype, device)       # (the engine built from this key keeps the tensors behind the pointers alive)
        return (self.kind, self.dim, float(self.sign), tuple(r.scalars), tuple(int(v or 0) for v in r.w),
                tuple(int(v or 0) for v in r.b), int(r.hidden))

    fixed_grid_fused = True
    row_local = False             # True: runs on the one-trajectory-per-thread kernels (which also cover dopri8 / adaptive_heun)

    @property
    def multistep_fused(self):
        """True: 'explicit_adams' / 'fixed_adams' / 'adams' run as one launch (csrc/mi_ode_adams.h, mi_ode_adams_vc.h): the row-local
        catalogue systems and (plugin ABI 2) user-defined row-local systems."""
        return bool(self.row_local)

    def supports(self, y0):
        """True when the fused kernels can take this state tensor."""
        return y0.dim() >= 1 and y0.shape[-1] == self.dim and y0.dtype in (torch.float32, torch.float64)

    def supports_multistep(self, y0):
        """True when the one-launch Adams kernels can take this state tensor (a family may take more there than on its Runge-Kutta kernels)."""
        return self.supports(y0)

    def fill(self, rhs, dtype, device):
        """Fill a _native.Rhs struct; returns objects that must stay alive as long as the handle."""
        rhs.kind = self.kind
        rhs.sign = self.sign
        rhs.hidden = 0
        return []


class _MatRHS(DeviceRHS):
    def __init__(self, W, b=None):
        super(_MatRHS, self).__init__()
        W = torch.as_tensor(W)
        assert W.dim() == 2 and W.shape[0] == W.shape[1], 'W must be square [dim, dim]'
        self.W = W
        self.b = None if b is None else torch.as_tensor(b)
        self.dim = int(W.shape[0])
        self.row_local = self.dim == 2 and self.b is None       # 2x2 systems travel by value to the row-local kernels
        self.tile_dopri8 = isinstance(self, Linear) and 3 <= self.dim <= 128    # the MFMA tile kernels also exist for the 13-row tableau

    @property
    def multistep_fused(self):
        """The Adams family in one launch: 2 x 2 systems on the thread-per-trajectory kernels, any other dim <= 256 with a thread per
        state element (csrc/mi_ode_stage_rowlocal.h: RhsLinearCoop; round 4) - as long as the batch's workgroups are co-resident."""
        return self.row_local or 1 <= self.dim <= 256

    def fill(self, rhs, dtype, device):
        keep = super(_MatRHS, self).fill(rhs, dtype, device)
        Wd = self._dev(self.W, dtype, device)
        rhs.w[0] = Wd.data_ptr()
        keep.append(Wd)
        if self.b is not None:
            bd = self._dev(self.b, dtype, device)
            rhs.b[0] = bd.data_ptr()
            keep.append(bd)
        if self.dim == 2:                        # tiny systems travel by value (row-local kernels)
            flat = self.W.detach().to(torch.float64).reshape(-1).tolist()
            for i in range(4):
                rhs.scalars[i] = flat[i]
        return keep


class Linear(_MatRHS):
    """f(t, y) = y @ W (+ b)."""
    kind = N.RHS_LINEAR

    @classmethod
    def from_matrix(cls, A, b=None):
        """f(t, y) = A y for row-vector states, i.e. y @ A^T."""
        return cls(torch.as_tensor(A).t().contiguous(), b)

    def forward(self, t, y):
        W = self._dev(self.W, y.dtype, y.device)
        out = torch.matmul(y, W)
        if self.b is not None:
            out = out + self._dev(self.b, y.dtype, y.device)
        return out


class CubicLinear(_MatRHS):
    """f(t, y) = (y ** 3) @ W  (the spiral of examples/ode_demo.py)."""
    kind = N.RHS_CUBIC_LINEAR

    def __init__(self, W):
        super(CubicLinear, self).__init__(W, None)

    def forward(self, t, y):
        return torch.matmul(y ** 3, self._dev(self.W, y.dtype, y.device))


class LotkaVolterra(DeviceRHS):
    kind = N.RHS_LOTKA_VOLTERRA
    dim = 2
    row_local = True

    def __init__(self, a=1.5, b=1.0, c=3.0, d=1.0):
        super(LotkaVolterra, self).__init__()
        self.p = (float(a), float(b), float(c), float(d))

    def forward(self, t, y):
        a, b, c, d = self.p
        u, v = y[..., 0], y[..., 1]
        return torch.stack([a * u - b * u * v, -c * v + d * u * v], dim=-1)

    def fill(self, rhs, dtype, device):
        keep = super(LotkaVolterra, self).fill(rhs, dtype, device)
        for i, v in enumerate(self.p):
            rhs.scalars[i] = v
        return keep


class Lorenz(DeviceRHS):
    kind = N.RHS_LORENZ
    dim = 3
    row_local = True

    def __init__(self, sigma=10., beta=8. / 3., rho=28.):
        super(Lorenz, self).__init__()
        self.p = (float(sigma), float(beta), float(rho))

    def forward(self, t, y):
        s, be, r = self.p
        x0, x1, x2 = y[..., 0], y[..., 1], y[..., 2]
        return torch.stack([s * (x1 - x0), x0 * (r - x2) - x1, x0 * x1 - be * x2], dim=-1)

    def fill(self, rhs, dtype, device):
        keep = super(Lorenz, self).fill(rhs, dtype, device)
        for i, v in enumerate(self.p):
            rhs.scalars[i] = v
        return keep


class MLP(DeviceRHS):
    """ODEFunc-shaped MLP dim -> hidden -> hidden -> dim (tfdiffeq/models/dense_odenet.py:41-92).
    Weights are [in, out] (Keras Dense layout).  activation: 'tanh', 'relu' (the reference's default) or 'softplus'.
    time_dependent: the first layer sees concat([t, x]) (dense_odenet.py:79-84): W1 is [dim + 1, hidden], row 0 for t."""
    kind = N.RHS_MLP_TANH
    ACTIVATIONS = {'tanh': 0, 'relu': 1, 'softplus': 2}
    tile_dopri8 = True            # the MFMA tile kernels are instantiated for the 13-row tableau as well (dopri8.py:12-77)

    def __init__(self, W1, b1, W2, b2, W3, b3, activation='tanh', time_dependent=False):
        super(MLP, self).__init__()
        if activation not in self.ACTIVATIONS:
            raise ValueError('the fused MLP kernels know %s, not %r' % (sorted(self.ACTIVATIONS), activation))
        self.activation = activation
        self.time_dependent = bool(time_dependent)
        self.Ws = [torch.as_tensor(w) for w in (W1, W2, W3)]
        self.bs = [None if b is None else torch.as_tensor(b) for b in (b1, b2, b3)]
        self.dim = int(self.Ws[0].shape[0]) - (1 if self.time_dependent else 0)
        self.hidden = int(self.Ws[0].shape[1])
        assert self.Ws[1].shape == (self.hidden, self.hidden) and self.Ws[2].shape == (self.hidden, self.dim)

    def forward(self, t, y):
        act = {'tanh': torch.tanh, 'relu': torch.relu, 'softplus': torch.nn.functional.softplus}[self.activation]
        h = y
        if self.time_dependent:
            t_vec = torch.ones(y.shape[:-1] + (1,), dtype=y.dtype, device=y.device) * torch.as_tensor(t, dtype=y.dtype, device=y.device)
            h = torch.cat([t_vec, y], dim=-1)
        for i in range(3):
            h = torch.matmul(h, self._dev(self.Ws[i], y.dtype, y.device))
            if self.bs[i] is not None:
                h = h + self._dev(self.bs[i], y.dtype, y.device)
            if i < 2:
                h = act(h)
        return h

    fixed_grid_fused = True      # euler / rk4 (3/8 rule): the whole fixed-grid integration in one launch (k_fixed_mlp, round 4)

    MAX_DIM, MAX_HIDDEN = 64, 128     # what the tile kernels are instantiated for: float32 (weight slices resident in registers,
                                      # csrc/mi_ode_mlp.h) and - round 6 - float64 (weights streamed from a packed copy, csrc/mi_ode_mlp64.h)
    _told_limits = set()

    def supports(self, y0):
        """The MFMA tile kernels (Runge-Kutta, adaptive and fixed grid): float32 or float64, dim <= 64, hidden <= 128."""
        return (y0.dim() >= 1 and y0.shape[-1] == self.dim and y0.dtype in (torch.float32, torch.float64)
                and self.dim <= self.MAX_DIM and self.hidden <= self.MAX_HIDDEN)

    def supports_coop(self, y0, any_box=False):
        """Outside that box (round 5): the adaptive Runge-Kutta solvers in ONE launch on the cooperative kernel (a thread per state element,
        the three layers through LDS: csrc/mi_ode_stage_rowlocal.h RhsMlpCoop under k_persist_rowlocal) - float32 / float64, dim and hidden
        up to 256; state in registers while the batch's workgroups (min(256 / dim, 2048 / hidden) trajectories each) are co-resident, streamed
        through HBM planes beyond (csrc/mi_ode_persist.h: k_persist_rowlocal_planes)."""
        # (any_box: a tableau the tile kernels are not instantiated for - adaptive_heun - takes the cooperative kernel inside their box too)
        in_box = self.coop_in_box(y0) or (any_box and y0.dim() >= 1 and y0.shape[-1] == self.dim and y0.dtype in (torch.float32, torch.float64)
                                          and self.dim <= self.MS_MAX_DIM and self.hidden <= self.MS_MAX_HIDDEN)
        if not in_box:
            return False
        # The cooperative evaluation is vector-ALU work fed from L2 (~0.9e12 fma/s measured): it beats the callable engine (three rocBLAS
        # products per evaluation, but 130 - 550 us of launches per attempt) while an evaluation stays under ~50 M multiply-adds -
        # 1500 trajectories of a 64-128-128-64 network, 24 000 of 16-32-32-16; beyond that the callable engine is the faster route
        # (profiles/r05_mlp_coop_kernel_stats.csv: 3.1 ms against 1.7 ms per call at 4096 x 64-128-128-64).
        rows = y0.numel() // max(self.dim, 1)
        return rows * (self.dim * self.hidden + self.hidden * self.hidden + self.hidden * self.dim) <= self.COOP_MAX_FMA

    COOP_MAX_FMA = 5e7

    def coop_in_box(self, y0):
        """What the cooperative kernels can take at all (dtype, widths) - whether they are the faster route is supports_coop's business."""
        return (not self.supports(y0) and y0.dim() >= 1 and y0.shape[-1] == self.dim and y0.dtype in (torch.float32, torch.float64)
                and self.dim <= self.MS_MAX_DIM and self.hidden <= self.MS_MAX_HIDDEN)

    def warn_limits(self, y0, why=''):
        """(round-4 review, item 4) the reference's ODEFunc takes any width and tf.float64 (dense_odenet.py:41-92); where no kernel of this
        family takes a problem the network is served as a Python callable (rocBLAS products, plane kernels, the controller on the device) -
        correct, slower, and not silent: said once per (dtype, dim, hidden)."""
        if y0.dim() < 1 or y0.shape[-1] != self.dim:
            return
        key = (str(y0.dtype), self.dim, self.hidden, why)
        if key not in MLP._told_limits:
            MLP._told_limits.add(key)
            import warnings
            warnings.warn('tfdiffeq_amd.rhs.MLP: the MFMA tile kernels take float32 / float64 states with dim <= %d and hidden <= %d, the cooperative '
                          'one-launch kernel float32 / float64 up to %d wide (every adaptive method, euler / rk4, the Adams family); this problem (%s, dim %d, '
                          'hidden %d%s) runs as a Python callable on the device-controlled engine instead' % (
                              self.MAX_DIM, self.MAX_HIDDEN, self.MS_MAX_DIM, str(y0.dtype).replace('torch.', ''), self.dim, self.hidden,
                              (', ' + why) if why else ''))

    MS_MAX_DIM = MS_MAX_HIDDEN = 256    # the one-launch Adams kernels (csrc/mi_ode_stage_rowlocal.h: RhsMlpCoop): float32 and float64

    @property
    def multistep_fused(self):
        """'explicit_adams' / 'fixed_adams' / 'adams' in one launch (round 5): a thread per state element, the threads of a trajectory
        evaluate the three layers together through LDS - float32 and float64, dim and hidden up to 256."""
        return self.dim <= self.MS_MAX_DIM and self.hidden <= self.MS_MAX_HIDDEN

    def supports_multistep(self, y0):
        return (y0.dim() >= 1 and y0.shape[-1] == self.dim and y0.dtype in (torch.float32, torch.float64)
                and self.dim <= self.MS_MAX_DIM and self.hidden <= self.MS_MAX_HIDDEN)

    def fill(self, rhs, dtype, device):
        keep = super(MLP, self).fill(rhs, dtype, device)
        rhs.hidden = self.hidden
        rhs.scalars[0] = float(self.ACTIVATIONS[self.activation])
        rhs.scalars[1] = 1.0 if self.time_dependent else 0.0
        for i in range(3):
            Wd = self._dev(self.Ws[i], dtype, device)
            rhs.w[i] = Wd.data_ptr()
            keep.append(Wd)
            if self.bs[i] is not None:
                bd = self._dev(self.bs[i], dtype, device)
                rhs.b[i] = bd.data_ptr()
                keep.append(bd)
        return keep


class MLPTanh(MLP):
    """The tanh network (BASELINE config 5)."""

    def __init__(self, W1, b1, W2, b2, W3, b3):
        super(MLPTanh, self).__init__(W1, b1, W2, b2, W3, b3, activation='tanh')


def from_sequential(seq, time_dependent=False):
    """A torch.nn.Sequential of Linear layers with ONE kind of activation between them, as the device descriptor of the fused MLP kernels
    (round-4 review, item 5: a user's own dense network should not have to go through the Python-callable engine):

        Linear(d, h), act, Linear(h, h), act, Linear(h, d)     ->  rhs.MLP, as it is (the shape of the reference's ODEFunc, dense_odenet.py:41-92)
        Linear(d, h), ReLU, Linear(h, d)                       ->  rhs.MLP with an identity middle layer (relu(relu(z)) = relu(z): the same function,
                                                                   bit for bit - relu is idempotent; tanh / softplus are not, and raise)

    act: nn.Tanh, nn.ReLU or nn.Softplus (default beta / threshold).  The descriptor holds detached COPIES of the weights in [in, out]
    layout and the module's own dtype (call again after an optimizer step).  Kernels as rhs.MLP: the MFMA tile kernels for float32 states,
    d <= 64, h <= 128; the cooperative one-launch kernel for float32 / float64 up to 256 wide.  (`odeint(lambda t, y: net(y), y0, t)` gets
    the same kernels by itself - tfdiffeq_amd/lower.py - and reads the weights on every call.)"""
    nn = torch.nn
    layers = list(seq)
    lin = [m for m in layers if isinstance(m, nn.Linear)]
    acts = [m for m in layers if not isinstance(m, nn.Linear)]
    if len(layers) != 2 * len(lin) - 1 or any(isinstance(layers[i], nn.Linear) != (i % 2 == 0) for i in range(len(layers))) or len(lin) not in (2, 3):
        raise ValueError('from_sequential: expected Linear, act, Linear[, act, Linear]')
    kinds = {nn.Tanh: 'tanh', nn.ReLU: 'relu', nn.Softplus: 'softplus'}
    names = {kinds.get(type(a)) for a in acts}
    if len(names) != 1 or None in names:
        raise ValueError('from_sequential: one kind of activation (Tanh, ReLU or Softplus) between the layers')
    act = names.pop()
    if act == 'softplus' and any(a.beta != 1 or a.threshold != 20 for a in acts):
        raise ValueError('from_sequential: Softplus with its default beta / threshold only')
    w = [m.weight.detach().t().contiguous() for m in lin]             # the module's own dtype (fill() converts per state dtype)
    b = [None if m.bias is None else m.bias.detach() for m in lin]
    if len(lin) == 2:
        if act != 'relu':
            raise ValueError('from_sequential: a two-layer network maps onto the three-layer kernel only for ReLU (idempotent)')
        h = w[0].shape[1]
        w = [w[0], torch.eye(h, dtype=w[0].dtype, device=w[0].device), w[1]]
        b = [b[0], None, b[1]]
    return MLP(w[0], b[0], w[1], b[1], w[2], b[2], activation=act, time_dependent=time_dependent)


class PerComponent(object):
    """Lift of a row-local DeviceRHS to TUPLE states: func(t, (y_1, .., y_K)) = (f(t, y_1), .., f(t, y_K)) - what the reference's
    own tuple tests do (tests/problems.py `construct_problem(tuple_state=True)`), K <= 8 components of shape [..., dim] each.
    With a row-local `base` (catalogue or CustomRowLocal) and an adaptive non-pooled method the components travel in one
    buffer with a segment table and the whole integration is one kernel launch: one error ratio per component, all must be
    <= 1, python max() drives the step size (misc.py:250-287) - exactly the reference's tuple semantics.  Otherwise (and on the
    plane-kernel path) it is an ordinary callable."""

    def __init__(self, base):
        if not getattr(base, 'kind', 0):
            raise TypeError('PerComponent wraps a DeviceRHS')
        self.base = base
        self.device_rhs = base
        self.per_component = True

    def __call__(self, t, ys):
        return tuple(self.base(t, y) for y in ys)


# ---------------------------------------------------------------------------------------------
# user-defined device right-hand sides
# ---------------------------------------------------------------------------------------------
_PLUGIN_TEMPLATE = """// generated by tfdiffeq_amd.rhs.CustomRowLocal - do not edit
#define {dtype_macro} 1
#include "mi_ode_plugin.h"
namespace mi {{
template <typename T>
struct RhsUser {{
  static constexpr int D = {dim};
  T p[8];                                                  // mi_ode_rhs.scalars in the state dtype
  __device__ explicit RhsUser(const RhsParams& r) {{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (T)r.s[i];
  }}
  // t: stage time, y[D]: state of this trajectory, k[D]: derivative to fill in
  __device__ __forceinline__ void operator()(T t, const T* y, T* k) const {{
    (void)t;
{body}
  }}
}};
}}  // namespace mi
MI_ODE_DEFINE_ROWLOCAL_PLUGIN(mi::RhsUser)
"""


class CustomRowLocal(DeviceRHS):
    """A trajectory-local system f(t, y) of small dimension written as device code.

    `body` is HIP C++ that fills `k[0..dim-1]` from `t`, `y[0..dim-1]` and the parameters `p[0..7]` (all of the state
    dtype `T`).  It is compiled once (hipcc, gfx950; cached by source hash) into a plugin that instantiates the same
    row-local kernels the built-in catalogue uses, so the system gets the whole-integration kernel (one launch per
    odeint call), the whole-attempt kernel and the one-launch fixed-grid kernel:

        vdp = rhs.CustomRowLocal(2, "k[0] = y[1]; k[1] = p[0] * (1 - y[0] * y[0]) * y[1] - y[0];", params=[5.0])
        sol = odeint(vdp, y0, t, method='dopri5')          # y0: [..., 2] on the GPU

    Write the arithmetic the way the reference's Python callable evaluates it (no FMA contraction is applied).
    `torch_fn(t, y)` is optional: the same function over torch tensors, used where a Python callable is needed (tuple
    states, `odeint_adjoint`, `options={'force_plane_kernels': True}`)."""
    kind = N.RHS_PLUGIN
    MAX_DIM = 32                  # round 5 (was 8): a thread still owns one trajectory - the state and its S + 1 stage derivatives are
    row_local = True              # thread-private.  Up to dim 16 (float64) / 32 (float32) the Dopri5 whole-call kernel keeps them in
                                  # registers (256 VGPRs, <= 160 B of scratch at dim 16 float64); beyond that they spill to scratch
                                  # memory - still one launch per call, but the per-attempt time grows with the spill traffic

    def __init__(self, dim, body, params=(), torch_fn=None):
        super(CustomRowLocal, self).__init__()
        self.dim = int(dim)
        if not 1 <= self.dim <= self.MAX_DIM:
            raise ValueError('CustomRowLocal supports 1 <= dim <= %d (one trajectory per thread, state thread-private)' % self.MAX_DIM)
        self.params = [float(v) for v in params]
        if len(self.params) > 8:
            raise ValueError('at most 8 scalar parameters travel by value')
        self.body = str(body)
        self.torch_fn = torch_fn
        self._plugins = {}

    def forward(self, t, y):
        if self.torch_fn is None:
            raise NotImplementedError('this CustomRowLocal has no torch_fn: only the fused kernels can evaluate it')
        return self.torch_fn(t, y)

    def source(self, dtype):
        body = '\n'.join('    ' + ln for ln in self.body.strip().splitlines())
        return _PLUGIN_TEMPLATE.format(dtype_macro='MI_ODE_PLUGIN_F32' if dtype == torch.float32 else 'MI_ODE_PLUGIN_F64',
                                       dim=self.dim, body=body)

    def _plugin(self, dtype):
        hit = self._plugins.get(dtype)
        if hit is None:
            from . import _plugin_build
            lib = _plugin_build.build_and_load(self.source(dtype))
            table = lib.mi_ode_plugin_get(N.dtype_code(dtype))
            if not table:
                raise N.NativeError('plugin exports no table for %s' % dtype)
            hit = (lib, int(table))
            self._plugins[dtype] = hit
        return hit

    def fill(self, rhs, dtype, device):
        keep = super(CustomRowLocal, self).fill(rhs, dtype, device)
        lib, table = self._plugin(dtype)
        rhs.plugin = table
        for i, v in enumerate(self.params):
            rhs.scalars[i] = v
        keep.append(lib)
        return keep

    def cache_key(self, dtype, device):
        return super(CustomRowLocal, self).cache_key(dtype, device) + (self._plugin(dtype)[1],)


_COOP_TEMPLATE = """// generated by tfdiffeq_amd.rhs.CustomCoop - do not edit
#define {dtype_macro} 1
#include "mi_ode_plugin.h"
namespace mi {{
template <typename T>
struct RhsUserCoop {{
  static constexpr int D = 1;                              // a thread owns ONE state element ...
  static constexpr bool kCoop = true;                      // ... and the threads of a trajectory evaluate f together
  static constexpr int DIM = {dim};
  T p[8];                                                  // mi_ode_rhs.scalars in the state dtype
  const T *w0, *w1, *w2;                                   // mi_ode_rhs.w[0..2]: the caller's device arrays in the state dtype (nullable)
  __device__ explicit RhsUserCoop(const RhsParams& r) : w0((const T*)r.w[0]), w1((const T*)r.w[1]), w2((const T*)r.w[2]) {{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (T)r.s[i];
  }}
  static __host__ __device__ int tpw(const RhsParams&, int) {{ return 256 / DIM; }}     // trajectories per 256-thread workgroup
  // t: stage time; yv[0]: this thread's element; kv[0]: its derivative.  Called from uniform control flow (barriers inside).
  __device__ __forceinline__ void operator()(T t, const T* yv, T* kv) const {{
    __shared__ T s_y[256];
    const int slot = (int)threadIdx.x / DIM, i = (int)threadIdx.x - slot * DIM;
    __syncthreads();                                       // the previous evaluation's readers are done
    s_y[threadIdx.x] = yv[0];
    __syncthreads();
    const T* y = s_y + slot * DIM;                         // the trajectory's whole state, read only
    T k = (T)0;
    if (slot < 256 / DIM) {{
      (void)t; (void)y; (void)i;
{body}
    }}
    kv[0] = k;
  }}
}};
}}  // namespace mi
MI_ODE_DEFINE_COOP_PLUGIN(mi::RhsUserCoop)
"""


class CustomCoop(DeviceRHS):
    """A system f(t, y) of dimension up to 256 written as device code for ONE state element (round 5; round-4 review, item 5: user code
    beyond what a thread can keep - coupled oscillator chains, stencils, a dense layer with the user's own pointwise function).

    `body` is HIP C++ that sets `k` - the derivative of element `i` - from `t`, the trajectory's state `y[0..DIM-1]` (read only, shared
    by the trajectory's threads through LDS), the parameters `p[0..7]` and up to three device arrays `w0`, `w1`, `w2` (`tensors=`, in
    the state dtype).  A thread owns one element; floor(256 / dim) trajectories share a workgroup:

        ring = rhs.CustomCoop(100, "k = p[0] * (y[(i + 1) % DIM] - 2 * y[i] + y[(i + DIM - 1) % DIM]);", params=[0.3])   # heat equation on a ring
        sol = odeint(ring, y0, t, method='dopri5')            # y0: [batch, 100] on the GPU; ONE launch per call

    Kernels: every adaptive Runge-Kutta method as one launch per call at any batch size (csrc/mi_ode_persist.h: k_persist_rowlocal with
    the state in registers while the batch's workgroups are co-resident - about 1000 workgroups: 2000 trajectories of dim 100 -,
    k_persist_rowlocal_planes with the state streamed through HBM planes beyond), euler / rk4 (one launch, any batch) and the Adams
    family (one launch, co-resident grid).  midpoint / heun, tuple states and `odeint_adjoint` need `torch_fn`."""
    kind = N.RHS_PLUGIN
    MAX_DIM = 256
    row_local = False
    fixed_grid_fused = True       # euler / rk4 (3/8 rule) in one launch, any batch size (trajectories never interact on a fixed grid)
    multistep_fused = True
    wide_tableaus = True          # (solvers._make_engine: dopri8 / adaptive_heun exist for this family - the same whole-call kernel)

    def __init__(self, dim, body, params=(), tensors=(), torch_fn=None):
        super(CustomCoop, self).__init__()
        self.dim = int(dim)
        if not 1 <= self.dim <= self.MAX_DIM:
            raise ValueError('CustomCoop supports 1 <= dim <= %d (a thread per state element, 256-thread workgroups)' % self.MAX_DIM)
        self.params = [float(v) for v in params]
        if len(self.params) > 8:
            raise ValueError('at most 8 scalar parameters travel by value')
        self.tensors = [torch.as_tensor(v) for v in tensors]
        if len(self.tensors) > 3:
            raise ValueError('at most 3 device arrays (w0, w1, w2)')
        self.body = str(body)
        self.torch_fn = torch_fn
        self._plugins = {}

    def forward(self, t, y):
        if self.torch_fn is None:
            raise NotImplementedError('this CustomCoop has no torch_fn: only the one-launch kernels (adaptive Runge-Kutta, Adams family) can evaluate it')
        return self.torch_fn(t, y)

    def source(self, dtype):
        body = '\n'.join('      ' + ln for ln in self.body.strip().splitlines())
        return _COOP_TEMPLATE.format(dtype_macro='MI_ODE_PLUGIN_F32' if dtype == torch.float32 else 'MI_ODE_PLUGIN_F64', dim=self.dim, body=body)

    _plugin = CustomRowLocal._plugin

    def fill(self, rhs, dtype, device):
        keep = super(CustomCoop, self).fill(rhs, dtype, device)
        lib, table = self._plugin(dtype)
        rhs.plugin = table
        for i, v in enumerate(self.params):
            rhs.scalars[i] = v
        for i, w in enumerate(self.tensors):
            wd = self._dev(w, dtype, device)
            rhs.w[i] = wd.data_ptr()
            keep.append(wd)
        keep.append(lib)
        return keep

    def cache_key(self, dtype, device):
        return super(CustomCoop, self).cache_key(dtype, device) + (self._plugin(dtype)[1],)

