"""Automatic lowering of Python callables `func(t, y)` onto the fused kernels (round 6; north star: "each RK stage fuses the
USER-SUPPLIED f(t, y)" - every caller of the reference passes a Python callable: tests/problems.py:13-68, examples/ode_demo.py:32-35
and :115-129, examples/lorenz_attractor.py:20-37, tests/DETEST/detest.py).

    trace     `func` is evaluated ONCE on proxy values (`Sym`, dispatched through `__torch_function__` and Python operators): the
              result is a small graph of array operations over the TAIL of the state (the axes of one trajectory; the leading `nb`
              axes of y are the batch and stay symbolic - an operation that would mix trajectories is refused, never guessed).
    classify  y @ W (+ b)                         -> rhs.Linear      (MFMA tile kernels, dim 5 .. 256)
              (y ** 3) @ W                        -> rhs.CubicLinear
              Linear, act, Linear, act, Linear    -> rhs.MLP         (MFMA tile kernels / cooperative kernel)
              anything else, state dim <= 32      -> generated HIP C++ for ONE trajectory per thread, compiled through the plugin ABI
                                                     (csrc/mi_ode_plugin.h, the kernels of rhs.CustomRowLocal)
              anything else, state dim <= 256     -> generated cooperative code, one state element per thread (rhs.CustomCoop's kernels)
    bind      Python floats travel by value (mi_ode_rhs.scalars), tensors the callable closes over are copied into a persistent
              device pool on EVERY call (in-place parameter updates are seen; the compiled code depends on shapes only)

No Triton, no torch.compile, no CPU path: generated code goes through hipcc for gfx950 like a hand-written plugin.  A callable outside
the op set keeps running on the device-controlled callable engine; `odeint.last_stats['lower']` says which way a call went and why.

The arithmetic follows the Python expression operation for operation (no reassociation, no FMA contraction; pow follows torch's
own special cases), so elementwise systems agree with the callable engine to the bit; matrix products are summed in index order.
"""
import hashlib
import math

import numpy as np
import torch

from . import _native as N
from . import rhs as R


class TraceError(Exception):
    """The callable cannot be lowered (the message says why); the caller falls back to the callable engine."""


# ---------------------------------------------------------------------------------------------
# graph
# ---------------------------------------------------------------------------------------------
class Node(object):
    __slots__ = ('id', 'op', 'args', 'attr', 'shape', 'batched', 'is_bool', 'akey')

    def __init__(self, id_, op, args, attr, shape, batched, is_bool, akey):
        self.id, self.op, self.args, self.attr = id_, op, tuple(args), attr
        self.shape, self.batched, self.is_bool, self.akey = tuple(int(s) for s in shape), bool(batched), bool(is_bool), akey

    @property
    def rank(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def __repr__(self):
        return 'n%d=%s%s%s' % (self.id, self.op, list(a.id for a in self.args), list(self.shape))


def _prod(shape):
    return int(np.prod(shape, dtype=np.int64)) if len(shape) else 1


class Trace(object):
    """One evaluation of a callable on proxies: the graph, the scalars and the tensors it read."""

    def __init__(self, full_shape, nb, dtype, device):
        self.full_shape = tuple(int(s) for s in full_shape)
        self.nb = int(nb)
        self.batch_shape = self.full_shape[:self.nb]
        self.tail = self.full_shape[self.nb:]
        self.dtype, self.device = dtype, torch.device(device)
        self.nodes = []
        self._cse = {}
        self.scalars = []             # Python floats, by first use
        self.tensors = []             # dicts: t (the tensor), lead (leading axes dropped at index 0), shape (what the graph sees)
        self._ids = {}
        self._keep = []               # every constant object seen (ids must not be recycled while the trace lives)
        self._par_obj = []            # the object behind scalar i
        self.out = None
        self.time_trace = None        # tuple states: the trace that holds expressions of t alone (imported on use)
        self._imported = {}

    def node(self, op, args, attr, shape, batched, is_bool=False, akey=None):
        akey = attr if akey is None else akey
        key = (op, akey, tuple(a.id for a in args), tuple(shape), bool(batched), bool(is_bool))
        hit = self._cse.get(key)
        if hit is not None:
            return hit
        n = Node(len(self.nodes), op, args, attr, shape, batched, is_bool, akey)
        self.nodes.append(n)
        self._cse[key] = n
        return n

    # -- constants ---------------------------------------------------------------------------
    def lit(self, v):
        return self.node('lit', (), int(v), (), False)

    def litf(self, v):
        return self.node('litf', (), float(v), (), False)

    def par(self, obj):
        """A Python float: a by-value parameter.  The same OBJECT used twice is one parameter (ids are stable while `_keep` holds it)."""
        hit = self._ids.get(('s', id(obj)))
        if hit is not None:
            return hit
        self._keep.append(obj)
        v = float(obj)
        n = self.node('par', (), len(self.scalars), (), False)
        self.scalars.append(v)
        self._par_obj.append(obj)
        self._ids[('s', id(obj))] = n
        return n

    def import_node(self, n, src):
        """A node of the time trace `src` (expressions of t and constants only) re-created here."""
        hit = self._imported.get(n.id)
        if hit is not None:
            return hit
        if n.op == 't':
            out = self.node('t', (), None, (), False)
        elif n.op == 'lit':
            out = self.lit(n.attr)
        elif n.op == 'litf':
            out = self.litf(n.attr)
        elif n.op == 'par':
            out = self.par(src._par_obj[n.attr])
        elif n.op == 'ten':
            e = src.tensors[n.attr]
            out = self._tensor_entry(e['t'], e['lead'], e['shape'], n.batched)
        elif n.op == 'y':
            raise TraceError('a value of another component of the state')
        else:
            out = self.node(n.op, [self.import_node(a, src) for a in n.args], n.attr, n.shape, n.batched, n.is_bool, n.akey)
        self._imported[n.id] = out
        return out

    def tensor(self, x, mode='ew'):
        """A real tensor the callable closed over.  mode 'ew': operand of an elementwise operation (leading axes that ARE the batch axes
        must be uniform along them and are dropped); 'mat': operand of a matrix product (taken as it is)."""
        if x.is_complex():
            raise TraceError('complex constant')
        if x.dim() == 0 and x.device.type == 'cpu':
            if x.dtype.is_floating_point:
                return self.par(x) if not x.requires_grad else self._tensor_entry(x, 0, (), False)
            return self.lit(int(x.item()))
        s = tuple(int(v) for v in x.shape)
        lead, batched = 0, False
        if mode == 'ew' and self.nb > 0 and len(s) >= self.nb and s[:self.nb] == self.batch_shape:
            if _prod(self.batch_shape) > 1:
                first = x[(slice(0, 1),) * self.nb]
                if not bool((x == first).all().item()):
                    raise TraceError('a constant tensor of shape %s varies along the batch axes' % (list(s),))
            lead, batched, s = self.nb, True, s[self.nb:]
        return self._tensor_entry(x, lead, s, batched)

    def _tensor_entry(self, x, lead, shape, batched):
        key = ('t', id(x), lead, batched)
        hit = self._ids.get(key)
        if hit is not None:
            return hit
        self._keep.append(x)
        idx = len(self.tensors)
        self.tensors.append({'t': x, 'lead': lead, 'shape': tuple(shape)})
        n = self.node('ten', (), idx, shape, batched)
        if x.dtype == torch.bool:
            n = self.node('ew', (n, self.lit(0)), ('ne', None), shape, batched, is_bool=True)
        self._ids[key] = n
        return n

    def lift(self, x, mode='ew'):
        if isinstance(x, Sym):
            if x.tr is not self:
                if x.tr is self.time_trace and self.time_trace is not None:
                    return self.import_node(x.node, x.tr)
                raise TraceError('the components of a tuple state interact (a value of another component)')
            return x.node
        if isinstance(x, (bool, np.bool_)):
            return self.lit(int(x))
        if isinstance(x, (int, np.integer)):
            return self.lit(int(x))
        if isinstance(x, (float, np.floating)):
            return self.par(x)
        if isinstance(x, torch.Tensor):
            return self.tensor(x, mode)
        if isinstance(x, np.ndarray):
            t_ = torch.as_tensor(x)
            self._keep.append(x)
            return self.tensor(t_, mode)
        raise TraceError('operand of type %s' % type(x).__name__)

    # -- structure ---------------------------------------------------------------------------
    def key(self):
        """Structural identity of the traced function: the compiled code depends on nothing else."""
        memo = getattr(self, '_key_memo', None)
        if memo is not None and memo[0] == (len(self.nodes), None if self.out is None else self.out.id):
            return memo[1]
        h = hashlib.sha256()
        h.update(repr((self.nb, self.tail, str(self.dtype))).encode())
        for n in self.nodes:
            ak = n.akey
            if isinstance(ak, tuple):
                ak = tuple(a.tobytes() if isinstance(a, np.ndarray) else a for a in ak)
            elif isinstance(ak, np.ndarray):
                ak = ak.tobytes()
            h.update(repr((n.op, ak, tuple(a.id for a in n.args), n.shape, n.batched, n.is_bool)).encode())
        h.update(repr(self.out.id if self.out is not None else None).encode())
        self._key_memo = ((len(self.nodes), None if self.out is None else self.out.id), h.hexdigest()[:32])
        return self._key_memo[1]

    def live(self):
        """Nodes the output depends on, in creation (= topological) order."""
        seen, todo = set(), [self.out]
        while todo:
            n = todo.pop()
            if n.id in seen:
                continue
            seen.add(n.id)
            todo.extend(n.args)
        return [n for n in self.nodes if n.id in seen]


# ---------------------------------------------------------------------------------------------
# proxy
# ---------------------------------------------------------------------------------------------
_UNARY = ('neg', 'abs', 'sin', 'cos', 'tan', 'exp', 'log', 'sqrt', 'tanh', 'sinh', 'cosh', 'asin', 'acos', 'atan', 'log1p', 'expm1',
          'exp2', 'log2', 'log10', 'erf', 'floor', 'ceil', 'sigmoid', 'relu', 'softplus', 'square', 'reciprocal', 'rsqrt', 'sign', 'silu')
_UNARY_ALIAS = {'negative': 'neg', 'absolute': 'abs', 'arcsin': 'asin', 'arccos': 'acos', 'arctan': 'atan', '__neg__': 'neg', '__abs__': 'abs',
                'positive': 'pos', '__pos__': 'pos'}
_BINARY = {'add': 'add', '__add__': 'add', 'sub': 'sub', 'subtract': 'sub', '__sub__': 'sub', 'mul': 'mul', 'multiply': 'mul', '__mul__': 'mul',
           'div': 'div', 'divide': 'div', 'true_divide': 'div', '__truediv__': 'div', 'maximum': 'max', 'minimum': 'min', 'fmax': 'max', 'fmin': 'min',
           'atan2': 'atan2', 'arctan2': 'atan2'}
_RBINARY = {'__radd__': 'add', '__rsub__': 'sub', 'rsub': 'sub', '__rmul__': 'mul', '__rtruediv__': 'div', '__rdiv__': 'div'}
_COMPARE = {'gt': 'gt', 'greater': 'gt', '__gt__': 'gt', 'lt': 'lt', 'less': 'lt', '__lt__': 'lt', 'ge': 'ge', 'greater_equal': 'ge', '__ge__': 'ge',
            'le': 'le', 'less_equal': 'le', '__le__': 'le', 'eq': 'eq', '__eq__': 'eq', 'ne': 'ne', 'not_equal': 'ne', '__ne__': 'ne'}
_IDENTITY = ('clone', 'contiguous', 'detach', 'requires_grad_', 'pos')


class Sym(object):
    """A traced value: behaves like the tensor the callable expects for the operations this module knows, raises TraceError otherwise."""
    __slots__ = ('tr', 'node')
    __array_priority__ = 1000.0
    __array_ufunc__ = None

    def __init__(self, tr, node):
        self.tr, self.node = tr, node

    # -- what callables ask a tensor -------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size((self.tr.batch_shape if self.node.batched else ()) + self.node.shape)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return len(self.shape)

    ndim = property(dim)
    ndimension = dim

    def numel(self):
        return _prod(tuple(self.shape))

    nelement = numel

    @property
    def dtype(self):
        return torch.bool if self.node.is_bool else self.tr.dtype

    @property
    def device(self):
        return self.tr.device

    @property
    def is_cuda(self):
        return self.tr.device.type == 'cuda'

    requires_grad = False
    grad_fn = None
    is_leaf = True

    def is_floating_point(self):
        return not self.node.is_bool

    def is_complex(self):
        return False

    def get_device(self):
        return self.tr.device.index if self.tr.device.index is not None else -1

    def __len__(self):
        s = self.shape
        if not s:
            raise TypeError('len() of a 0-d tensor')
        return s[0]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __bool__(self):
        raise TraceError('data-dependent control flow (bool() of a traced value)')

    def __float__(self):
        raise TraceError('the callable reads a traced value on the host (float())')

    __int__ = __index__ = __float__

    def item(self):
        raise TraceError('the callable reads a traced value on the host (.item())')

    tolist = numpy = cpu = item

    def __hash__(self):
        return id(self)

    # -- dispatch ----------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', None) or str(func)
        return _dispatch(name, args, kwargs or {})

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        if name.endswith('_') and not name.endswith('__'):
            raise TraceError('in-place operation .%s()' % name)

        def method(*args, **kwargs):
            return _dispatch(name, (self,) + args, kwargs)
        method.__name__ = name
        return method

    @property
    def T(self):
        return _dispatch('t' if self.dim() <= 2 else 'permute', (self,) + (() if self.dim() <= 2 else (tuple(range(self.dim() - 1, -1, -1)),)), {})

    @property
    def mT(self):
        return _dispatch('transpose', (self, -2, -1), {})

    @property
    def data(self):
        return self

    def __getitem__(self, index):
        return _getitem(self, index)

    def __setitem__(self, index, value):
        raise TraceError('in-place assignment into a traced value')

    def __matmul__(self, o):
        return _dispatch('matmul', (self, o), {})

    def __rmatmul__(self, o):
        return _dispatch('matmul', (o, self), {})

    def __pow__(self, o):
        return _dispatch('pow', (self, o), {})

    def __rpow__(self, o):
        return _dispatch('pow', (o, self), {})


def _install_operators():
    for name in ('__add__', '__sub__', '__mul__', '__truediv__', '__gt__', '__lt__', '__ge__', '__le__', '__eq__', '__ne__', '__neg__', '__abs__', '__pos__'):
        def op(self, *o, _n=name):
            return _dispatch(_n, (self,) + o, {})
        setattr(Sym, name, op)
    for name in ('__radd__', '__rsub__', '__rmul__', '__rtruediv__'):
        def rop(self, o, _n=name):
            return _dispatch(_n, (self, o), {})
        setattr(Sym, name, rop)


_install_operators()


def _trace_of(args):
    todo = list(args)
    found = None
    while todo:
        a = todo.pop()
        if isinstance(a, Sym):
            if a.tr.time_trace is not None or found is None:     # (a component's trace wins over the time trace of a tuple state)
                found = a.tr
                if a.tr.time_trace is not None:
                    return found
        elif isinstance(a, (list, tuple)):
            todo.extend(a)
    if found is None:
        raise TraceError('no traced operand')
    return found


def _ew(tr, fn, vals, attr=None, out_bool=False):
    nodes = [tr.lift(v, 'ew') for v in vals]
    if tr.nb > 0:
        bn = [n for n in nodes if n.batched]
        if bn:
            r = bn[0].rank
            if any(n.rank != r for n in bn):
                raise TraceError('elementwise operation between per-trajectory values of rank %s (it would align a batch axis with a state axis)'
                                 % sorted({n.rank for n in bn}))
            for i, n in enumerate(nodes):
                while not n.batched and n.rank > r:
                    if n.shape[0] != 1:
                        raise TraceError('a constant of shape %s against a per-trajectory value of rank %d' % (list(n.shape), r))
                    n = _move(tr, n, lambda a: a[0])
                nodes[i] = n
    try:
        shape = np.broadcast_shapes(*[n.shape for n in nodes])
    except ValueError:
        raise TraceError('shapes %s do not broadcast' % [list(n.shape) for n in nodes])
    return Sym(tr, tr.node('ew', nodes, (fn, attr), shape, any(n.batched for n in nodes), is_bool=out_bool))


def _move(tr, node, fn):
    """Pure data movement on the tail, expressed as a gather: `fn` maps the array of flat input positions to the output arrangement."""
    src = np.arange(node.size, dtype=np.int64).reshape(node.shape)
    try:
        idx = np.asarray(fn(src), dtype=np.int64)
    except (IndexError, ValueError, TypeError) as e:
        raise TraceError('indexing / reshaping the state: %s' % e)
    if idx.shape == node.shape and np.array_equal(idx, src):
        return node
    return tr.node('gather', (node,), (None, idx), idx.shape, node.batched, is_bool=node.is_bool, akey=('g1', idx.shape, idx.tobytes()))


def _nbn(tr, node):
    return tr.nb if node.batched else 0


def _tail_axis(tr, node, dim, extra=0):
    """A dimension of the full shape as an axis of the tail (extra: 1 for insertions).  Batch axes are refused."""
    nbn = _nbn(tr, node)
    full = nbn + node.rank + extra
    d = int(dim)
    if d < -full or d >= full:
        raise TraceError('dimension %d out of range' % d)
    d = d + full if d < 0 else d
    if d < nbn:
        raise TraceError('operates on a batch axis (dim %d of a value with %d batch axes)' % (dim, nbn))
    return d - nbn


def _getitem(x, index):
    tr, node = x.tr, x.node
    if not isinstance(index, tuple):
        index = (index,)
    idx = []
    for e in index:
        if isinstance(e, Sym):
            raise TraceError('indexing with a traced value')
        if isinstance(e, torch.Tensor):
            if e.dtype == torch.bool:
                raise TraceError('boolean mask indexing')
            e = e.detach().cpu().numpy()
        idx.append(e)
    nbn = _nbn(tr, node)
    full = nbn + node.rank
    n_spec = sum(1 for e in idx if e is not None and e is not Ellipsis)
    if any(e is Ellipsis for e in idx):
        k = [i for i, e in enumerate(idx) if e is Ellipsis]
        if len(k) > 1:
            raise TraceError('two ellipses in an index')
        idx = idx[:k[0]] + [slice(None)] * max(full - n_spec, 0) + idx[k[0] + 1:]
    out, consumed = [], 0
    for e in idx:
        if consumed < nbn:
            if isinstance(e, slice) and e == slice(None):
                consumed += 1
                continue
            raise TraceError('indexes a batch axis')
        out.append(e)
    return Sym(tr, _move(tr, node, lambda a: a[tuple(out)]))


def _shape_args(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        args = tuple(args[0])
    return [int(a) for a in args]


def _reshape(x, shape):
    tr, node = x.tr, x.node
    shape = list(shape)
    nbn = _nbn(tr, node)
    total = _prod(tr.batch_shape if node.batched else ()) * node.size
    if shape.count(-1) > 1:
        raise TraceError('reshape with two -1')
    if -1 in shape:
        rest = _prod([s for s in shape if s != -1])
        if rest == 0 or total % rest:
            raise TraceError('reshape: sizes do not match')
        shape[shape.index(-1)] = total // rest
    if _prod(shape) != total:
        raise TraceError('reshape: sizes do not match')
    if nbn:
        if tuple(shape[:nbn]) != tr.batch_shape:
            raise TraceError('reshape changes the batch axes')
        shape = shape[nbn:]
    return Sym(tr, _move(tr, node, lambda a: a.reshape(shape)))


def _matmul(tr, a, b):
    na = tr.lift(a, 'mat')
    nb_ = tr.lift(b, 'mat')
    if na.rank == 0 or nb_.rank == 0:
        raise TraceError('matmul with a 0-d operand')
    if tr.nb > 0:
        if na.batched and nb_.batched and (na.rank < 2 or nb_.rank < 2):
            raise TraceError('matmul of two per-trajectory values of rank < 2')
        if nb_.batched and not na.batched and nb_.rank < 2:
            raise TraceError('constant @ per-trajectory vector: the product would run over the batch axis (write y @ A.T or A @ y[..., None])')
        if not na.batched and na.rank > 2 or not nb_.batched and nb_.rank > 2:
            raise TraceError('matmul with a constant of rank > 2')
    try:
        shape = np.matmul(np.empty(na.shape, dtype=np.int8), np.empty(nb_.shape, dtype=np.int8)).shape
    except ValueError as e:
        raise TraceError('matmul: %s' % e)
    return Sym(tr, tr.node('matmul', (na, nb_), None, shape, na.batched or nb_.batched))


def _sum(tr, x, dim=None, keepdim=False, mean=False):
    node = tr.lift(x)
    if dim is None or (isinstance(dim, (tuple, list)) and len(dim) == 0):
        if _nbn(tr, node):
            raise TraceError('a reduction over all axes includes the batch axes')
        axes = tuple(range(node.rank))
    else:
        dims = dim if isinstance(dim, (tuple, list)) else (dim,)
        axes = tuple(sorted({_tail_axis(tr, node, d) for d in dims}))
    if node.rank == 0:
        return Sym(tr, node)
    shape = tuple(1 if i in axes else s for i, s in enumerate(node.shape)) if keepdim else tuple(s for i, s in enumerate(node.shape) if i not in axes)
    out = Sym(tr, tr.node('sum', (node,), (axes, bool(keepdim)), shape, node.batched))
    if mean:
        out = _ew(tr, 'div', (out, _prod([node.shape[i] for i in axes])))
    return out


def _cat(tr, items, dim, stack):
    nodes = [tr.lift(v, 'ew') for v in items]
    if not nodes:
        raise TraceError('cat of nothing')
    if tr.nb > 0 and any(n.batched for n in nodes) and not all(n.batched for n in nodes):
        raise TraceError('stack / cat of per-trajectory values with constants that have no batch axes')
    batched = any(n.batched for n in nodes)
    ref = nodes[0]
    ax = _tail_axis(tr, ref, dim, extra=1 if stack else 0)
    srcs, idxs = [], []
    for k, n in enumerate(nodes):
        a = np.arange(n.size, dtype=np.int64).reshape(n.shape)
        srcs.append(np.full(n.shape, k, dtype=np.int64))
        idxs.append(a)
    try:
        if stack:
            src, idx = np.stack(srcs, ax), np.stack(idxs, ax)
        else:
            src, idx = np.concatenate(srcs, ax), np.concatenate(idxs, ax)
    except ValueError as e:
        raise TraceError('stack / cat: %s' % e)
    uniq = []
    remap = {}
    for k, n in enumerate(nodes):                       # the same node stacked twice is one source
        if n.id not in remap:
            remap[n.id] = len(uniq)
            uniq.append(n)
    lut = np.array([remap[n.id] for n in nodes], dtype=np.int64)
    src = lut[src]
    return Sym(tr, tr.node('gather', uniq, (src, idx), idx.shape, batched, is_bool=all(n.is_bool for n in nodes),
                           akey=('gN', idx.shape, src.tobytes(), idx.tobytes())))


def _pow(tr, a, b):
    if isinstance(b, (int, float, np.integer, np.floating)) and not isinstance(b, bool):
        e = float(b)
        # torch's own special cases (aten pow_tensor_scalar): the exponent of a power is STRUCTURE, not a parameter
        if e == 1.0:
            return Sym(tr, tr.lift(a))
        special = {2.0: 'square', 3.0: 'cube', 0.5: 'sqrt', -0.5: 'rsqrt', -1.0: 'reciprocal', -2.0: 'rsquare', 0.0: 'one'}.get(e)
        if special is not None:
            return _ew(tr, special, (a,))
        return _ew(tr, 'powc', (a,), attr=e)
    return _ew(tr, 'pow', (a, b))


def _dispatch(name, args, kw):
    tr = _trace_of(args) if not kw else _trace_of(list(args) + list(kw.values()))
    if 'out' in kw and kw['out'] is not None:
        raise TraceError('%s(out=...)' % name)
    name = _UNARY_ALIAS.get(name, name)
    if name in _IDENTITY:
        return args[0]
    if name in _UNARY:
        if name == 'softplus':
            beta, thr = kw.get('beta', args[1] if len(args) > 1 else 1.0), kw.get('threshold', args[2] if len(args) > 2 else 20.0)
            if float(beta) != 1.0 or float(thr) != 20.0:
                raise TraceError('softplus with non-default beta / threshold')
        if name == 'relu' and (kw.get('inplace') or (len(args) > 1 and args[1])):
            raise TraceError('in-place relu')
        return _ew(tr, name, (args[0],))
    if name in _BINARY:
        fn = _BINARY[name]
        a, b = args[0], args[1]
        alpha = kw.get('alpha', 1)
        if kw.get('rounding_mode') is not None:
            raise TraceError('div with a rounding mode')
        if alpha != 1:
            b = _ew(tr, 'mul', (b, alpha))
        return _ew(tr, fn, (a, b))
    if name in _RBINARY:
        return _ew(tr, _RBINARY[name], (args[1], args[0]))
    if name in _COMPARE:
        return _ew(tr, _COMPARE[name], (args[0], args[1]), out_bool=True)
    if name in ('pow', '__pow__'):
        return _pow(tr, args[0], args[1] if len(args) > 1 else kw['exponent'])
    if name == '__rpow__':
        return _pow(tr, args[1], args[0])
    if name in ('logical_not', '__invert__', 'bitwise_not'):
        return _ew(tr, 'not', (args[0],), out_bool=True)
    if name in ('logical_and', '__and__', 'bitwise_and', 'logical_or', '__or__', 'bitwise_or'):
        return _ew(tr, 'and' if 'and' in name else 'or', (args[0], args[1]), out_bool=True)
    if name == 'where':
        if len(args) != 3:
            raise TraceError('where(condition) without values')
        c = tr.lift(args[0])
        if not c.is_bool:
            raise TraceError('where() on a non-boolean condition')
        return _ew(tr, 'where', (args[0], args[1], args[2]))
    if name in ('clamp', 'clip'):
        lo = kw.get('min', args[1] if len(args) > 1 else None)
        hi = kw.get('max', args[2] if len(args) > 2 else None)
        out = args[0]
        if lo is not None:
            out = _ew(tr, 'max', (out, lo))
        if hi is not None:
            out = _ew(tr, 'min', (out, hi))
        return out
    if name in ('elu', 'leaky_relu'):
        a = kw.get('alpha', kw.get('negative_slope', args[1] if len(args) > 1 else (1.0 if name == 'elu' else 0.01)))
        x = args[0]
        neg = _ew(tr, 'mul', (_ew(tr, 'expm1', (x,)), a)) if name == 'elu' else _ew(tr, 'mul', (x, a))
        return _ew(tr, 'where', (_ew(tr, 'gt', (x, 0), out_bool=True), x, neg))
    if name == 'dropout':
        if kw.get('training', args[2] if len(args) > 2 else True):
            raise TraceError('dropout in training mode')
        return args[0]
    # ---- constructors ----
    if name in ('zeros_like', 'ones_like', 'full_like', 'new_zeros_like'):
        n = tr.lift(args[0])
        v = 0 if name.startswith('zeros') else 1 if name.startswith('ones') else (args[1] if len(args) > 1 else kw['fill_value'])
        vn = tr.lift(v)
        return Sym(tr, tr.node('gather', (vn,), (None, np.zeros(n.shape, dtype=np.int64)), n.shape, n.batched,
                               akey=('g1', n.shape, np.zeros(n.shape, dtype=np.int64).tobytes())))
    if name in ('new_zeros', 'new_ones', 'new_full', 'new_tensor', 'new_empty'):
        fn = getattr(torch.zeros(0, dtype=tr.dtype, device=tr.device), name)
        return fn(*args[1:], **kw)
    # ---- dtype / device: the trace has one dtype ----
    if name in ('to', 'type', 'float', 'double', 'half', 'bfloat16', 'type_as', 'cuda'):
        want = {'float': torch.float32, 'double': torch.float64, 'half': torch.float16, 'bfloat16': torch.bfloat16}.get(name)
        for a in list(args[1:]) + list(kw.values()):
            if isinstance(a, torch.dtype):
                want = a
            elif isinstance(a, (torch.Tensor, Sym)) and name in ('to', 'type_as'):
                want = a.dtype
        if want is not None and want != args[0].dtype:
            raise TraceError('cast of the state to %s' % want)
        return args[0]
    # ---- data movement ----
    x = args[0]
    if name in ('reshape', 'view'):
        if len(args) == 2 and isinstance(args[1], torch.dtype):
            raise TraceError('view(dtype)')
        return _reshape(x, _shape_args(args[1:]) if len(args) > 1 else _shape_args((kw['shape'],)))
    if name == 'view_as' or name == 'reshape_as':
        return _reshape(x, list(args[1].shape))
    if name == 'flatten':
        node = x.node
        s, e = kw.get('start_dim', args[1] if len(args) > 1 else 0), kw.get('end_dim', args[2] if len(args) > 2 else -1)
        if node.rank == 0:
            return _reshape(x, [-1] if not _nbn(tr, node) else list(tr.batch_shape) + [1])
        s, e = _tail_axis(tr, node, s), _tail_axis(tr, node, e)
        return Sym(tr, _move(tr, node, lambda a: a.reshape(a.shape[:s] + (-1,) + a.shape[e + 1:])))
    if name == 'unsqueeze':
        ax = _tail_axis(tr, x.node, args[1] if len(args) > 1 else kw['dim'], extra=1)
        return Sym(tr, _move(tr, x.node, lambda a: np.expand_dims(a, ax)))
    if name == 'squeeze':
        d = args[1] if len(args) > 1 else kw.get('dim')
        if d is None:
            if _nbn(tr, x.node) and 1 in tr.batch_shape:
                raise TraceError('squeeze() would drop a batch axis of size 1')
            return Sym(tr, _move(tr, x.node, lambda a: a.reshape([s for s in a.shape if s != 1])))
        ax = _tail_axis(tr, x.node, d)
        return Sym(tr, _move(tr, x.node, lambda a: a.reshape(a.shape[:ax] + a.shape[ax + 1:]) if a.shape[ax] == 1 else a))
    if name in ('transpose', 'swapaxes', 'swapdims'):
        a0, a1 = _tail_axis(tr, x.node, args[1]), _tail_axis(tr, x.node, args[2])
        return Sym(tr, _move(tr, x.node, lambda a: np.swapaxes(a, a0, a1)))
    if name == 't':
        if x.dim() > 2:
            raise TraceError('t() of a value with more than 2 axes')
        if x.dim() < 2:
            return x
        return _dispatch('transpose', (x, 0, 1), {})
    if name == 'permute':
        dims = _shape_args(args[1:]) if len(args) > 1 else list(kw['dims'])
        nbn = _nbn(tr, x.node)
        full = nbn + x.node.rank
        dims = [d + full if d < 0 else d for d in dims]
        if sorted(dims) != list(range(full)) or dims[:nbn] != list(range(nbn)):
            raise TraceError('permute moves a batch axis')
        perm = [d - nbn for d in dims[nbn:]]
        return Sym(tr, _move(tr, x.node, lambda a: np.transpose(a, perm)))
    if name == 'roll':
        shifts = kw.get('shifts', args[1] if len(args) > 1 else None)
        dims = kw.get('dims', args[2] if len(args) > 2 else None)
        if dims is None:
            if _nbn(tr, x.node):
                raise TraceError('roll over the flattened value includes the batch axes')
            return Sym(tr, _move(tr, x.node, lambda a: np.roll(a, shifts)))
        dims_ = dims if isinstance(dims, (tuple, list)) else (dims,)
        axes = tuple(_tail_axis(tr, x.node, d) for d in dims_)
        sh = tuple(shifts) if isinstance(shifts, (tuple, list)) else (shifts,) * len(axes)
        return Sym(tr, _move(tr, x.node, lambda a: np.roll(a, sh, axes)))
    if name == 'flip':
        dims = _shape_args(args[1:]) if len(args) > 1 else list(kw['dims'])
        axes = tuple(_tail_axis(tr, x.node, d) for d in dims)
        return Sym(tr, _move(tr, x.node, lambda a: np.flip(a, axes)))
    if name in ('expand', 'broadcast_to', 'expand_as'):
        shape = list(args[1].shape) if name == 'expand_as' else _shape_args(args[1:])
        nbn = _nbn(tr, x.node)
        if len(shape) < nbn + x.node.rank:
            raise TraceError('expand to fewer axes')
        extra = len(shape) - (nbn + x.node.rank)
        if nbn:
            if extra or any(s not in (-1, b) for s, b in zip(shape[:nbn], tr.batch_shape)):
                raise TraceError('expand changes the batch axes')
            shape = shape[nbn:]
            extra = 0
        tgt = [(x.node.shape[i - extra] if s == -1 else s) for i, s in enumerate(shape)]
        return Sym(tr, _move(tr, x.node, lambda a: np.broadcast_to(a, tgt)))
    if name in ('stack', 'cat', 'concat', 'concatenate'):
        items = args[0]
        dim = kw.get('dim', kw.get('axis', args[1] if len(args) > 1 else 0))
        return _cat(tr, list(items), dim, name == 'stack')
    if name == 'unbind':
        dim = kw.get('dim', args[1] if len(args) > 1 else 0)
        ax = _tail_axis(tr, x.node, dim)
        return tuple(Sym(tr, _move(tr, x.node, lambda a, i=i: np.take(a, i, axis=ax))) for i in range(x.node.shape[ax]))
    if name in ('chunk', 'split', 'tensor_split'):
        dim = kw.get('dim', args[2] if len(args) > 2 else 0)
        ax = _tail_axis(tr, x.node, dim)
        n = x.node.shape[ax]
        arg = args[1] if len(args) > 1 else kw.get('chunks', kw.get('split_size_or_sections', kw.get('split_size')))
        if name == 'chunk':
            size = -(-n // int(arg))
            cuts = list(range(0, n, size))
            ends = [min(c + size, n) for c in cuts]
        elif isinstance(arg, (list, tuple)):
            ends = list(np.cumsum(arg))
            cuts = [0] + ends[:-1]
        else:
            cuts = list(range(0, n, int(arg)))
            ends = [min(c + int(arg), n) for c in cuts]
        return tuple(Sym(tr, _move(tr, x.node, lambda a, c=c, e=e: np.take(a, np.arange(c, e), axis=ax))) for c, e in zip(cuts, ends))
    if name in ('select', 'narrow', 'index_select'):
        ax = _tail_axis(tr, x.node, args[1])
        if name == 'select':
            return Sym(tr, _move(tr, x.node, lambda a: np.take(a, int(args[2]), axis=ax)))
        if name == 'narrow':
            return Sym(tr, _move(tr, x.node, lambda a: np.take(a, np.arange(int(args[2]), int(args[2]) + int(args[3])), axis=ax)))
        ind = args[2].detach().cpu().numpy() if isinstance(args[2], torch.Tensor) else np.asarray(args[2])
        return Sym(tr, _move(tr, x.node, lambda a: np.take(a, ind, axis=ax)))
    # ---- reductions / products ----
    if name in ('sum', 'mean'):
        dim = kw.get('dim', kw.get('axis', args[1] if len(args) > 1 else None))
        keep = kw.get('keepdim', kw.get('keepdims', args[2] if len(args) > 2 else False))
        if isinstance(dim, torch.dtype):
            dim = None
        return _sum(tr, x, dim, keep, mean=(name == 'mean'))
    if name == 'norm' or name == 'vector_norm':
        p = kw.get('p', kw.get('ord', args[1] if len(args) > 1 else 2))
        if p not in (2, 2.0, 'fro', None):
            raise TraceError('norm with p = %r' % (p,))
        dim = kw.get('dim', args[2] if len(args) > 2 else None)
        return _ew(tr, 'sqrt', (_sum(tr, _ew(tr, 'square', (x,)), dim, kw.get('keepdim', False)),))
    if name in ('matmul', '__matmul__', 'mm', 'mv', 'bmm'):
        return _matmul(tr, args[0], args[1])
    if name == '__rmatmul__':
        return _matmul(tr, args[1], args[0])
    if name in ('dot', 'inner', 'vdot'):
        return _sum(tr, _ew(tr, 'mul', (args[0], args[1])), -1)
    if name == 'linear':
        inp, w = args[0], args[1]
        b = args[2] if len(args) > 2 else kw.get('bias')
        if isinstance(w, Sym) or isinstance(b, Sym) or not isinstance(w, torch.Tensor) or w.dim() != 2:
            out = _matmul(tr, inp, _dispatch('t', (w,), {}) if isinstance(w, Sym) else w.t())
            return out if b is None else _ew(tr, 'add', (out, b))
        xn, wn = tr.lift(inp), tr.lift(w, 'mat')
        if xn.rank < 1 or xn.shape[-1] != wn.shape[1]:
            raise TraceError('linear: input of shape %s against a weight of shape %s' % (list(xn.shape), list(wn.shape)))
        bn = None if b is None else tr.lift(b, 'mat')
        if bn is not None and bn.shape != (wn.shape[0],):
            raise TraceError('linear: bias of shape %s' % (list(bn.shape),))
        return Sym(tr, tr.node('linear', (xn, wn) + (() if bn is None else (bn,)), None, xn.shape[:-1] + (wn.shape[0],), xn.batched))
    if name in ('addmm', 'addmv'):
        return _ew(tr, 'add', (args[0], _matmul(tr, args[1], args[2])))
    raise TraceError('operation `%s` is outside the op set' % name)


# ---------------------------------------------------------------------------------------------
# tracing a callable
# ---------------------------------------------------------------------------------------------
def _finish(tr, out, shape):
    """The callable's result for one state tensor -> tr.out (expanded to the state's shape where it is f(t) only or broadcastable)."""
    if not isinstance(out, Sym):
        if isinstance(out, torch.Tensor):
            raise TraceError('the result does not depend on t or y through traced operations')
        raise TraceError('the callable returned %s' % type(out).__name__)
    node = tr.lift(out)                                      # (tuple states: an expression of t alone lives in the time trace)
    if node.is_bool:
        raise TraceError('boolean result')
    if not node.batched or node.shape != tr.tail:             # f(t) only, or a broadcastable result: expand to the state's shape
        if node.batched and node.rank != len(tr.tail):
            raise TraceError('result of shape %s for a state of shape %s' % (list(out.shape), list(shape)))
        tail = tr.tail

        def fit(a):
            while a.ndim > len(tail) and a.shape[0] == 1:
                a = a[0]
            return np.broadcast_to(a, tail)
        try:
            fitted = _move(tr, node, fit)
        except TraceError:
            raise TraceError('result of shape %s for a state of shape %s' % (list(out.shape), list(shape)))
        node = fitted
        if not node.batched:
            node = tr.node('gather', (node,), (None, np.arange(node.size, dtype=np.int64).reshape(node.shape)), node.shape, True,
                           akey=('gb', node.shape))
    tr.out = node
    return tr


def trace(func, y0, nb=None, t_dtype=None):
    """Evaluate `func(t, y)` on proxies.  y0: the state tensor (only shape, dtype and device are used).  nb: number of leading batch axes
    (default: all but the last; on "indexes / operates on a batch axis" fewer are tried, down to none - the whole tensor as ONE system -
    while that system stays small enough for a kernel)."""
    shape = tuple(int(s) for s in y0.shape)
    tries = [nb] if nb is not None else list(range(max(len(shape) - 1, 0), -1, -1))
    last = None
    for k in tries:
        if _prod(shape[k:]) > MAX_COOP_DIM:
            break
        tr = Trace(shape, k, y0.dtype, y0.device)
        y = Sym(tr, tr.node('y', (), 0, tr.tail, True))
        t = Sym(tr, tr.node('t', (), None, (), False))
        try:
            out = func(t, y)
            if isinstance(out, (tuple, list)) and len(out) == 1:
                out = out[0]
            return _finish(tr, out, shape)
        except TraceError as e:
            last = e
            if 'batch axis' not in str(e) and 'batch axes' not in str(e):
                raise
    if last is None:
        raise TraceError('state of %d elements per trajectory (the generated kernels take up to %d)' % (_prod(shape[-1:]), MAX_COOP_DIM))
    raise last


def trace_tuple(func, y0s):
    """A TUPLE state (the reference's tests/api_tests.py:29-34: `tuple_f = lambda t, y: (f(t, y[0]), f(t, y[1]))`): every component gets
    a trace of its own, expressions of t alone live in a shared time trace and are imported on use; components must not interact.  Returns
    the list of traces if they are the SAME function of their component (equal structure, equal constants) - what rhs.PerComponent
    integrates in one launch with one error ratio per component - and raises TraceError otherwise."""
    K = len(y0s)
    shapes = [tuple(int(s) for s in y.shape) for y in y0s]
    ranks = {len(s) for s in shapes}
    if len(ranks) != 1 or len({y.dtype for y in y0s}) != 1:
        raise TraceError('tuple components of different rank / dtype')
    rank = ranks.pop()
    last = None
    for nb in range(max(rank - 1, 0), -1, -1):
        if any(_prod(s[nb:]) > MAX_ROW_DIM for s in shapes):
            break
        tt = Trace((), 0, y0s[0].dtype, y0s[0].device)
        t = Sym(tt, tt.node('t', (), None, (), False))
        trs = []
        for s in shapes:
            tr = Trace(s, nb, y0s[0].dtype, y0s[0].device)
            tr.time_trace = tt
            trs.append(tr)
        ys = tuple(Sym(tr, tr.node('y', (), 0, tr.tail, True)) for tr in trs)
        try:
            out = func(t, ys)
            if not isinstance(out, (tuple, list)) or len(out) != K:
                raise TraceError('the callable returned %s for a state of %d components' % (type(out).__name__, K))
            for tr, o, s in zip(trs, out, shapes):
                if isinstance(o, Sym) and o.tr is not tr and o.tr is not tt:
                    raise TraceError('the components of a tuple state interact (component result computed from another component)')
                _finish(tr, o, s)
            k0 = trs[0].key()
            for tr in trs[1:]:
                same = tr.key() == k0 and tr.scalars == trs[0].scalars and len(tr.tensors) == len(trs[0].tensors) and \
                    all(a['t'] is b['t'] and a['lead'] == b['lead'] for a, b in zip(tr.tensors, trs[0].tensors))
                if not same:
                    raise TraceError('the components of the tuple state follow different functions (one kernel integrates ONE function per component)')
            return trs
        except TraceError as e:
            last = e
            if 'batch axis' not in str(e) and 'batch axes' not in str(e):
                raise
    if last is None:
        raise TraceError('tuple components of more than %d elements per trajectory' % MAX_ROW_DIM)
    raise last


# ---------------------------------------------------------------------------------------------
# numpy evaluation of a trace (test infrastructure for the tracer itself: one trajectory at a time)
# ---------------------------------------------------------------------------------------------
_NP_UN = {'neg': np.negative, 'abs': np.abs, 'sin': np.sin, 'cos': np.cos, 'tan': np.tan, 'exp': np.exp, 'log': np.log, 'sqrt': np.sqrt,
          'tanh': np.tanh, 'sinh': np.sinh, 'cosh': np.cosh, 'asin': np.arcsin, 'acos': np.arccos, 'atan': np.arctan, 'log1p': np.log1p,
          'expm1': np.expm1, 'exp2': np.exp2, 'log2': np.log2, 'log10': np.log10, 'floor': np.floor, 'ceil': np.ceil,
          'sigmoid': lambda x: 1 / (1 + np.exp(-x)), 'relu': lambda x: np.where(x > 0, x, np.where(x != x, x, 0)),
          'softplus': lambda x: np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))), 'square': lambda x: x * x, 'cube': lambda x: x * x * x,
          'reciprocal': lambda x: 1 / x, 'rsqrt': lambda x: 1 / np.sqrt(x), 'rsquare': lambda x: 1 / (x * x), 'one': lambda x: np.ones_like(x),
          'sign': np.sign, 'silu': lambda x: x / (1 + np.exp(-x)), 'not': np.logical_not,
          'erf': lambda x: np.vectorize(math.erf)(x)}
_NP_BIN = {'add': np.add, 'sub': np.subtract, 'mul': np.multiply, 'div': np.divide, 'max': np.maximum, 'min': np.minimum, 'atan2': np.arctan2,
           'pow': np.power, 'gt': np.greater, 'lt': np.less, 'ge': np.greater_equal, 'le': np.less_equal, 'eq': np.equal, 'ne': np.not_equal,
           'and': np.logical_and, 'or': np.logical_or}


def evaluate_row(tr, t, y_tail):
    """f(t, y) of ONE trajectory from the graph, in numpy (float64).  y_tail: array of shape tr.tail."""
    val = {}
    for n in tr.live():
        a = [val[x.id] for x in n.args]
        if n.op == 'y':
            v = np.asarray(y_tail, dtype=np.float64).reshape(tr.tail)
        elif n.op == 't':
            v = np.float64(t)
        elif n.op in ('lit', 'litf'):
            v = np.float64(n.attr)
        elif n.op == 'par':
            v = np.float64(tr.scalars[n.attr])
        elif n.op == 'ten':
            e = tr.tensors[n.attr]
            x = e['t'].detach()
            if e['lead']:
                x = x[(0,) * e['lead']]
            v = x.to('cpu', torch.float64).numpy().reshape(n.shape)
        elif n.op == 'ew':
            fn, attr = n.attr
            with np.errstate(all='ignore'):
                if fn == 'where':
                    v = np.where(a[0], a[1], a[2])
                elif fn == 'powc':
                    v = np.power(a[0], attr)
                elif fn in _NP_UN:
                    v = _NP_UN[fn](a[0])
                else:
                    v = _NP_BIN[fn](a[0], a[1])
            v = np.broadcast_to(v, n.shape)
        elif n.op == 'gather':
            src, idx = n.attr
            flat = [np.broadcast_to(x, m.shape).reshape(-1) for x, m in zip(a, n.args)]
            if src is None:
                v = flat[0][idx]
            else:
                v = np.empty(idx.shape, dtype=flat[0].dtype)
                for k, f_ in enumerate(flat):
                    m = src == k
                    v[m] = f_[idx[m]]
        elif n.op == 'matmul':
            v = np.matmul(a[0], a[1])
        elif n.op == 'linear':
            v = np.matmul(a[0], a[1].T)
            if len(a) > 2:
                v = v + a[2]
        elif n.op == 'sum':
            axes, keep = n.attr
            v = np.sum(a[0], axis=axes, keepdims=keep)
        else:
            raise AssertionError(n.op)
        val[n.id] = v
    return np.broadcast_to(val[tr.out.id], tr.tail).astype(np.float64)


# ---------------------------------------------------------------------------------------------
# code generation: one trajectory per thread (the kernels of rhs.CustomRowLocal)
# ---------------------------------------------------------------------------------------------
MAX_ROW_DIM = 32              # rhs.CustomRowLocal.MAX_DIM: state + S + 1 stage derivatives thread-private
MAX_COOP_DIM = 256            # rhs.CustomCoop.MAX_DIM
MAX_ROW_STATEMENTS = 6000     # straight-line statements per evaluation beyond which the row-local form is not attempted
WIDE = 12                     # values wider than this are thread-private ARRAYS filled by loops (compact code: hipcc's time grows faster
                              # than linearly in the length of a straight-line block), narrower ones scalars
UNROLL_MACS = 64              # matrix products up to this many multiply-adds are written out; larger ones become loops

_C_UN = {'neg': '-{0}', 'abs': 'fabs({0})', 'sigmoid': '((T)1 / ((T)1 + exp(-{0})))', 'relu': '({0} > (T)0 ? {0} : ({0} != {0} ? {0} : (T)0))',
         'softplus': '({0} > (T)20 ? {0} : log1p(exp({0})))', 'square': '{0} * {0}', 'cube': '{0} * {0} * {0}', 'reciprocal': '(T)1 / {0}',
         'rsqrt': '(T)1 / sqrt({0})', 'rsquare': '(T)1 / ({0} * {0})', 'one': '(T)1', 'sign': '(T)(({0} > (T)0) - ({0} < (T)0))',
         'silu': '({0} / ((T)1 + exp(-{0})))', 'not': '!{0}'}
for _f in ('sin', 'cos', 'tan', 'exp', 'log', 'sqrt', 'tanh', 'sinh', 'cosh', 'asin', 'acos', 'atan', 'log1p', 'expm1', 'exp2', 'log2', 'log10', 'erf',
           'floor', 'ceil'):
    _C_UN[_f] = _f + '({0})'
_C_BIN = {'add': '{0} + {1}', 'sub': '{0} - {1}', 'mul': '{0} * {1}', 'div': '{0} / {1}', 'max': 'fmax({0}, {1})', 'min': 'fmin({0}, {1})',
          'atan2': 'atan2({0}, {1})', 'pow': 'pow({0}, {1})', 'gt': '{0} > {1}', 'lt': '{0} < {1}', 'ge': '{0} >= {1}', 'le': '{0} <= {1}',
          'eq': '{0} == {1}', 'ne': '{0} != {1}', 'and': '{0} && {1}', 'or': '{0} || {1}'}


def _flit(v):
    if v == int(v) and abs(v) < 2 ** 53:
        return '(T)%d' % int(v)
    return '(T)%r' % float(v)


class Layout(object):
    """Where the constants of a trace live: the first 8 scalars by value (`p[i]`), everything else in one device pool (`cw[...]`)."""

    def __init__(self, tr):
        self.n_scalars = len(tr.scalars)
        off = 0
        self.tensor_off = []
        for e in tr.tensors:
            self.tensor_off.append(off)
            off += _prod(e['shape'])
        self.extra_off = off
        off += max(self.n_scalars - 8, 0)
        self.size = off

    def par(self, i):
        return 'p[%d]' % i if i < 8 else 'cw[%d]' % (self.extra_off + i - 8)


class _Arr(object):
    """A node's value as a thread-private C array `name[size]` (row-major over `shape`)."""
    __slots__ = ('name', 'shape')

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)

    @property
    def size(self):
        return _prod(self.shape)


class _RowCG(object):
    """Code for one trajectory per thread.  Narrow values are scalarised: every node an array of C atoms (names / literals), every
    arithmetic operation one `const T v = ...;` statement in the order Python performed them.  Values wider than WIDE elements (the
    hidden layer of a small network) are thread-private arrays filled by loops - the same operations in the same order, compact code."""

    def __init__(self, tr):
        self.tr, self.lay = tr, Layout(tr)
        self.lines, self.cse, self.n = [], {}, 0

    def tmp(self, expr, is_bool=False):
        hit = self.cse.get(expr)
        if hit is not None:
            return hit
        name = '%s%d' % ('c' if is_bool else 'v', self.n)
        self.n += 1
        self.lines.append('const %s %s = %s;' % ('bool' if is_bool else 'T', name, expr))
        self.cse[expr] = name
        return name

    def _new(self, prefix):
        self.n += 1
        return '%s%d' % (prefix, self.n)

    def _arr(self, atoms, shape):
        a = np.empty(len(atoms), dtype=object)
        for i, s in enumerate(atoms):
            a[i] = s
        return a.reshape(shape)

    def atoms(self, v):
        """Any value as an object array of atoms."""
        if isinstance(v, _Arr):
            return self._arr(['%s[%d]' % (v.name, i) for i in range(v.size)], v.shape)
        return v

    def carray(self, atoms, is_bool=False):
        """Atoms as a C array (one declaration; the same list is declared once)."""
        key = ('arr', tuple(atoms))
        hit = self.cse.get(key)
        if hit is None:
            hit = self._new('a')
            self.lines.append('%s %s[%d] = {%s};' % ('bool' if is_bool else 'T', hit, len(atoms), ', '.join(atoms)))
            self.cse[key] = hit
        return hit

    def _ew_expr(self, n, xs):
        fn, attr = n.attr
        if fn == 'where':
            return '(%s ? %s : %s)' % tuple(xs)
        if fn == 'powc':
            return 'pow(%s, %s)' % (xs[0], _flit(attr))
        return (_C_UN[fn] if fn in _C_UN else _C_BIN[fn]).format(*xs)

    def _ew(self, n, a):
        if n.size > WIDE:
            # loop form: every operand is a scalar atom, an array of the result's own shape, or is laid out as one
            ops = []
            for v, m in zip(a, n.args):
                if m.size == 1:
                    ops.append(self.atoms(v).reshape(-1)[0])
                elif isinstance(v, _Arr) and v.shape == n.shape:
                    ops.append('%s[i_]' % v.name)
                else:
                    full = np.broadcast_to(self.atoms(v), n.shape).reshape(-1)
                    ops.append('%s[i_]' % self.carray(list(full), m.is_bool))
            out = self._new('b' if n.is_bool else 'w')
            self.lines.append('%s %s[%d];' % ('bool' if n.is_bool else 'T', out, n.size))
            self.lines.append('for (int i_ = 0; i_ < %d; ++i_) %s[i_] = %s;' % (n.size, out, self._ew_expr(n, ['(%s)' % o for o in ops])))
            return _Arr(out, n.shape)
        a = [self.atoms(v) for v in a]
        bs = np.broadcast_arrays(*a) if len(a) > 1 else [a[0]]
        flat = [b.reshape(-1) for b in bs]
        out = [self.tmp(self._ew_expr(n, [f_[i] for f_ in flat]), n.is_bool) for i in range(flat[0].size)]
        return np.broadcast_to(self._arr(out, bs[0].shape), n.shape)

    def _matvec(self, x, K, E, w_at, bias_at):
        """out[i] = sum_j x[j] * w_at(j, i) (+ bias): written out for small products (atoms), a loop nest over arrays otherwise (_Arr)."""
        if K * E <= UNROLL_MACS:
            x_atoms = list(self.atoms(x).reshape(-1))
            out = []
            for i in range(E):
                acc = None
                for j in range(K):
                    pr = self.tmp('%s * %s' % (x_atoms[j], w_at(j, i)))
                    acc = pr if acc is None else self.tmp('%s + %s' % (acc, pr))
                if bias_at is not None:
                    acc = self.tmp('%s + %s' % (acc, bias_at(i)))
                out.append(acc)
            return self._arr(out, (E,))
        xa = x.name if isinstance(x, _Arr) else self.carray(list(x.reshape(-1)))
        oa = self._new('o')
        self.lines.append('T %s[%d];' % (oa, E))
        self.lines.append('for (int i_ = 0; i_ < %d; ++i_) {' % E)
        self.lines.append('  T acc_ = %s;' % (bias_at('i_') if bias_at is not None else '(T)0'))
        self.lines.append('  for (int j_ = 0; j_ < %d; ++j_) acc_ = fma(%s[j_], %s, acc_);' % (K, xa, w_at('j_', 'i_')))
        self.lines.append('  %s[i_] = acc_;' % oa)
        self.lines.append('}')
        return _Arr(oa, (E,))

    def _const_at(self, node):
        """index -> C atom for a constant tensor node: pool offset arithmetic."""
        off = self.lay.tensor_off[node.attr]
        strides, s = [], 1
        for d in reversed(node.shape):
            strides.append(s)
            s *= d
        strides = strides[::-1]

        def at(*ix):
            if all(isinstance(i, int) for i in ix):
                return 'cw[%d]' % (off + sum(i * st for i, st in zip(ix, strides)))
            terms = [str(off)] + ['%s * %d' % (i, st) if st != 1 else str(i) for i, st in zip(ix, strides)]
            return 'cw[%s]' % ' + '.join(terms)
        return at

    def _rows(self, v, node, K):
        """The rows (last axis K) of a value: a list of atoms arrays / _Arr."""
        if isinstance(v, _Arr) and v.size == K:
            return [v]
        return list(np.broadcast_to(self.atoms(v), node.shape).reshape(-1, K))

    def _join(self, parts, shape):
        if len(parts) == 1 and isinstance(parts[0], _Arr):
            return _Arr(parts[0].name, shape)
        return self._arr([x for p_ in parts for x in self.atoms(p_).reshape(-1)], shape)

    def body(self):
        tr, val = self.tr, {}
        for n in tr.live():
            a = [val[x.id] for x in n.args]
            if n.op == 'y':
                v = self._arr(['y[%d]' % i for i in range(n.size)], n.shape)
            elif n.op == 't':
                v = self._arr(['t'], ())
            elif n.op == 'lit':
                v = self._arr(['(T)%d' % n.attr], ())
            elif n.op == 'litf':
                v = self._arr([_flit(n.attr)], ())
            elif n.op == 'par':
                v = self._arr([self.lay.par(n.attr)], ())
            elif n.op == 'ten':
                off = self.lay.tensor_off[n.attr]
                v = self._arr(['cw[%d]' % (off + i) for i in range(n.size)], n.shape)
            elif n.op == 'ew':
                v = self._ew(n, a)
            elif n.op == 'gather':
                src, idx = n.attr
                if src is None and isinstance(a[0], _Arr) and idx.size == a[0].size and np.array_equal(idx.reshape(-1), np.arange(idx.size)):
                    v = _Arr(a[0].name, idx.shape)                     # a reshape of an array: the same array
                else:
                    flat = [np.broadcast_to(self.atoms(x), m.shape).reshape(-1) for x, m in zip(a, n.args)]
                    v = np.empty(idx.size, dtype=object)
                    srcf = None if src is None else src.reshape(-1)
                    for i_, j_ in enumerate(idx.reshape(-1)):
                        v[i_] = flat[0 if srcf is None else int(srcf[i_])][int(j_)]
                    v = v.reshape(idx.shape)
            elif n.op == 'linear':
                x, w = n.args[0], n.args[1]
                E, K = w.shape
                wat = self._const_at(w)
                bat = self._const_at(n.args[2]) if len(n.args) > 2 else None
                v = self._join([self._matvec(r, K, E, lambda j, i: wat(i, j), (lambda i: bat(i)) if bat is not None else None)
                                for r in self._rows(a[0], x, K)], n.shape)
            elif n.op == 'matmul':
                v = self._matmul(n, a)
            elif n.op == 'sum':
                axes, keep = n.attr
                src = np.broadcast_to(self.atoms(a[0]), n.args[0].shape)
                moved = np.moveaxis(src, axes, tuple(range(-len(axes), 0)))
                moved = moved.reshape(moved.shape[:moved.ndim - len(axes)] + (-1,))
                out = []
                for r in moved.reshape(-1, moved.shape[-1]):
                    acc = r[0]
                    for x in r[1:]:
                        acc = self.tmp('%s + %s' % (acc, x))
                    out.append(acc)
                v = self._arr(out, n.shape)
            else:
                raise AssertionError(n.op)
            val[n.id] = v
            if len(self.lines) > MAX_ROW_STATEMENTS:
                raise TraceError('more than %d statements per evaluation in one-trajectory-per-thread form' % MAX_ROW_STATEMENTS)
        outv = np.broadcast_to(self.atoms(val[tr.out.id]), tr.tail).reshape(-1)
        for i, s in enumerate(outv):
            self.lines.append('k[%d] = %s;' % (i, s))
        return '\n'.join(self.lines)

    def _matmul(self, n, a):
        x, w = n.args
        # the two shapes a constant matrix usually appears in keep their loop form; everything else is written out by index
        if w.op == 'ten' and w.rank == 2 and x.rank >= 1:
            K, E = w.shape
            wat = self._const_at(w)
            return self._join([self._matvec(r, K, E, lambda j, i: wat(j, i), None) for r in self._rows(a[0], x, K)], n.shape)
        if x.op == 'ten' and x.rank == 2 and w.rank in (1, 2) and (w.rank == 1 or w.shape[1] == 1):
            E, K = x.shape
            xat = self._const_at(x)
            return self._join([self._matvec(self._rows(a[1], w, K)[0] if w.rank == 1 else a[1] if isinstance(a[1], _Arr) else
                                            np.broadcast_to(self.atoms(a[1]), w.shape).reshape(-1), K, E, lambda j, i: xat(i, j), None)], n.shape)
        A, B = np.broadcast_to(self.atoms(a[0]), x.shape), np.broadcast_to(self.atoms(a[1]), w.shape)
        a2 = A.reshape((1,) + A.shape) if A.ndim == 1 else A
        b2 = B.reshape(B.shape + (1,)) if B.ndim == 1 else B
        lead = np.broadcast_shapes(a2.shape[:-2], b2.shape[:-2])
        a2 = np.broadcast_to(a2, lead + a2.shape[-2:]).reshape((-1,) + a2.shape[-2:])
        b2 = np.broadcast_to(b2, lead + b2.shape[-2:]).reshape((-1,) + b2.shape[-2:])
        out = []
        for am, bm in zip(a2, b2):
            for i in range(am.shape[0]):
                for j in range(bm.shape[1]):
                    acc = None
                    for k_ in range(am.shape[1]):
                        pr = self.tmp('%s * %s' % (am[i, k_], bm[k_, j]))
                        acc = pr if acc is None else self.tmp('%s + %s' % (acc, pr))
                    out.append(acc)
        return self._arr(out, n.shape)


_ROW_TEMPLATE = """// generated by tfdiffeq_amd.lower from a traced Python callable - do not edit
#define {dtype_macro} 1
#include "mi_ode_plugin.h"
namespace mi {{
template <typename T>
struct RhsUser {{
  static constexpr int D = {dim};
  T p[8];                                                  // mi_ode_rhs.scalars: the Python floats of the callable, in the state dtype
  const T* cw;                                             // mi_ode_rhs.w[0]: the tensors it closes over (one pool, refreshed per call)
  __device__ explicit RhsUser(const RhsParams& r) : cw((const T*)r.w[0]) {{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (T)r.s[i];
  }}
  __device__ __forceinline__ void operator()(T t, const T* y, T* k) const {{
    (void)t; (void)cw;
{body}
  }}
}};
}}  // namespace mi
MI_ODE_DEFINE_ROWLOCAL_PLUGIN(mi::RhsUser)
"""

_HOST_TEMPLATE = """// host build of a generated right-hand side (tests: the same statements, compiled by g++)
#include <cmath>
using namespace std;
template <typename T> static void f_(T t, const T* y, T* k, const double* ps, const T* cw) {{
  T p[8];
  for (int i = 0; i < 8; ++i) p[i] = (T)ps[i];
  (void)t; (void)cw;
{body}
}}
extern "C" void rhs_f64(double t, const double* y, double* k, const double* ps, const double* cw) {{ f_<double>(t, y, k, ps, cw); }}
extern "C" void rhs_f32(float t, const float* y, float* k, const double* ps, const float* cw) {{ f_<float>(t, y, k, ps, cw); }}
"""


def rowlocal_body(tr):
    return _RowCG(tr).body()


def host_source(tr):
    """The generated statements as a host function (the CPU tests compile it with g++ and compare with the callable itself)."""
    body = '\n'.join('  ' + ln for ln in rowlocal_body(tr).splitlines())
    return _HOST_TEMPLATE.format(body=body)


def _indent(body, n):
    return '\n'.join(' ' * n + ln for ln in body.splitlines())


# ---------------------------------------------------------------------------------------------
# code generation: one state ELEMENT per thread (the kernels of rhs.CustomCoop) - systems of dimension 33 .. 256
# ---------------------------------------------------------------------------------------------
class _CoopCG(object):
    """Every node that is not a pure function of constants becomes an array of `w` values per trajectory in LDS, filled by the
    workgroup's threads together (units dealt to all 256 threads, a barrier between dependent nodes); elementwise chains are folded
    into the expression of the node that consumes them.  Gathers read through `static const` index tables."""

    LDS_BYTES = 32 * 1024

    def __init__(self, tr, dim):
        self.tr, self.lay, self.dim = tr, Layout(tr), dim
        self.tables, self.decl, self.code = [], [], []
        self.mat = {}                     # node id -> (lds name, width) for materialised nodes
        self.words = 0

    def _table(self, arr):
        name = 'g%d' % len(self.tables)
        flat = [int(v) for v in np.asarray(arr).reshape(-1)]
        self.tables.append('static __device__ const short %s[%d] = {%s};' % (name, len(flat), ', '.join(map(str, flat))))
        return name

    def _expr(self, n, e):
        """C expression for flat element `e` (a C expression string) of node n, for the trajectory slot `sl`."""
        tr = self.tr
        if n.id in self.mat:
            name, w = self.mat[n.id]
            return '%s[sl * %d + %s]' % (name, w, e) if w > 1 else '%s[sl]' % name
        if n.op == 't':
            return 't'
        if n.op == 'lit':
            return '(T)%d' % n.attr
        if n.op == 'litf':
            return _flit(n.attr)
        if n.op == 'par':
            return self.lay.par(n.attr)
        if n.op == 'ten':
            off = self.lay.tensor_off[n.attr]
            return 'cw[%d + %s]' % (off, e) if n.size > 1 else 'cw[%d]' % off
        if n.op == 'ew':
            fn, attr = n.attr
            xs = [self._expr(a, self._bidx(a, n, e)) for a in n.args]
            if fn == 'where':
                return '(%s ? %s : %s)' % tuple(xs)
            if fn == 'powc':
                return 'pow(%s, %s)' % (xs[0], _flit(attr))
            xs = ['(%s)' % x for x in xs]
            return '(' + (_C_UN[fn] if fn in _C_UN else _C_BIN[fn]).format(*xs) + ')'
        raise AssertionError('node %r must be materialised' % (n,))

    def _bidx(self, a, n, e):
        """Flat index into `a` for flat element e of the broadcast result n (a table unless the two coincide)."""
        if a.size == 1:
            return '0'
        if a.shape == n.shape:
            return e
        idx = np.broadcast_to(np.arange(a.size, dtype=np.int64).reshape(a.shape), n.shape)
        return '%s[%s]' % (self._table(idx), e)

    def _materialise(self, n, per_elem, needs_barrier=True):
        name, w = 's%d' % n.id, max(n.size, 1)
        self.mat[n.id] = (name, w)
        self.words += w
        self.decl.append((name, w))
        self.code.append('for (int u = (int)threadIdx.x; u < TPW * %d; u += 256) {' % w)
        self.code.append('  const int sl = u / %d, e = u - sl * %d; (void)e;' % (w, w))
        for ln in per_elem:
            self.code.append('  ' + ln)
        self.code.append('}')
        self.code.append('__syncthreads();')

    def body(self):
        tr = self.tr
        live = tr.live()
        users = {}
        for n in live:
            for a in n.args:
                users.setdefault(a.id, []).append(n)
        ynode = [n for n in live if n.op == 'y']
        if ynode:
            self.mat[ynode[0].id] = ('s_y', self.dim)
        for n in live:
            if n.op in ('y', 't', 'lit', 'litf', 'par', 'ten'):
                continue
            if n.op == 'ew':
                # folded into its consumers unless a gather / product / reduction / the output reads it at other positions
                if all(u.op == 'ew' for u in users.get(n.id, [])) and n is not tr.out:
                    continue
                self._materialise(n, ['%s[u] = %s;' % ('s%d' % n.id, self._expr_ew_inline(n))])
            elif n.op == 'gather':
                src, idx = n.attr
                tab = self._table(idx)
                if src is None:
                    a = n.args[0]
                    self._materialise(n, ['s%d[u] = %s;' % (n.id, self._expr(a, '%s[e]' % tab if a.size > 1 else '0'))])
                else:
                    stab = self._table(src)
                    lines = ['T r_ = (T)0;', 'switch (%s[e]) {' % stab]
                    for k, a in enumerate(n.args):
                        lines.append('  case %d: r_ = %s; break;' % (k, self._expr(a, '%s[e]' % tab if a.size > 1 else '0')))
                    lines += ['}', 's%d[u] = r_;' % n.id]
                    self._materialise(n, lines)
            elif n.op in ('linear', 'matmul'):
                self._product(n)
            elif n.op == 'sum':
                axes, keep = n.attr
                a = n.args[0]
                idx = np.arange(a.size).reshape(a.shape)
                moved = np.moveaxis(idx, axes, tuple(range(-len(axes), 0))).reshape(max(n.size, 1), -1)
                tab = self._table(moved)
                m = moved.shape[1]
                self._force(a)
                self._materialise(n, ['T acc_ = %s;' % self._expr(a, '%s[e * %d]' % (tab, m)),
                                      'for (int j_ = 1; j_ < %d; ++j_) acc_ = acc_ + %s;' % (m, self._expr(a, '%s[e * %d + j_]' % (tab, m))),
                                      's%d[u] = acc_;' % n.id])
            else:
                raise AssertionError(n.op)
        out = tr.out
        self._force(out)
        return out

    def _expr_ew_inline(self, n):
        fn, attr = n.attr
        saved = self.mat.pop(n.id, None)
        try:
            return self._expr(n, 'e')
        finally:
            if saved is not None:
                self.mat[n.id] = saved

    def _force(self, n):
        """Make sure n can be read at arbitrary positions (materialise a folded elementwise node on demand)."""
        if n.id in self.mat or n.op in ('t', 'lit', 'litf', 'par', 'ten'):
            return
        if n.op == 'ew':
            self._materialise(n, ['s%d[u] = %s;' % (n.id, self._expr_ew_inline(n))])
            return
        raise AssertionError(n)

    def _product(self, n):
        if n.op == 'linear':
            x, w = n.args[0], n.args[1]
            E, K = w.shape
            woff = self.lay.tensor_off[w.attr]
            wexpr = 'cw + %d + (long long)i_ * %d' % (woff, K)          # row i of [E, K]: contiguous in j
            ld = 1
            bias = 'cw[%d + i_]' % self.lay.tensor_off[n.args[2].attr] if len(n.args) > 2 else '(T)0'
        else:
            x, w = n.args
            if w.op == 'ten' and w.rank == 2 and x.rank >= 1:
                K, E = w.shape
                wexpr, ld, bias = 'cw + %d + i_' % self.lay.tensor_off[w.attr], E, '(T)0'
            elif x.op == 'ten' and x.rank == 2 and (w.rank == 1 or (w.rank == 2 and w.shape[1] == 1)):
                E, K = x.shape
                wexpr, ld, bias = 'cw + %d + (long long)i_ * %d' % (self.lay.tensor_off[x.attr], K), 1, '(T)0'
                x = w
            else:
                raise TraceError('a matrix product of two traced values in a system of more than %d elements' % MAX_ROW_DIM)
        self._force(x)
        rows = max(x.size // K, 1)
        xs = self.mat.get(x.id)
        if xs is None:
            raise TraceError('matrix product of a constant')
        xname, xw = xs
        self._materialise(n, ['const int r_ = e / %d, i_ = e - r_ * %d;' % (E, E),
                              's%d[u] = coop_dot_col<T>(%s + sl * %d + r_ * %d, %s, %d, %d, %s);' % (n.id, xname, xw, K, wexpr, K, ld, bias)])


_COOP_TEMPLATE = """// generated by tfdiffeq_amd.lower from a traced Python callable (a thread per state element) - do not edit
#define {dtype_macro} 1
#include "mi_ode_plugin.h"
namespace mi {{
{tables}
template <typename T>
struct RhsUserCoop {{
  static constexpr int D = 1;
  static constexpr bool kCoop = true;
  static constexpr int DIM = {dim};
  static constexpr int TPW = {tpw};                        // trajectories per 256-thread workgroup
  T p[8];
  const T* cw;
  __device__ explicit RhsUserCoop(const RhsParams& r) : cw((const T*)r.w[0]) {{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (T)r.s[i];
  }}
  static __host__ __device__ int tpw(const RhsParams&, int) {{ return TPW; }}
  __device__ __forceinline__ void operator()(T t, const T* yv, T* kv) const {{
    __shared__ T s_y[TPW * DIM];
{decl}
    const int slot = (int)threadIdx.x / DIM, col = (int)threadIdx.x - slot * DIM;
    (void)t; (void)cw;
    __syncthreads();                                       // the previous evaluation's readers are done
    if (slot < TPW) s_y[threadIdx.x] = yv[0];
    __syncthreads();
{body}
    kv[0] = slot < TPW ? {out}[slot * DIM + col] : (T)0;
  }}
}};
}}  // namespace mi
MI_ODE_DEFINE_COOP_PLUGIN(mi::RhsUserCoop)
"""


def coop_source(tr, dtype):
    dim = _prod(tr.tail)
    cg = _CoopCG(tr, dim)
    out = cg.body()
    esz = 4 if dtype == torch.float32 else 8
    words = cg.words + dim
    tpw = max(min(256 // dim, _CoopCG.LDS_BYTES // (esz * words)), 0)
    if tpw < 1:
        raise TraceError('the intermediate values of one trajectory (%d) do not fit the workgroup\'s LDS' % words)
    name, _w = cg.mat[out.id]
    return _COOP_TEMPLATE.format(dtype_macro='MI_ODE_PLUGIN_F32' if dtype == torch.float32 else 'MI_ODE_PLUGIN_F64', dim=dim, tpw=tpw,
                                 tables='\n'.join(cg.tables), decl='\n'.join('    __shared__ T %s[TPW * %d];' % d for d in cg.decl),
                                 body=_indent('\n'.join(cg.code), 4), out=name)


# ---------------------------------------------------------------------------------------------
# classification
# ---------------------------------------------------------------------------------------------
def _see_through(n):
    """Skip gathers that keep the flat element order (reshape, unsqueeze, [..., 0] of a trailing axis of 1)."""
    while n.op == 'gather' and n.attr[0] is None and len(n.args) == 1 and n.args[0].size == n.size and \
            np.array_equal(n.attr[1].reshape(-1), np.arange(n.size)):
        n = n.args[0]
    return n


def _affine(n):
    """(source node, ('W' | 'Wt', tensor index), bias tensor index or None, K, E) if n is `x @ W (+ b)` of one vector, else None.
    'W': the constant is [K, E] (in, out); 'Wt': it is [E, K]."""
    n = _see_through(n)
    if n.op == 'linear':
        x, w = n.args[0], n.args[1]
        if _see_through(x).size != w.shape[1] or w.op != 'ten':
            return None
        b = n.args[2] if len(n.args) > 2 else None
        if b is not None and b.op != 'ten':
            return None
        return _see_through(x), ('Wt', w.attr), None if b is None else b.attr, w.shape[1], w.shape[0]
    if n.op == 'matmul':
        x, w = n.args
        if w.op == 'ten' and w.rank == 2 and _see_through(x).size == w.shape[0]:
            return _see_through(x), ('W', w.attr), None, w.shape[0], w.shape[1]
        if x.op == 'ten' and x.rank == 2 and _see_through(w).size == x.shape[1]:
            return _see_through(w), ('Wt', x.attr), None, x.shape[1], x.shape[0]
        return None
    if n.op == 'ew' and n.attr[0] == 'add':
        for lin, c in (n.args, n.args[::-1]):
            got = _affine(lin)
            c = _see_through(c)
            if got is not None and got[2] is None and c.op == 'ten' and c.size == got[4]:
                return got[0], got[1], c.attr, got[3], got[4]
    return None


def _activation(n):
    n = _see_through(n)
    if n.op == 'ew' and n.attr[0] in ('tanh', 'relu', 'softplus') and len(n.args) == 1:
        return n.attr[0], n.args[0]
    return None


def graph_macs(tr):
    """Multiply-adds of the matrix products of ONE evaluation of one trajectory."""
    macs = 0
    for n in tr.live():
        if n.op == 'linear':
            macs += (n.args[0].size // max(n.args[1].shape[1], 1)) * n.args[1].shape[0] * n.args[1].shape[1]
        elif n.op == 'matmul':
            a, b = n.args
            k = a.shape[-1]
            macs += n.size * k
    return macs


COOP_MAX_FMA = 5e7            # generated cooperative code is vector-ALU work fed from L2 (~1e12 fma/s, rhs.MLP.COOP_MAX_FMA): beyond this many
                              # multiply-adds per evaluation of the whole batch the callable engine's rocBLAS products are the faster route
ROW_MAX_MACS = 8192           # ... and a thread that owns a trajectory does its products alone


def classify(tr, generic=False, rows=1):
    """('linear' | 'cubic' | 'mlp' | 'rowlocal' | 'coop', details).  generic: generated code only (a method the catalogue families have
    no kernel for - adaptive_heun exists on the row-local / cooperative kernels alone).  rows: trajectories of this call (the cost guard
    of the generated kinds)."""
    dim = _prod(tr.tail)
    ynode = next((n for n in tr.nodes if n.op == 'y'), None)
    aff = None if generic else _affine(tr.out)
    if aff is not None and dim >= 5 and aff[3] == aff[4] == dim:
        src = aff[0]
        if src is ynode:
            return 'linear', {'W': aff[1], 'b': aff[2]}
        if src.op == 'ew' and src.attr[0] == 'cube' and _see_through(src.args[0]) is ynode and aff[2] is None:
            return 'cubic', {'W': aff[1]}
    if aff is not None and aff[4] == dim:
        # Linear, act, Linear[, act, Linear] straight on y
        layers, cur, acts = [aff], aff[0], []
        while len(layers) < 3:
            act = _activation(cur)
            if act is None:
                break
            prev = _affine(act[1])
            if prev is None:
                break
            acts.append(act[0])
            layers.append(prev)
            cur = prev[0]
        layers = layers[::-1]
        if cur is ynode and len(layers) in (2, 3) and len(set(acts)) == 1 and layers[0][3] == dim:
            hid = layers[0][4]
            ok3 = len(layers) == 3 and layers[1][3] == hid and layers[1][4] == hid and layers[2][3] == hid
            ok2 = len(layers) == 2 and layers[1][3] == hid and acts[0] == 'relu'
            if (ok3 or ok2) and (dim > 4 or hid > 16):
                return 'mlp', {'layers': [(l_[1], l_[2]) for l_ in layers], 'act': acts[0], 'hidden': hid}
    macs = graph_macs(tr)
    if dim <= MAX_ROW_DIM and macs <= ROW_MAX_MACS:
        return 'rowlocal', {}
    if dim <= MAX_COOP_DIM:
        if macs * rows > COOP_MAX_FMA:
            raise TraceError('%d multiply-adds per trajectory x %d trajectories per evaluation: generated cooperative code (vector ALU, weights from L2) '
                             'would be slower than the callable engine\'s rocBLAS products' % (macs, rows))
        return 'coop', {}
    raise TraceError('a system of %d elements per trajectory with no matrix structure the catalogue knows' % dim)


# ---------------------------------------------------------------------------------------------
# programs: the device side of a traced callable, cached by structure
# ---------------------------------------------------------------------------------------------
class _GeneratedRHS(R.DeviceRHS):
    """Generated device code behind the plugin ABI: `pool` carries the tensors (and the scalars beyond the eighth)."""
    kind = N.RHS_PLUGIN
    fixed_grid_fused = True

    def __init__(self, dim, sources, coop):
        super(_GeneratedRHS, self).__init__()
        self.dim = int(dim)
        self._sources = sources              # dtype -> source text
        self.row_local = not coop
        self.coop = bool(coop)
        if coop:
            self.wide_tableaus = True
        self.params = []
        self.pool = None
        self.torch_fn = None
        self._plugins = {}

    @property
    def multistep_fused(self):
        return True

    def forward(self, t, y):
        return self.torch_fn(t, y)

    def source(self, dtype):
        return self._sources[dtype]

    _plugin = R.CustomRowLocal._plugin

    def fill(self, rhs, dtype, device):
        keep = super(_GeneratedRHS, self).fill(rhs, dtype, device)
        lib, table = self._plugin(dtype)
        rhs.plugin = table
        for i, v in enumerate(self.params[:8]):
            rhs.scalars[i] = v
        if self.pool is not None:
            rhs.w[0] = self.pool.data_ptr()
            keep.append(self.pool)
        keep.append(lib)
        return keep

    def cache_key(self, dtype, device):
        return super(_GeneratedRHS, self).cache_key(dtype, device) + (self._plugin(dtype)[1],)


class Program(object):
    """What a structure key maps to: the classification, the DeviceRHS that carries it and its persistent device buffers."""

    def __init__(self, tr, generic=False):
        self.key = tr.key() + ('g' if generic else '')
        self.kind, self.info = classify(tr, generic)
        self.macs = graph_macs(tr)
        self.dim = _prod(tr.tail)
        self.dtype = tr.dtype
        self.layout = Layout(tr)
        self.rhs = None
        self.source = None
        self._bufs = {}
        self._extra = None
        if self.kind == 'rowlocal':
            body = _indent(rowlocal_body(tr), 4)
            self.source = _ROW_TEMPLATE.format(dtype_macro='MI_ODE_PLUGIN_F32' if tr.dtype == torch.float32 else 'MI_ODE_PLUGIN_F64',
                                               dim=self.dim, body=body)
            self.rhs = _GeneratedRHS(self.dim, {tr.dtype: self.source}, coop=False)
        elif self.kind == 'coop':
            self.source = coop_source(tr, tr.dtype)
            self.rhs = _GeneratedRHS(self.dim, {tr.dtype: self.source}, coop=True)

    def _buf(self, name, shape, device):
        b = self._bufs.get((name, str(device)))
        if b is None or tuple(b.shape) != tuple(shape):
            b = torch.zeros(shape, dtype=self.dtype, device=device)
            self._bufs[(name, str(device))] = b
        return b

    @staticmethod
    def _const(tr, idx):
        e = tr.tensors[idx]
        x = e['t'].detach()
        return x[(0,) * e['lead']] if e['lead'] else x

    def _mat(self, tr, spec, name, device):
        """A persistent [in, out] copy of a traced matrix constant."""
        how, idx = spec
        x = self._const(tr, idx)
        x = x if how == 'W' else x.t()
        buf = self._buf(name, tuple(x.shape), device)
        buf.copy_(x)
        return buf

    def bind(self, tr, device):
        """Copy THIS call's constants into the program's buffers; returns the DeviceRHS to integrate."""
        device = torch.device(device)
        with torch.no_grad():
            if self.kind in ('linear', 'cubic'):
                W = self._mat(tr, self.info['W'], 'W', device)
                b = None
                if self.info.get('b') is not None:
                    b = self._buf('b', (self.dim,), device)
                    b.copy_(self._const(tr, self.info['b']).reshape(-1))
                if self.rhs is None or self.rhs.W is not W or (b is None) != (self.rhs.b is None) or (b is not None and self.rhs.b is not b):
                    self.rhs = R.Linear(W, b) if self.kind == 'linear' else R.CubicLinear(W)
                return self.rhs
            if self.kind == 'mlp':
                layers, hid = self.info['layers'], self.info['hidden']
                Ws = [self._mat(tr, l_[0], 'W%d' % i, device) for i, l_ in enumerate(layers)]
                bs = []
                for i, l_ in enumerate(layers):
                    if l_[1] is None:
                        bs.append(None)
                    else:
                        b = self._buf('b%d' % i, (Ws[i].shape[1],), device)
                        b.copy_(self._const(tr, l_[1]).reshape(-1))
                        bs.append(b)
                if len(layers) == 2:                       # relu(relu(z)) = relu(z): an identity middle layer (rhs.from_sequential)
                    eye = self._bufs.get(('eye', str(device)))
                    if eye is None:
                        eye = torch.eye(hid, dtype=self.dtype, device=device)
                        self._bufs[('eye', str(device))] = eye
                    Ws, bs = [Ws[0], eye, Ws[1]], [bs[0], None, bs[1]]
                if self.rhs is None or any(a is not b_ for a, b_ in zip(self.rhs.Ws, Ws)) or \
                        any((a is None) != (b_ is None) or (a is not None and a is not b_) for a, b_ in zip(self.rhs.bs, bs)):
                    self.rhs = R.MLP(Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2], activation=self.info['act'])
                return self.rhs
            # generated code: scalars by value, everything else through the pool
            lay = self.layout
            pool = self._buf('pool', (max(lay.size, 1),), device)
            for idx, off in enumerate(lay.tensor_off):
                n = _prod(tr.tensors[idx]['shape'])
                if n:
                    pool[off:off + n].view(tr.tensors[idx]['shape']).copy_(self._const(tr, idx).reshape(tr.tensors[idx]['shape']))
            extra = [float(v) for v in tr.scalars[8:]]
            if extra and extra != self._extra:
                pool[lay.extra_off:lay.extra_off + len(extra)].copy_(torch.tensor(extra, dtype=torch.float64).to(self.dtype))
                self._extra = extra
            self.rhs.params = [float(v) for v in tr.scalars[:8]]
            self.rhs.pool = pool
            return self.rhs


_PROGRAMS = {}
_PROGRAMS_MAX = 64


def program_for(tr, generic=False):
    key = tr.key() + ('g' if generic else '')
    prog = _PROGRAMS.get(key)
    if prog is None:
        prog = Program(tr, generic)
        while len(_PROGRAMS) >= _PROGRAMS_MAX:
            _PROGRAMS.pop(next(iter(_PROGRAMS)))
        _PROGRAMS[key] = prog
    return prog


# ---------------------------------------------------------------------------------------------
# the callable's own Python state (is it a pure function of t and y?)
# ---------------------------------------------------------------------------------------------
def _scalar_state(obj, depth=0):
    out = {}
    d = getattr(obj, '__dict__', None)
    if isinstance(d, dict):
        for k, v in d.items():
            if isinstance(v, (int, float, bool, str)) or v is None:
                out[('a', k)] = v
            elif depth == 0 and isinstance(v, (list, dict)) and len(v) <= 16:
                vals = list(v.values()) if isinstance(v, dict) else v
                if all(isinstance(x, (int, float, bool, str)) or x is None for x in vals):
                    out[('c', k)] = tuple(vals)
    return out


def fingerprint(func):
    """Shallow snapshot of the scalar Python state a callable could be counting in: attributes of the callable / its bound object,
    closure cells holding numbers or small lists of numbers."""
    fp = {}
    for tag, obj in (('f', func), ('s', getattr(func, '__self__', None))):
        if obj is not None:
            for k, v in _scalar_state(obj).items():
                fp[(tag,) + k] = v
    fn = getattr(func, '__func__', func)
    for i, cell in enumerate(getattr(fn, '__closure__', None) or ()):
        try:
            v = cell.cell_contents
        except ValueError:
            continue
        if isinstance(v, (int, float, bool)):
            fp[('cell', i)] = v
        elif isinstance(v, (list, dict)) and len(v) <= 16:
            vals = list(v.values()) if isinstance(v, dict) else v
            if all(isinstance(x, (int, float, bool)) for x in vals):
                fp[('cell', i)] = tuple(vals)
    return fp


def _restore_nfe(func, before, after):
    """True if the only difference is an integer `nfe` counter (put back: the probe is not one of the solver's evaluations)."""
    diff = [k for k in set(before) | set(after) if before.get(k) != after.get(k)]
    if not diff:
        return True
    for k in diff:
        if not (len(k) == 3 and k[1] == 'a' and k[2] == 'nfe' and isinstance(before.get(k), int) and isinstance(after.get(k), int)):
            return False
    for k in diff:
        obj = {'f': func, 's': getattr(func, '__self__', None)}[k[0]]
        try:
            setattr(obj, 'nfe', before[k])
        except Exception:
            return False
    return True


# ---------------------------------------------------------------------------------------------
# entry point
# ---------------------------------------------------------------------------------------------
class Lowered(object):
    """A callable lowered for one call: `rhs` integrates the state reshaped to `state_shape` ([*batch, dim])."""

    def __init__(self, prog, rhs, tr, func):
        self.program, self.rhs, self.trace = prog, rhs, tr
        self.kind = prog.kind
        self.dim = prog.dim
        self.state_shape = tr.batch_shape + (prog.dim,)
        self.full_shape = tr.full_shape
        self.py_calls = 0
        self.per_component = False
        full, state = tr.full_shape, self.state_shape

        def torch_fn(t, y, _f=func):
            # the callable itself, for the paths that need one (midpoint / heun, a batch no one-launch kernel takes)
            self.py_calls += 1
            return _f(t, y.reshape(full)).reshape(state)
        self.torch_fn = torch_fn

    def describe(self):
        return {'kind': self.kind, 'dim': self.dim, 'batch_axes': self.trace.nb, 'graph_nodes': len(self.trace.live()),
                'scalars': len(self.trace.scalars), 'tensors': len(self.trace.tensors), 'program': self.program.key}


TRACE_CACHE = True            # reuse a trace while nothing the callable can name has changed (see _cached_trace)
GENERIC_ONLY_METHODS = ('adaptive_heun',)       # tableaus only the row-local / cooperative kernels are instantiated for


# ---------------------------------------------------------------------------------------------
# trace cache: a callable is traced again only when something it can NAME has changed
# ---------------------------------------------------------------------------------------------
# Tracing costs 0.3 - 0.5 ms of Python per call - more than the kernel of a short integration.  A trace may be reused when a repeat of the
# evaluation would record the same graph with the same constants.  That is decided without running the callable: `_snapshot` walks what
# the callable can name (closure cells, its bound object, its own attributes, the globals and constants its code mentions, defaults;
# through containers, plain objects and nn.Modules, a few levels deep) and records every number BY VALUE and every tensor / array BY
# IDENTITY.  A trace is cacheable only if EVERY constant it used is one of those named objects (a float computed inside the callable -
# `2 * self.a` - is not, and such a callable is simply traced on every call); a later call reuses it when the snapshot is equal.
_SNAP_MAX = 512


def _snapshot(func):
    """(entries, objects): entries - a tuple of (path, kind, value | id) for every number / tensor the callable can name, or None if
    there are too many; objects - the ids of those objects (to decide whether a trace is explained by them)."""
    import functools
    entries, ids, seen = [], set(), set()

    def leaf(path, obj):
        if isinstance(obj, (bool, int, float)) or obj is None:
            entries.append((path, 'n', obj if obj is None else (type(obj).__name__, obj)))
            ids.add(id(obj))
            return True
        if isinstance(obj, (np.floating, np.integer, np.bool_)):
            entries.append((path, 'n', (type(obj).__name__, obj.item())))
            ids.add(id(obj))
            return True
        if isinstance(obj, (torch.Tensor, np.ndarray)):
            entries.append((path, 't', id(obj)))
            ids.add(id(obj))
            return True
        return False

    def visit(path, obj, depth):
        if len(entries) > _SNAP_MAX:
            return
        if leaf(path, obj):
            return
        if isinstance(obj, (str, bytes, type)) or id(obj) in seen or depth > 3:
            return
        seen.add(id(obj))
        if isinstance(obj, torch.nn.Module):
            entries.append((path, 'm', (id(obj), obj.training)))
            for k, v in obj._parameters.items():
                visit(path + ('p', k), v, depth)
            for k, v in obj._buffers.items():
                visit(path + ('b', k), v, depth)
            for k, v in obj._modules.items():
                visit(path + ('m', k), v, depth)
            for k, v in obj.__dict__.items():
                if k[:1] != '_' or k in ('_mi_extra_params',):
                    visit(path + ('a', k), v, depth + 1)
            return
        if isinstance(obj, (list, tuple)):
            entries.append((path, 'l', len(obj)))
            for i, v in enumerate(obj[:64]):
                visit(path + (i,), v, depth + 1)
            return
        if isinstance(obj, dict):
            entries.append((path, 'l', len(obj)))
            for k, v in list(obj.items())[:64]:
                if isinstance(k, (str, int, float, bool)):
                    visit(path + ('k', k), v, depth + 1)
            return
        if isinstance(obj, functools.partial):
            visit(path + ('pf',), obj.func, depth)
            visit(path + ('pa',), obj.args, depth)
            visit(path + ('pk',), obj.keywords, depth)
            return
        fn = getattr(obj, '__func__', None)
        if fn is not None:                                   # bound method
            visit(path + ('f',), fn, depth)
            visit(path + ('s',), getattr(obj, '__self__', None), depth)
            return
        code = getattr(obj, '__code__', None)
        if code is not None:                                 # a function: closure, defaults, the globals and constants its code mentions
            entries.append((path, 'c', id(code)))
            for c in code.co_consts:
                if isinstance(c, (int, float, bool)):
                    ids.add(id(c))
            for i, cell in enumerate(obj.__closure__ or ()):
                try:
                    visit(path + ('c', i), cell.cell_contents, depth + 1)
                except ValueError:
                    pass
            for i, v in enumerate(obj.__defaults__ or ()):
                visit(path + ('d', i), v, depth + 1)
            for k, v in (obj.__kwdefaults__ or {}).items():
                visit(path + ('kd', k), v, depth + 1)
            g = getattr(obj, '__globals__', {})
            for name in code.co_names:
                if name in g and not isinstance(g[name], type(torch)):
                    visit(path + ('g', name), g[name], depth + 1)
            return
        call = getattr(type(obj), '__call__', None)
        if call is not None and hasattr(call, '__code__'):
            visit(path + ('call',), call, depth)
        d = getattr(obj, '__dict__', None)
        if isinstance(d, dict):
            entries.append((path, 'o', id(type(obj))))
            for k, v in list(d.items())[:64]:
                visit(path + ('a', k), v, depth + 1)
    visit((), func, 0)
    if len(entries) > _SNAP_MAX:
        return None, ids
    return tuple(entries), ids


_TRACES = {}
_TRACES_MAX = 64
trace_cache_stats = {'hits': 0, 'misses': 0, 'uncacheable': 0}


def _cached_trace(func, y0, nb):
    """(trace, hit) - see the comment above.  The entry keeps the trace (and through it the tensors it refers to) alive."""
    fn = getattr(func, '__func__', func)
    code = getattr(fn, '__code__', None) or getattr(getattr(type(func), '__call__', None), '__code__', None)
    if code is None:
        return None, False
    key = (id(code), tuple(y0.shape), y0.dtype, str(y0.device), nb)
    snap, ids = _snapshot(func)
    ent = _TRACES.get(key)
    if ent is not None and snap is not None and ent[0] is code and ent[1] == snap:
        trace_cache_stats['hits'] += 1
        return ent[2], True
    before = fingerprint(func)
    tr = trace(func, y0, nb=nb)
    pure = _restore_nfe(func, before, fingerprint(func))
    if not pure:
        raise TraceError('the callable changed its own Python state while it was traced: not a pure function of (t, y)')
    explained = snap is not None and all(id(o) in ids for o in tr._keep)
    if explained:
        snap2, _ = _snapshot(func)                           # (taken again: the evaluation itself must not have changed what it names)
        explained = snap2 == snap
    if explained:
        trace_cache_stats['misses'] += 1
        while len(_TRACES) >= _TRACES_MAX:
            _TRACES.pop(next(iter(_TRACES)))
        _TRACES[key] = (code, snap, tr)
    else:
        trace_cache_stats['uncacheable'] += 1
        _TRACES.pop(key, None)
    return tr, False


def lower(func, y0, nb=None, method=None):
    """Trace `func` for a state like y0 and bind this call's constants; raises TraceError with the reason when it cannot be lowered."""
    if isinstance(func, CompiledCallable):
        return func.lowered(y0, method)
    tr, _hit = _cached_trace(func, y0, nb) if TRACE_CACHE else (None, False)
    if tr is None:
        before = fingerprint(func)
        try:
            tr = trace(func, y0, nb=nb)
        finally:
            after = fingerprint(func)
            pure = _restore_nfe(func, before, after)
        if not pure:
            changed = sorted(str(k[-1]) for k in set(before) | set(after) if before.get(k) != after.get(k))
            raise TraceError('the callable changed its own Python state while it was traced (%s): not a pure function of (t, y)' % ', '.join(changed))
    rows = _prod(tr.batch_shape)
    classify(tr, generic=method in GENERIC_ONLY_METHODS, rows=rows)        # (the cost guard of generated cooperative code: raises)
    prog = program_for(tr, generic=method in GENERIC_ONLY_METHODS)
    rhs = prog.bind(tr, y0.device)
    low = Lowered(prog, rhs, tr, func)
    rhs.forward = low.torch_fn                  # (instance attribute: THIS call's callable, whatever the catalogue class computes itself)
    return low


class CompiledCallable(object):
    """`compile(func, y0)`: a callable traced ONCE.  `odeint(compiled, y0, t)` then skips the per-call trace (0.3 - 0.5 ms of Python for a
    small system - what a two-attempt call in a training loop cannot afford): the tensors `func` closes over are still re-read on every
    call (by reference: in-place updates and optimizer steps are seen), Python numbers are FROZEN at their values at compile time - the
    caller's promise.  A state of another shape / dtype is traced afresh.  It is still `func` wherever a Python callable is needed."""

    def __init__(self, func, y0, nb=None):
        self.func = func
        self._nb = nb
        self._traces = {}
        self._trace_for(y0, None)

    def __call__(self, t, y):
        return self.func(t, y)

    def _trace_for(self, y0, method):
        key = (tuple(y0.shape), y0.dtype, method in GENERIC_ONLY_METHODS)
        tr = self._traces.get(key)
        if tr is None:
            before = fingerprint(self.func)
            tr = trace(self.func, y0, nb=self._nb)
            if not _restore_nfe(self.func, before, fingerprint(self.func)):
                raise TraceError('the callable changed its own Python state while it was traced: not a pure function of (t, y)')
            self._traces[key] = tr
        return tr

    def lowered(self, y0, method=None):
        tr = self._trace_for(y0, method)
        if tr.device != y0.device:
            tr.device = y0.device
        classify(tr, generic=method in GENERIC_ONLY_METHODS, rows=_prod(tr.batch_shape))
        prog = program_for(tr, generic=method in GENERIC_ONLY_METHODS)
        rhs = prog.bind(tr, y0.device)
        low = Lowered(prog, rhs, tr, self.func)
        rhs.forward = low.torch_fn
        return low


def compile(func, y0, nb=None):                  # noqa: A001  (the name users expect)
    """Trace `func` once for states like y0; see CompiledCallable."""
    return CompiledCallable(func, y0, nb=nb)


def lower_tuple(func, y0s, method=None):
    """`lower` for a tuple state of K >= 2 components that all follow the same trajectory-local function: a `Lowered` whose `rhs` is the
    generated right-hand side of ONE component (to be lifted with rhs.PerComponent) and whose `state_shapes` reshape every component."""
    before = fingerprint(func)
    try:
        trs = trace_tuple(func, list(y0s))
    finally:
        pure = _restore_nfe(func, before, fingerprint(func))
    if not pure:
        raise TraceError('the callable changed its own Python state while it was traced: not a pure function of (t, y)')
    tr = trs[0]
    generic = True                                           # (PerComponent integrates row-local right-hand sides: generated code only)
    classify(tr, generic=generic, rows=max(_prod(t_.batch_shape) for t_ in trs))
    prog = program_for(tr, generic=generic)
    if prog.kind != 'rowlocal':
        raise TraceError('tuple states run in one launch for trajectory-local systems of up to %d elements per component' % MAX_ROW_DIM)
    rhs = prog.bind(tr, y0s[0].device)
    K = len(trs)

    def one_component(t_, yk, _f=func):                      # the same function of ONE component (verified above), for the callable paths
        return _f(t_, (yk,) * K)[0]
    low = Lowered(prog, rhs, tr, one_component)
    low.per_component = True
    low.state_shapes = [t_.batch_shape + (prog.dim,) for t_ in trs]
    low.full_shapes = [t_.full_shape for t_ in trs]
    rhs.forward = low.torch_fn
    return low


def sources_for(func, y0, nb=None, method=None):
    """The generated source(s) a call with this callable and state would compile (build-time prebuilding; [] for catalogue routes)."""
    if isinstance(y0, (tuple, list)):
        tr = trace_tuple(func, list(y0))[0]
        prog = program_for(tr, generic=True)
    else:
        tr = trace(func, y0, nb=nb)
        prog = program_for(tr, generic=method in GENERIC_ONLY_METHODS)
    return [prog.source] if prog.source is not None else []
