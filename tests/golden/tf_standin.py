"""Minimal numpy stand-in for the `tensorflow` module (fixture tooling ONLY).

TensorFlow is not installed in the build container and cannot be installed
(no network).  To capture golden vectors from the reference's *own* solver
source files (imported unmodified from /root/reference by make_golden.py) this
module impersonates the handful of `tf.*` symbols those files touch, with the
TF-eager semantics that matter for the numbers:

  * strict dtypes: Tensor (+) Tensor of different dtype raises (like TF eager);
  * a Python scalar combined with a Tensor adopts the Tensor's dtype;
  * `convert_to_tensor(python float)` is float32, `(python int)` is int32
    (the F3/F4 quirks of SURVEY.md section 0 depend on this);
  * `reduce_max([a, b])` stacks and reduces over everything.

It is NOT TensorFlow: reductions/pow use numpy/libm, so fixtures made through
it are labelled "reference control-flow over numpy stand-in; TensorFlow absent".
Nothing in the product path or in the GPU tests imports this file.
"""
import builtins
import contextlib
import sys
import types

import numpy as np

float16 = np.dtype('float16')
float32 = np.dtype('float32')
float64 = np.dtype('float64')
int32 = np.dtype('int32')
int64 = np.dtype('int64')
bool_ = np.dtype('bool')
complex64 = np.dtype('complex64')
complex128 = np.dtype('complex128')


def _np_of(x, like=None):
    """numpy value of x; python scalars adopt `like` dtype (TF scalar rule)."""
    if isinstance(x, Tensor):
        return x._a
    if isinstance(x, (bool, np.bool_)):
        return np.asarray(x)
    if isinstance(x, (int, float)) and like is not None:
        return np.asarray(x, dtype=like)
    if isinstance(x, float):
        return np.asarray(x, dtype=np.float32)
    if isinstance(x, int):
        return np.asarray(x, dtype=np.int32)
    return np.asarray(x)


class Tensor(object):
    __array_priority__ = 1000

    def __init__(self, a):
        self._a = np.asarray(a)

    # --- basic protocol -------------------------------------------------
    @property
    def dtype(self):
        return self._a.dtype

    @property
    def shape(self):
        return TensorShape(self._a.shape)

    @property
    def device(self):
        return ''

    def numpy(self):
        return self._a.copy() if self._a.ndim else self._a[()]

    def item(self):
        return self._a.item()

    def __bool__(self):
        return bool(self._a)

    def __float__(self):
        return float(self._a)

    def __int__(self):
        return int(self._a)

    def __len__(self):
        return self._a.shape[0]

    def __iter__(self):
        for i in builtins.range(self._a.shape[0]):
            yield Tensor(self._a[i])

    def __getitem__(self, idx):
        if isinstance(idx, Tensor):
            idx = idx._a
        if isinstance(idx, (int, np.integer)) and self._a.ndim >= 1:
            return Tensor(self._a[idx, ...])      # a writable VIEW: TF lets you `var[j].assign(v)` (sliced assign)
        return Tensor(self._a[idx])

    def assign(self, value):
        self._a[...] = _np_of(value, like=self._a.dtype)
        return self

    def __repr__(self):
        return 'standin.Tensor(%r, dtype=%s)' % (self._a, self._a.dtype)

    def __format__(self, spec):
        return format(repr(self), spec)

    __hash__ = object.__hash__

    # --- arithmetic with TF-eager strictness -----------------------------
    def _bin(self, other, op, reverse=False):
        if isinstance(other, Tensor):
            if other._a.dtype != self._a.dtype:
                raise TypeError('standin: dtype mismatch %s vs %s (TF eager would raise '
                                'InvalidArgumentError)' % (self._a.dtype, other._a.dtype))
            b = other._a
        elif isinstance(other, np.ndarray) and other.ndim > 0:
            if other.dtype != self._a.dtype:
                raise TypeError('standin: dtype mismatch %s vs ndarray %s' % (self._a.dtype, other.dtype))
            b = other
        else:
            b = np.asarray(other, dtype=self._a.dtype)
        with np.errstate(all='ignore'):
            return Tensor(op(b, self._a) if reverse else op(self._a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __rpow__(self, o): return self._bin(o, np.power, True)
    def __neg__(self): return Tensor(-self._a)
    def __abs__(self): return Tensor(np.abs(self._a))
    def __lt__(self, o): return self._bin(o, np.less)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __gt__(self, o): return self._bin(o, np.greater)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __eq__(self, o): return self._bin(o, np.equal)
    def __ne__(self, o): return self._bin(o, np.not_equal)
    def __and__(self, o): return self._bin(o, np.logical_and)
    def __or__(self, o): return self._bin(o, np.logical_or)
    def __invert__(self): return Tensor(np.logical_not(self._a))
    def __matmul__(self, o): return self._bin(o, np.matmul)


class TensorShape(tuple):
    def as_list(self):
        return list(self)


Variable_types = (Tensor,)


def convert_to_tensor(value, dtype=None, **_):
    if isinstance(value, Tensor):
        a = value._a
        if dtype is not None and a.dtype != np.dtype(dtype):
            raise TypeError('standin: convert_to_tensor dtype mismatch')
        return value
    if dtype is not None and isinstance(value, (int, float)) and not isinstance(value, bool):
        # TF converts a python scalar straight to the requested dtype (no float32 detour)
        return Tensor(np.asarray(value, dtype=np.dtype(dtype)))
    if isinstance(value, (list, tuple)):
        if len(value) == 0:
            a = np.zeros((0,), dtype=np.float32)
        elif any(isinstance(v, Tensor) for v in value):
            tdt = [v._a.dtype for v in value if isinstance(v, Tensor)]
            if any(d != tdt[0] for d in tdt):
                raise TypeError('standin: stacking tensors of different dtypes %s' % tdt)
            a = np.stack([_np_of(v, like=tdt[0]) for v in value])
        else:
            a = np.asarray(value)
            if dtype is not None and a.dtype.kind in 'fi':
                a = a.astype(np.dtype(dtype))          # TF converts python numbers straight to the requested dtype
            elif a.dtype == np.float64:
                a = a.astype(np.float32)
            elif a.dtype == np.int64:
                a = a.astype(np.int32)
    elif isinstance(value, (bool, np.bool_)):
        a = np.asarray(value)
    elif isinstance(value, int):
        a = np.asarray(value, dtype=np.int32)
    elif isinstance(value, float):
        a = np.asarray(value, dtype=np.float32)
    else:
        a = np.asarray(value)
    if dtype is not None:
        a = a.astype(np.dtype(dtype))
    return Tensor(a)


def cast(x, dtype, **_):
    return Tensor(_np_of(x).astype(np.dtype(dtype)))


def identity(x, **_):
    return Tensor(_np_of(x).copy())


def _t(x):
    return x if isinstance(x, Tensor) else convert_to_tensor(x)


def _unary(fn):
    def f(x, **_):
        with np.errstate(all='ignore'):
            return Tensor(fn(_t(x)._a))
    return f


abs = _unary(np.abs)  # noqa: A001
sqrt = _unary(np.sqrt)
ceil = _unary(np.ceil)
sin = _unary(np.sin)
cos = _unary(np.cos)
exp = _unary(np.exp)
tanh = _unary(np.tanh)
is_inf = _unary(np.isinf)
is_nan = _unary(np.isnan)
zeros_like = _unary(np.zeros_like)


def _reduce(fn):
    def f(x, axis=None, keepdims=False, **_):
        a = _t(x)._a
        with np.errstate(all='ignore'):
            return Tensor(fn(a, axis=axis, keepdims=keepdims))
    return f


reduce_max = _reduce(np.max)
reduce_min = _reduce(np.min)
reduce_sum = _reduce(np.sum)
reduce_mean = _reduce(np.mean)
reduce_all = _reduce(np.all)
reduce_any = _reduce(np.any)


def reduce_prod(x, axis=None, keepdims=False, **_):
    if isinstance(x, (tuple, list)):
        if len(x) == 0:
            return Tensor(np.asarray(1.0, dtype=np.float32))
        x = np.asarray(list(x), dtype=np.int32)
    return Tensor(np.prod(_t(x)._a, axis=axis, keepdims=keepdims).astype(_t(x)._a.dtype))


def norm(x, **_):
    a = _t(x)._a
    return Tensor(np.sqrt(np.sum(a * a)).astype(a.dtype))


def add_n(xs, **_):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return out


def multiply(a, b, **_):
    return _t(a) * b


def maximum(a, b, **_):
    a = a if isinstance(a, Tensor) else Tensor(_np_of(a, like=_t(b).dtype))
    return a._bin(b, np.maximum)


def minimum(a, b, **_):
    a = a if isinstance(a, Tensor) else Tensor(_np_of(a, like=_t(b).dtype))
    return a._bin(b, np.minimum)


def equal(a, b, **_):
    return _t(a) == b


def matmul(a, b, **_):
    return _t(a) @ b


def reshape(x, shape, **_):
    return Tensor(np.reshape(_t(x)._a, [int(s) for s in shape]))


def stack(xs, axis=0, **_):
    return Tensor(np.stack([_t(x)._a for x in xs], axis=axis))


def concat(xs, axis=0, **_):
    return Tensor(np.concatenate([_t(x)._a for x in xs], axis=axis))


def expand_dims(x, axis, **_):
    return Tensor(np.expand_dims(_t(x)._a, axis))


def transpose(x, perm=None, **_):
    return Tensor(np.transpose(_t(x)._a, perm))


def squeeze(x, axis=None, **_):
    return Tensor(np.squeeze(_t(x)._a, axis=axis))


def zeros(shape, dtype=float32, **_):
    return Tensor(np.zeros(tuple(shape), dtype=np.dtype(dtype)))


def ones(shape, dtype=float32, **_):
    return Tensor(np.ones(tuple(shape), dtype=np.dtype(dtype)))


def linspace(start, stop, num, **_):
    # tf.linspace(1., 8., n) with python floats yields float32
    return Tensor(np.linspace(start, stop, int(num), dtype=np.float64).astype(np.float32))


def range(*args, dtype=None, **_):  # noqa: A001
    a = np.arange(*[_np_of(v)[()] if isinstance(v, Tensor) else v for v in args])
    if dtype is not None:
        a = a.astype(np.dtype(dtype))
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    elif a.dtype == np.float64:
        a = a.astype(np.float32)
    return Tensor(a)


def is_tensor(x):
    return isinstance(x, Tensor)


def Variable(value, dtype=None, **_):
    if isinstance(value, Tensor):
        a = value._a
    else:
        a = np.asarray(value)
        if dtype is None and a.dtype == np.float64 and not isinstance(value, np.ndarray):
            a = a.astype(np.float32)
    if dtype is not None:
        a = a.astype(np.dtype(dtype))
    return Tensor(a)


@contextlib.contextmanager
def device(_name):
    yield


def custom_gradient(f):
    return f


def function(f=None, **_):
    if f is None:
        return lambda g: g
    return f


def executing_eagerly():
    return True


def enable_v2_behavior():
    pass


class _Model(object):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self.call(*a, **k)


class _Module(object):
    def __init__(self, *a, **k):
        pass


def _v1_assign(ref, value, **_):
    return ref.assign(value)


def install():
    """Install the stand-in as sys.modules['tensorflow'] (idempotent)."""
    me = sys.modules[__name__]
    tf = types.ModuleType('tensorflow')
    for k, v in vars(me).items():
        if not k.startswith('_') or k in ('__doc__',):
            setattr(tf, k, v)
    tf.bool = bool_
    tf.Tensor = Tensor
    math = types.ModuleType('tensorflow.math')
    for k in ('add_n', 'is_inf', 'is_nan', 'abs', 'sqrt', 'multiply', 'maximum', 'minimum',
              'reduce_max', 'reduce_min', 'reduce_sum', 'reduce_mean', 'sin', 'cos', 'exp', 'tanh'):
        setattr(math, k, getattr(me, k))
    tf.math = math
    dbg = types.ModuleType('tensorflow.debugging')
    dbg.is_numeric_tensor = lambda x: isinstance(x, Tensor) and x.dtype.kind in 'fiuc'
    tf.debugging = dbg
    ver = types.ModuleType('tensorflow.version')
    ver.VERSION = '2.0.0-standin'
    tf.version = ver
    test = types.ModuleType('tensorflow.test')
    test.is_gpu_available = lambda *a, **k: False
    tf.test = test
    keras = types.ModuleType('tensorflow.keras')
    keras.Model = _Model
    backend = types.ModuleType('tensorflow.keras.backend')
    backend.set_floatx = lambda *_a: None
    keras.backend = backend
    tf.keras = keras
    tf.Module = _Module
    tf.UnconnectedGradients = types.SimpleNamespace(ZERO='zero')
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(assign=_v1_assign))
    tf.assign = _v1_assign
    sys.modules['tensorflow'] = tf
    sys.modules['tensorflow.math'] = math
    sys.modules['tensorflow.keras'] = keras
    return tf
