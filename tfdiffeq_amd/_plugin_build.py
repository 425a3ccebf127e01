"""Compile and load right-hand-side plugins (csrc/mi_ode_plugin.h): hipcc, gfx950, cached by source hash."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

from . import _native as N

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-shared',
         '-I', N.CSRC, '-I', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')]
_loaded = {}


def _owned_private(path):
    """True if `path` belongs to this user and nobody else may write to it (what gets dlopen'ed must not be
    plantable by another account of the host)."""
    try:
        st = os.stat(path)
    except OSError:
        return False
    return st.st_uid == os.getuid() and (st.st_mode & 0o022) == 0


def plugin_dir():
    d = os.environ.get('TFDIFFEQ_AMD_PLUGIN_DIR')
    if not d:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_plugins')      # in-tree: travels with the checkout
    try:
        os.makedirs(d, mode=0o755, exist_ok=True)
        if os.access(d, os.W_OK) and _owned_private(d):
            return d
    except OSError:
        pass
    # package directory not writable (site-packages install): a per-user cache, never a world-shared /tmp path
    base = os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache')
    d = os.path.join(base, 'tfdiffeq_amd', 'plugins')
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError:
        d = tempfile.mkdtemp(prefix='tfdiffeq_amd_plugins_')                            # 0700, unpredictable name
    if not _owned_private(d):
        raise N.NativeError('refusing to use the plugin cache %s: it is not owned by this user or is writable by others' % d)
    return d


def _plugin_headers():
    """The headers a plugin translation unit actually sees: the #include closure of mi_ode_plugin.h inside csrc/ and include/ (a change to
    the MLP / adjoint / linear-adjoint kernels does not invalidate the cache of compiled right-hand sides)."""
    import re
    inc = os.path.join(os.path.dirname(N.CSRC.rstrip(os.sep).rsplit(os.sep, 1)[0]), 'include')
    seen, todo = [], ['mi_ode_plugin.h']
    while todo:
        name = todo.pop()
        for d in (N.CSRC, inc):
            path = os.path.join(d, os.path.basename(name))
            if os.path.exists(path) and path not in seen:
                seen.append(path)
                with open(path, 'r') as fh:
                    todo.extend(re.findall(r'^\s*#\s*include\s*"([^"]+)"', fh.read(), flags=re.M))
                break
    return sorted(seen)


def _headers_digest():
    h = hashlib.sha256()
    for path in _plugin_headers():
        with open(path, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(source, verbose=False):
    """Path of the compiled plugin for `source` (compiles on a cache miss; the key covers the kernel headers too)."""
    # (the include directories are NOT part of the key - their content is, through _headers_digest: a checkout that was moved, or copied to
    # another machine with its in-tree plugin cache, keeps hitting it)
    key = hashlib.sha256((source + _headers_digest() + ' '.join(f for f in FLAGS if not os.path.isabs(f))).encode()).hexdigest()[:24]
    out = os.path.join(plugin_dir(), 'rhs_%s.so' % key)
    if os.path.exists(out):
        if not _owned_private(out):
            raise N.NativeError('refusing to load %s: not owned by this user or writable by others' % out)
        return out
    src = out[:-3] + '.hip'
    with open(src, 'w') as fh:
        fh.write(source)
    tmp = out + '.tmp%d' % os.getpid()
    limit = float(os.environ.get('TFDIFFEQ_AMD_PLUGIN_TIMEOUT_S', '600'))
    try:
        res = subprocess.run([HIPCC] + FLAGS + [src, '-o', tmp], capture_output=True, text=True, timeout=limit)
    except subprocess.TimeoutExpired:
        for leftover in (tmp,):
            try:
                os.remove(leftover)
            except OSError:
                pass
        raise N.NativeError('compiling the RHS plugin failed: hipcc did not finish within %.0f s (TFDIFFEQ_AMD_PLUGIN_TIMEOUT_S); source: %s' % (limit, src))
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-6000:])
    if res.returncode != 0:
        raise N.NativeError('compiling the RHS plugin failed (hipcc output above); source: %s' % src)
    os.replace(tmp, out)
    return out


def build_and_load(source):
    path = build(source)
    lib = _loaded.get(path)
    if lib is None:
        import torch  # noqa: F401  (same reason as _native.load: bind to torch's HIP runtime)
        lib = C.CDLL(path)
        lib.mi_ode_plugin_get.restype = C.c_void_p
        lib.mi_ode_plugin_get.argtypes = [C.c_int]
        _loaded[path] = lib
    return lib
