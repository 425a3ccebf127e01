"""world_size-2 tests of the batch-sharded (N > 1) path on CPU with the gloo backend.

What shards: trajectories are independent, the ONLY coupling is the reference's global error
norm and the single global dt.  Per step attempt every rank contributes one small record
{max|y0|, max|y1|, sum err^2, flag, N_local}; the records are all-gathered and combined in rank order
(max, max, sum, max, sum); every rank then takes the identical accept / dt decision.

On the GPU the record exchange is the engine's all-gather hook (RCCL) and the combine runs in the device
controller; here the same product code paths that run on the host - `misc._Exchange.combine`,
`misc._ratio_from_norms`, `misc._optimal_step_size` - are exercised under gloo, with the oracle standing in for
the per-shard stage arithmetic (the oracle is test infrastructure).  The property checked is the one that makes
sharding legal: G = 2 reproduces the G = 1 step sequence and solution.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ode_numpy as O
from oracle.rhs_numpy import make_rhs

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)


def _worker_exchange(rank, port, outdir):
    from tfdiffeq_amd.misc import _Exchange
    _init(rank, port)
    try:
        ex = _Exchange(dist.group.WORLD)
        # [ncomp = 2, 5]: {max|y0|, max|y1|, sum err^2, flag, N}
        rec = torch.tensor([[1.0 + rank, 5.0 - rank, 0.25 * (rank + 1), float(rank == 1), 100.0 * (rank + 1)],
                            [7.0 - 3 * rank, 2.0 + rank, 1.5, 0.0, 10.0]], dtype=torch.float64)
        out = ex.combine(rec, max_slots=(0, 1, 3)).numpy()
        np.save(os.path.join(outdir, 'ex%d.npy' % rank), out)
    finally:
        dist.destroy_process_group()


def test_record_exchange_combines_in_rank_order():
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_exchange, args=(port, d), nprocs=WORLD, join=True)
        a, b = np.load(os.path.join(d, 'ex0.npy')), np.load(os.path.join(d, 'ex1.npy'))
    expect = np.array([[2.0, 5.0, 0.75, 1.0, 300.0], [7.0, 3.0, 3.0, 0.0, 20.0]])
    np.testing.assert_array_equal(a, expect)
    np.testing.assert_array_equal(b, expect)        # every rank ends with the identical combined record


def _sharded_dopri5(y0_shard, f, t_end, rtol, atol, exchange):
    """The sharded attempt loop: oracle arithmetic per shard, product host code for exchange + controller."""
    from tfdiffeq_amd import misc
    func = lambda t, ys: (f(t, ys[0]),)  # noqa: E731
    y = (y0_shard,)
    npdt = np.dtype(np.float64)
    f0 = func(np.float64(0.0), y)

    def gsum(x, sub):
        sc = atol + np.abs(y[0]) * rtol
        v = (x - sub) / sc if sub is not None else x / sc
        rec = torch.tensor([[float(np.sum(v * v)), float(v.size)]], dtype=torch.float64)
        h = exchange.combine(rec, max_slots=()).numpy()
        return misc._norm_from_sumsq(h[0, 0], h[0, 1], npdt)
    d0, d1 = gsum(y[0], None), gsum(f0[0], None)                       # misc._select_initial_step, globally normed
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = func(np.float64(h0), (y[0] + h0 * f0[0],))
    d2 = gsum(f1[0], f0[0]) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1. / 5.)
    dt, t, trace = np.float64(min(100 * h0, h1)), np.float64(0.0), []
    safety, dfactor = np.float64(np.float32(0.9)), np.float64(np.float32(0.2))
    f_cur = f0
    while t < t_end:
        y1, f1_, err, k = O.runge_kutta_step(func, y, f_cur, t, dt, O.DOPRI5)
        rec = torch.tensor([[np.abs(y[0]).max(), np.abs(y1[0]).max(), float(np.sum(err[0] ** 2)), 0.0, float(err[0].size)]],
                           dtype=torch.float64)
        g = exchange.combine(rec, max_slots=(0, 1, 3)).numpy()[0]
        ratio = misc._ratio_from_norms(g, g[4], rtol, atol, npdt)
        accept = bool(ratio <= 1)
        dt_next = misc._optimal_step_size(dt, (ratio,), safety=safety, ifactor=np.float64(10.0), dfactor=dfactor, order=5)
        trace.append((float(t), float(dt), float(accept)))
        if accept:
            t, y, f_cur = t + dt, y1, f1_
        dt = dt_next
    return y[0], np.asarray(trace)


def _worker_lockstep(rank, port, outdir):
    from tfdiffeq_amd.misc import _Exchange
    _init(rank, port)
    try:
        rng = np.random.default_rng(1)
        y0 = np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((64, 3))
        shard = y0[rank * 32:(rank + 1) * 32]
        f = make_rhs('lorenz', {'sigma': 10., 'beta': 8. / 3., 'rho': 28.})
        y, trace = _sharded_dopri5(shard, f, 0.5, 1e-6, 1e-9, _Exchange(dist.group.WORLD))
        np.save(os.path.join(outdir, 'y%d.npy' % rank), y)
        np.save(os.path.join(outdir, 'tr%d.npy' % rank), trace)
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_single_rank_step_sequence():
    from tfdiffeq_amd.misc import _Exchange
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_lockstep, args=(port, d), nprocs=WORLD, join=True)
        ys = [np.load(os.path.join(d, 'y%d.npy' % r)) for r in range(WORLD)]
        trs = [np.load(os.path.join(d, 'tr%d.npy' % r)) for r in range(WORLD)]
    np.testing.assert_array_equal(trs[0], trs[1])                      # lock-step: identical decisions on every rank
    rng = np.random.default_rng(1)
    y0 = np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((64, 3))
    f = make_rhs('lorenz', {'sigma': 10., 'beta': 8. / 3., 'rho': 28.})
    y_single, tr_single = _sharded_dopri5(y0, f, 0.5, 1e-6, 1e-9, _Exchange(None))      # G = 1
    assert trs[0].shape == tr_single.shape and np.array_equal(trs[0][:, 2], tr_single[:, 2])
    np.testing.assert_allclose(trs[0][:, 1], tr_single[:, 1], rtol=1e-9)               # same dt sequence (sum order differs)
    np.testing.assert_allclose(np.concatenate(ys), y_single, rtol=1e-9, atol=1e-12)
    # and the single-rank model is the oracle's own driver (global norm, F3)
    _, st = O.odeint(f, y0, np.array([0., 0.5]), rtol=1e-6, atol=1e-9, method='dopri5', return_stats=True)
    ref_tr = np.asarray(st.trace)
    assert np.array_equal(ref_tr[:, 2], tr_single[:, 2])
    np.testing.assert_allclose(ref_tr[:, 1], tr_single[:, 1], rtol=1e-9)     # mean((e/tol)^2) vs sum e^2/(N tol^2) rounding


def test_fused_engine_requires_a_hook_for_world_size_gt_1():
    """Host-side contract of the engine descriptor: world_size > 1 carries the exchange hook + buffers."""
    from tfdiffeq_amd import _native as N
    d = N.Desc()
    names = [f for f, _ in N.Desc._fields_]
    for f in ('world_size', 'rank', 'allgather', 'allgather_user', 'exchange_send_dev', 'exchange_recv_dev'):
        assert f in names
    assert d.world_size == 0 and not d.allgather
