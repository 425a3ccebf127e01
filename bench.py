#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on its headline configuration.

A "step" is one whole `odeint` call of config 4: linear f(t, y) = A y, dim 128,
batch 65536 PER GPU (weak scaling; the global error norm couples all ranks through one RCCL all-gather
of an 8-double record per step attempt), Dopri5, float64, rtol 1e-6, atol 1e-9, t = [0, 1].  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

roofline:     default (--fusion auto/step): the whole-attempt kernel k_step_linear_mfma<double,128,6> - all six Dopri5
              stages, error norms and dense output for a 16-row tile stay on chip, 4 planes of HBM traffic per attempt, so the
              bound is the fp64 matrix pipe: achieved = 6 * 2*dim flop per element per launch / launch duration.
              --fusion stage: one kernel per RK stage (the structure the north star describes, 34 planes per attempt);
              dominant kernel = stage 6 + error norms, 9 planes = 9 * batch*dim*8 B per launch, HBM bound.
              Durations are measured with hipEvents on the launch stream inside libmi_ode (desc.profile); `traffic` is the
              PMC figure of the committed rocprofv3 passes (profiles/*_summary.json).
cpu_baseline: the oracle (numpy restatement of the reference algorithm, kind "port") on a bounded sample of the
              same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH_PER_GPU = 65536
DIM = 128
RTOL, ATOL = 1e-6, 1e-9
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PREHEAT_CALLS = 30               # untimed calls before the W warm-up steps (see main)
FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD MI355X datasheet, FP64 matrix (= FP64 vector); the guide lists no fp64 figure


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC summary (scripts/pmc_summary.py), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')), reverse=True):
        try:
            pm = json.load(open(f)).get('pmc', {})
        except Exception:
            continue
        for k, v in pm.items():
            if kernel_substr in k and 'hbm_bytes_per_launch' in v:
                return v['hbm_bytes_per_launch']
    return None


def config4(batch, dim, seed_y):
    g2 = torch.Generator().manual_seed(2)
    S = torch.randn(dim, dim, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(dim, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(dim)
    g3 = torch.Generator().manual_seed(seed_y)
    y0 = torch.randn(batch, dim, generator=g3, dtype=torch.float64)
    return A, y0


def cpu_baseline(sample_batch=65536, repeats=2):
    """Oracle (numpy) on a bounded sample of config 4, single core (BLAS pinned to one thread)."""
    from oracle import ode_numpy as O
    try:
        from threadpoolctl import threadpool_limits
    except Exception:           # pragma: no cover
        threadpool_limits = None
    A, y0 = config4(sample_batch, DIM, 3)
    W = A.t().contiguous().numpy()
    y0 = y0.numpy()
    f = lambda t, y: y @ W  # noqa: E731
    t = np.array([0., 1.])

    def run():
        t0 = time.perf_counter()
        _, st = O.odeint(f, y0, t, rtol=RTOL, atol=ATOL, method='dopri5', return_stats=True)
        return time.perf_counter() - t0, st
    ctx = threadpool_limits(limits=1) if threadpool_limits is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        run()                                           # warm-up
        times, st = [], None
        for _ in range(repeats):
            dt, st = run()
            times.append(dt)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    wall = float(np.median(times))
    return {'value': sample_batch * DIM / wall, 'unit': 'state-elements/s', 'cores': 1, 'kind': 'port',
            'sample': 'config 4 at batch %d x dim %d (1/%d of one GPU shard), whole odeint call, numpy oracle, '
                      'median of %d runs, %.2f s each, %d attempts' % (sample_batch, DIM, BATCH_PER_GPU // sample_batch,
                                                                       repeats, wall, st.n_attempts),
            'element_steps_per_s': sample_batch * DIM * st.n_attempts / wall,
            'host_cpus': os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='rows per GPU (default: config 4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--linear-variant', type=int, default=0)
    ap.add_argument('--fusion', default='auto', choices=['auto', 'stage', 'step', 'whole'],
                    help="'stage': one kernel per RK stage (34 planes/attempt, HBM-bound); 'step': whole attempt in one kernel; "
                         "'whole'/'auto' (single GPU): the whole call in one launch")
    args = ap.parse_args()
    # stdout carries exactly ONE line (the JSON result): libraries that print to file descriptor 1 (RCCL's start-up
    # banner does) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    use_dist = world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1'     # the latter: exercise the N>1 plumbing on one GPU
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d; using %d' % (args.gpus, world, world), file=sys.stderr)

    from tfdiffeq_amd import odeint, rhs
    A, y0 = config4(args.batch, DIM, 3 + rank)
    f = rhs.Linear.from_matrix(A)
    y0 = y0.to(dev)
    t = torch.tensor([0., 1.], dtype=torch.float64)
    opts = {'profile': True, 'linear_variant': args.linear_variant, 'fusion': args.fusion}
    if group is not None:
        opts['process_group'] = group

    def step():
        out = odeint(f, y0, t, rtol=RTOL, atol=ATOL, method='dopri5', options=opts)
        return out, dict(odeint.last_stats)

    # engine / handle creation, module load and clock ramp happen here, outside both the warm-up and the timed steps: the
    # part idles between commands and the first ~20 calls after that run at ramping clocks (reported as `preheat_calls`)
    for _ in range(PREHEAT_CALLS):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    prof_last_ms = prof_all_ms = 0.0
    prof_n = 0
    stats = {}
    for _ in range(args.steps):
        out, stats = step()
        p = stats.get('profile', [0, 0, 0, 0])
        prof_last_ms += p[0]
        prof_all_ms += p[2]
        prof_n += int(p[1])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    if rank == 0:
        n_elem_rank = args.batch * DIM
        n_elem_global = n_elem_rank * n_gpus
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_elem_global * args.steps / elapsed
        attempts = int(stats.get('n_attempts', 0))
        last_ms = prof_last_ms / max(prof_n, 1)
        all_ms = prof_all_ms / max(prof_n, 1)
        step_fused = int(stats.get('n_launches', 0)) < 6 * max(attempts, 1)      # whole-attempt kernel in use
        bytes_attempt = 34 * n_elem_rank * 8                # 34 planes per Dopri5 attempt (per-stage structure)
        cfg = {'workload': 'config 4: linear f=Ay, dim 128, batch %d per GPU (global %d), Dopri5 fp64, '
                           'rtol 1e-6 atol 1e-9, t=[0,1], one odeint call per step' % (args.batch, args.batch * n_gpus),
               'parallelism': 'batch-sharded x%d; per-attempt record exchange: %s' % (n_gpus, stats.get('cross_rank', '?')),
               'fusion': 'step (whole attempt in one kernel)' if step_fused else 'stage (one kernel per RK stage)',
               'preheat_calls': PREHEAT_CALLS, 'attempts_per_step': attempts, 'accepted': int(stats.get('n_accepted', 0)),
               'nfe': int(stats.get('nfe', 0)), 'host_polls': int(stats.get('n_polls', 0)),
               'kernel_launches': int(stats.get('n_launches', 0)),
               'element_steps_per_s': n_elem_global * attempts * args.steps / elapsed,
               'attempt_kernels_ms': all_ms}
        whole = int(stats.get('n_launches', 0)) == 1          # the whole call ran in one launch (k_persist_linear_mfma)
        if whole:
            # One launch = before_integrate (2 RHS passes) + every attempt (6 stages each): nfe RHS evaluations of
            # 2*dim flop per element on the fp64 matrix pipe, plus the in-kernel grid hand-offs and controllers.
            # Algorithmic HBM bytes: f0 pass 3 planes (y0 in, f0 + solution[0] out), initial-step pass 2, each attempt 4
            # (y0, f0 in; y1, f1 out), one plane per output row.
            nfe = int(stats.get('nfe', 0))
            flops = nfe * 2 * DIM * n_elem_rank
            planes = 3 + 2 + 4 * attempts + (len(t) - 1)
            ach = flops / (last_ms * 1e-3) / 1e12 if last_ms > 0 else 0.0
            cfg['fusion'] = 'whole (the whole odeint call in one kernel launch)'
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / FP64_MFMA_PEAK_TFLOPS, 'traffic': pmc_traffic('k_persist_linear_mfma<double, 128, 6'),
                    'kernel': 'k_persist_linear_mfma<double,128,6> (before_integrate + all attempts: %d RHS evaluations on '
                              'v_mfma_f64_16x16x4_f64, error norms, in-kernel controller and dense output)' % nfe,
                    'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': planes * n_elem_rank * 8,
                    'hbm_GBps_at_algorithmic_bytes': (planes * n_elem_rank * 8) / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0,
                    'avg_launch_ms': last_ms, 'launches_timed': prof_n,
                    'peak_source': 'AMD MI355X datasheet FP64 matrix 78.6 TFLOP/s (not listed in MI355X_MICROARCH.md)'}
        elif step_fused:
            # k_step_linear_mfma<double,128,6>: 6 stages x 2*dim flop per element on the fp64 matrix pipe;
            # HBM traffic is only 5 planes (y0,f0 in; y1,f1,y_mid out), so the bound is the fp64 MFMA rate.
            flops = 6 * 2 * DIM * n_elem_rank
            ach = flops / (last_ms * 1e-3) / 1e12 if last_ms > 0 else 0.0
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / FP64_MFMA_PEAK_TFLOPS, 'traffic': pmc_traffic('k_step_linear_mfma<double, 128, 6'),
                    'kernel': 'k_step_linear_mfma<double,128,6> (all 6 Dopri5 stages + error norms + y_mid, v_mfma_f64_16x16x4_f64)',
                    'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': 5 * n_elem_rank * 8,
                    'hbm_GBps_at_algorithmic_bytes': (5 * n_elem_rank * 8) / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0,
                    'avg_launch_ms': last_ms, 'launches_timed': prof_n,
                    'peak_source': 'AMD MI355X datasheet FP64 matrix 78.6 TFLOP/s (not listed in MI355X_MICROARCH.md)'}
        else:
            bytes_last = 9 * n_elem_rank * 8                # y0,k1..k6 in; k7,y1 out
            ach = bytes_last / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0
            cfg['all_stage_kernels_GBps'] = (bytes_attempt / (all_ms * 1e-3) / 1e9) if all_ms > 0 else 0.0
            roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': ach / HBM_PEAK_GBS, 'traffic': pmc_traffic('k_stage_linear_mfma<double, 128, 6, 1, false>'),
                    'kernel': 'k_stage_linear_mfma<double,128,6,LAST_FSAL> (Dopri5 stage 6 + error norms)',
                    'algorithmic_bytes_per_launch': bytes_last, 'avg_launch_ms': last_ms, 'launches_timed': prof_n}
        stage_roof = None
        if step_fused and n_gpus == 1 and not use_dist:
            # The north star names the per-stage structure's fused "stage + error" kernel and an HBM target (>= 60 %).
            # The default schedule above replaced it (it is 1.8x faster end to end); measure that kernel too, OUTSIDE
            # the timed region, so one bench line carries both rooflines.
            sopts = dict(opts, fusion='stage')
            s_last = s_all = 0.0
            s_n = 0
            t_s = 0.0
            for i in range(2 + 5):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                odeint(f, y0, t, rtol=RTOL, atol=ATOL, method='dopri5', options=sopts)
                torch.cuda.synchronize()
                if i >= 2:
                    t_s += time.perf_counter() - t_a
                    p = dict(odeint.last_stats).get('profile', [0, 0, 0, 0])
                    s_last += p[0]
                    s_all += p[2]
                    s_n += int(p[1])
            s_last_ms, s_all_ms = s_last / max(s_n, 1), s_all / max(s_n, 1)
            bytes_last = 9 * n_elem_rank * 8
            s_ach = bytes_last / (s_last_ms * 1e-3) / 1e9 if s_last_ms > 0 else 0.0
            stage_roof = {'bound': 'hbm', 'achieved': s_ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': s_ach / HBM_PEAK_GBS,
                          'traffic': pmc_traffic('k_stage_linear_mfma<double, 128, 6, 1, false>'),
                          'kernel': 'k_stage_linear_mfma<double,128,6,LAST_FSAL> (Dopri5 stage 6 + error norms), options fusion=stage',
                          'algorithmic_bytes_per_launch': bytes_last, 'avg_launch_ms': s_last_ms, 'launches_timed': s_n,
                          'all_stage_kernels_GBps': (bytes_attempt / (s_all_ms * 1e-3) / 1e9) if s_all_ms > 0 else 0.0,
                          'ms_per_step_with_this_schedule': 1e3 * t_s / 5}
        res = {
            'metric': 'state-elements/sec (batch x dim / wall-s) Dopri5 float64',
            'value': value, 'unit': 'state-elements/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic', 'config': cfg, 'roofline': roof,
        }
        if stage_roof is not None:
            res['roofline_per_stage_schedule'] = stage_roof
        if not args.no_cpu_baseline and n_gpus == 1:
            res['cpu_baseline'] = cpu_baseline()
        else:
            res['cpu_baseline'] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
