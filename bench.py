#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on its headline configuration (and, with --config, the other four).

A "step" is one whole `odeint` call.  Default = config 4: linear f(t, y) = A y, dim 128, Dopri5,
float64, rtol 1e-6, atol 1e-9, t = [0, 1]; inputs resident in HBM before the timed region.  ONE JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--config 1..5]
    python bench.py --config published [--published-all]     # the reference's own published workloads, one line each

N > 1: one process per GPU over RCCL.  Launched by the driver as `python -m torch.distributed.run ... bench.py --gpus N`
(RANK / WORLD_SIZE in the environment); a bare `python bench.py --gpus N` re-executes itself under
torch.distributed.run with N ranks.  The line is refused (exit 2) if the ranks that ran differ from --gpus.
  --scaling weak   (default) 65536 rows PER GPU; value = N x 65536 x 128 x steps / wall
  --scaling strong the fixed 65536 x 128 batch of BASELINE config 4 sharded N ways (N = 1: the same run as weak)
The documented pair for a multi-GPU node is `--gpus N` (weak, what the driver runs) and `--gpus N --scaling strong`.
N > 1 lines explain themselves: `config.per_rank[*].transport_log` = what every rank tried, in order, to carry the record and
why each attempt ended as it did; `config.transport_survey` = every transport pinned in turn for 2 + 3 calls after the timed
steps (ms_per_step each, or the reason it could not run; BENCH_NO_SURVEY=1 skips it).
The batch shards by rows (trajectories are independent); the only exchange is one 6-double record per rank per step
attempt (global error norm -> identical accept / dt decision on every rank); `config.cross_rank` names the
transport that carried it (peer-device-memory mailboxes over xGMI inside the one-launch kernel, else ncclAllGather
enqueued by libmi_ode per attempt).

roofline:     the dominant kernel of the measured schedule; durations from HIP events on the launch stream inside
              libmi_ode (desc.profile), algorithmic bytes / flops per DESIGN.md section 4; `traffic` is the PMC figure of the
              committed rocprofv3 pass of this same command (profiles/*_summary.json; a counter pass cannot run inside
              the timed process), null if none is committed for the kernel.
cpu_baseline: SURVEY.md 8(d): the op-for-op torch-CPU eager restatement of the reference path (oracle/ode_torch_cpu.py,
              kind "port") on this host at the METRIC's configuration - the full 65536 x 128 shard - in worker processes whose
              threads are pinned (the parent restricts each worker to n CPUs on n distinct cores before exec, OMP_PROC_BIND=close; 8 = one CCD of this EPYC):
              8 threads 1 warm-up + up to 3 runs, 32 threads 1 + 2, 1 thread 1 + 1 (a 10 s budget per leg: one call takes 5 - 10 s
              at this size - the eager ops stream 64 MB planes through freshly mapped memory), min / median / max reported; the numpy oracle's
              full-shard run is kept as `numpy_oracle` and supplies `parity_max_abs_diff` (GPU result vs oracle on the SAME
              full-size input).  About 50 s of CPU work.  N = 1, rank 0 only.
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

BATCH = 65536
DIM = 128
TILE_W = 128                 # tile width of the linear MFMA kernels that take DIM (set with --dim)
RTOL, ATOL = 1e-6, 1e-9
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PREHEAT_CALLS = 30               # untimed calls before the W warm-up steps (clock ramp; see main)
FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD MI355X datasheet, FP64 matrix (= FP64 vector); the guide lists no fp64 figure
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate
NOMINAL_CLOCK_MHZ = 2400.0       # the engine clock both peaks are quoted at
CPU_LEG_BUDGET_S = 10.0
CPU_PLAN = ((8, 3), (32, 2), (1, 1))    # (pinned threads, timed runs after one warm-up) of the CPU baseline, full shard each


class Hwmon(object):
    """Power, clocks and temperatures of the visible GPU(s) from the amdgpu hwmon files (what rocm-smi reads; a read costs ~0.1 ms, so the
    part can be sampled while it works).  One entry per card that exposes the files (a one-GPU box: one).  Fields: socket power and its
    cap (W), sclk / mclk (MHz), junction / memory temperature (C).  `clock_mhz` of the bench line is the kernel's own cycle counter;
    this is what the driver reports beside it - so that a box that grants 2150-2250 MHz can be told apart from a kernel that trips a
    power cap (round-4 review, item 2a)."""
    FILES = (('power_w', 'power1_input', 1e-6), ('power_cap_w', 'power1_cap', 1e-6), ('sclk_mhz', 'freq1_input', 1e-6), ('mclk_mhz', 'freq2_input', 1e-6),
             ('temp_junction_c', 'temp2_input', 1e-3), ('temp_memory_c', 'temp3_input', 1e-3))

    def __init__(self, device_index=None):
        import glob
        self.dirs = [d for d in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')) if os.path.exists(os.path.join(d, 'power1_input'))]
        self.note = '%d card(s) expose hwmon files' % len(self.dirs)
        # a node shows the files of ALL its GPUs (other tenants' included): keep the card this process computes on, found by its PCI address
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device() if device_index is None else device_index)
            bdf = '%04x:%02x:%02x.' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            mine = [d for d in self.dirs if bdf in os.path.realpath(os.path.join(d, '..', '..'))]
            if len(mine) == 1:
                self.dirs = mine
                self.note += '; this device: PCI %s0' % bdf
        except Exception as e:                                # (older torch without the PCI fields: every card is reported)
            self.note += '; device not identified (%s)' % type(e).__name__

    def read(self):
        out = []
        for d in self.dirs:
            rec = {}
            for key, name, scale in self.FILES:
                try:
                    rec[key] = round(float(open(os.path.join(d, name)).read()) * scale, 2)
                except (OSError, ValueError):
                    rec[key] = None
            out.append(rec)
        return out

    def sample_while(self, fn, seconds, period=0.01):
        """Call fn() in a loop for `seconds` while a thread reads the files every `period`: per card, mean / min / max of every field."""
        import threading
        rows, stop = [], threading.Event()

        def run():
            while not stop.is_set():
                rows.append(self.read())
                stop.wait(period)
        th = threading.Thread(target=run, daemon=True)
        t0 = time.perf_counter()
        th.start()
        n = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
        torch.cuda.synchronize()
        stop.set()
        th.join()
        cards = []
        for c in range(len(self.dirs)):
            agg = {}
            for key, _, _ in self.FILES:
                vals = [r[c][key] for r in rows if r[c].get(key) is not None]
                agg[key] = {'mean': round(sum(vals) / len(vals), 1), 'min': min(vals), 'max': max(vals)} if vals else None
            cards.append(agg)
        return {'seconds': round(time.perf_counter() - t0, 2), 'calls': n, 'samples': len(rows), 'cards': cards}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC summary (scripts/pmc_summary.py), or (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')), reverse=True):
        try:
            pm = json.load(open(f)).get('pmc', {})
        except Exception:
            continue
        for k, v in pm.items():
            if kernel_substr in k and 'hbm_bytes_per_launch' in v:
                return v['hbm_bytes_per_launch'], os.path.relpath(f, ROOT)
    return None, None


def rocprof_avg_us(kernel_substr):
    """Average duration (us) of a kernel in the newest committed rocprofv3 kernel trace of this command (profiles/r*_summary.json)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_summary.json')), reverse=True):
        try:
            ku = json.load(open(f)).get('kernel_us', {})
        except Exception:
            continue
        for k, v in ku.items():
            if kernel_substr in k and 'avg_real_us' in v:
                return v['avg_real_us'], os.path.relpath(f, ROOT)
    return None, None


def config4(batch, dim, seed_y):
    g2 = torch.Generator().manual_seed(2)
    S = torch.randn(dim, dim, generator=g2, dtype=torch.float64)
    A = -0.5 * torch.eye(dim, dtype=torch.float64) + 0.5 * (S - S.t()) / np.sqrt(dim)
    g3 = torch.Generator().manual_seed(seed_y)
    y0 = torch.randn(batch, dim, generator=g3, dtype=torch.float64)
    return A, y0


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_worker(threads, runs):
    """Child process of cpu_baseline(): the torch-CPU restatement on the full config-4 shard with `threads` threads.  The PARENT
    restricted this process to `threads` CPUs - one per physical core - before exec (the OpenMP runtime reads its binding variables
    and the affinity mask when `import torch` loads it, i.e. before any line of this function runs).  Prints one JSON line."""
    cpus = [int(c) for c in os.environ.get('BENCH_WORKER_CPUS', '').split(',') if c] or sorted(os.sched_getaffinity(0))   # (the main
    torch.set_num_threads(threads)             # thread's own mask is one place by now: the parent says what the process was given)
    from oracle import ode_torch_cpu as TC
    A, y0 = config4(BATCH, DIM, 3)
    W = A.t().contiguous()
    f = lambda t, y: y @ W  # noqa: E731
    times, st = [], None
    t_begin = time.perf_counter()
    for i in range(1 + runs):
        t0 = time.perf_counter()
        _, st = TC.odeint_dopri5(f, y0, [0., 1.], rtol=RTOL, atol=ATOL)
        if i > 0:
            times.append(time.perf_counter() - t0)
        if times and time.perf_counter() - t_begin > CPU_LEG_BUDGET_S:     # a slow host: fewer runs, never fewer than one
            break
    print(json.dumps({'threads': threads, 'cpus': cpus, 'runs': len(times), 'warmup_runs': 1, 'times_s': times, 'attempts': int(st.n_attempts)}))


def _one_cpu_per_core(allowed, n):
    """The first n CPUs of `allowed` that sit on n different physical cores (SMT siblings skipped), in CPU order: on the EPYC hosts
    of the GPU boxes CPUs 0-7 are the eight cores of one CCD, their siblings are 128-135."""
    seen, out = set(), []
    for c in sorted(allowed):
        try:
            sib = open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c).read().strip()
        except OSError:
            sib = str(c)
        if sib in seen:
            continue
        seen.add(sib)
        out.append(c)
        if len(out) == n:
            break
    return out


def _ranges(cpus):
    """[0, 1, 2, 3, 128, 129] -> '0-3,128-129'"""
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else '%d-%d' % (cpus[i], cpus[j]))
        i = j + 1
    return ','.join(out)


def cpu_baseline(gpu_result):
    """SURVEY.md 8(d) 'CPU baseline beside it'.  gpu_result: the GPU solution [2, BATCH, DIM] of the full config-4 shard."""
    import subprocess
    from oracle import ode_numpy as O
    n_all = os.cpu_count() or 1
    allowed = os.sched_getaffinity(0)
    res = {}
    attempts = None
    for nt, runs in CPU_PLAN:
        cpus = _one_cpu_per_core(allowed, nt)
        if len(cpus) < nt:
            continue
        env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt),
                   OMP_PROC_BIND='close', OMP_PLACES='cores',      # (read by the OpenMP runtime when the worker loads it)
                   BENCH_WORKER_CPUS=','.join(str(c) for c in cpus))
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(nt), '--cpu-runs', str(runs)],
                                 capture_output=True, text=True, timeout=600, env=env,
                                 preexec_fn=lambda c=cpus: os.sched_setaffinity(0, set(c)))     # before exec: the mask the runtime sees
            rec = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:                              # pragma: no cover - the line then carries what went wrong
            res['%d_threads' % nt] = {'threads': nt, 'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
            continue
        ts = rec['times_s']
        attempts = rec['attempts']
        med = float(np.median(ts))
        res['%d_thread%s' % (nt, '' if nt == 1 else 's')] = {
            'threads': nt, 'pinned_cpus': _ranges(rec['cpus']),
            'runs': rec['runs'], 'warmup_runs': rec['warmup_runs'], 'median_s': med, 'min_s': float(min(ts)), 'max_s': float(max(ts)),
            'spread_max_over_min': float(max(ts) / min(ts)), 'state_elements_per_s': BATCH * DIM / med,
            'element_steps_per_s': BATCH * DIM * attempts / med}
    # numpy oracle, one core, the full shard: the parity check at BASELINE size rides on it
    A, y0_full = config4(BATCH, DIM, 3)
    Wn = A.t().contiguous().numpy()
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:           # pragma: no cover
        ctx = None
    t0 = time.perf_counter()
    ref, st_np = O.odeint(lambda t, y: y @ Wn, y0_full.numpy(), np.array([0., 1.]), rtol=RTOL, atol=ATOL, method='dopri5',
                          return_stats=True)
    wall_np = time.perf_counter() - t0
    if ctx is not None:
        ctx.__exit__(None, None, None)
    parity = float(np.abs(gpu_result.cpu().numpy() - ref).max()) if gpu_result is not None else None
    ok = [r_ for r_ in res.values() if 'median_s' in r_]
    if not ok:
        return {'value': BATCH * DIM / wall_np, 'unit': 'state-elements/s', 'cores': 1, 'kind': 'port',
                'sample': 'numpy oracle only (the torch-CPU workers failed: %s)' % res, 'torch_threads': res}, parity, st_np.n_attempts
    best = max(ok, key=lambda r_: r_['state_elements_per_s'])
    return {'value': best['state_elements_per_s'], 'unit': 'state-elements/s', 'cores': best['threads'], 'kind': 'port',
            'sample': 'config 4 at its FULL size, batch %d x dim %d (the metric\'s configuration), whole odeint call, torch-CPU eager restatement of '
                      'the reference path (one tensor op per reference op, same host syncs) in a worker process pinned to CPUs %s, 1 warm-up + %d '
                      'runs: median %.3f s, min %.3f, max %.3f (the fastest of the 8 / 32 / 1-thread legs; the host has %d logical CPUs), %d attempts'
                      % (BATCH, DIM, best['pinned_cpus'], best['runs'], best['median_s'], best['min_s'], best['max_s'], n_all, attempts),
            'cpu_model': cpu_model(), 'host_cpus': n_all, 'allowed_cpus': _ranges(os.sched_getaffinity(0)), 'torch_threads': res,
            'numpy_oracle': {'threads': 1, 'batch': BATCH, 'wall_s': wall_np, 'state_elements_per_s': BATCH * DIM / wall_np,
                             'attempts': st_np.n_attempts}}, parity, st_np.n_attempts


def respawn(args):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report an N-GPU number\n' % (args.gpus, n_dev))
        sys.exit(2)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    os.execvpe(cmd[0], cmd, env)


def workload(cfg, args, rank, world, dev):
    """(func, y0, t, odeint kwargs, description, elements) of BASELINE config `cfg`."""
    from tfdiffeq_amd import rhs
    if cfg == 4:
        if args.scaling == 'strong':
            A, y_all = config4(args.batch, DIM, 3)
            lo, hi = args.batch * rank // world, args.batch * (rank + 1) // world
            y0 = y_all[lo:hi].contiguous()
        else:
            A, y0 = config4(args.batch, DIM, 3 + rank)
        return (rhs.Linear.from_matrix(A), y0.to(dev), torch.tensor([0., 1.], dtype=torch.float64), dict(rtol=RTOL, atol=ATOL, method='dopri5'),
                'config 4: linear f=Ay, dim %d, Dopri5 fp64, rtol 1e-6 atol 1e-9, t=[0,1]' % DIM +
                ('' if DIM == 128 else ' (NOT the BASELINE configuration: --dim %d; dims 129 .. 256 run on the 256-wide tile kernels with W streamed)' % DIM))
    if cfg == 1:
        y0 = torch.tensor([[1., 1.]], dtype=torch.float64)
        return (rhs.LotkaVolterra(1.5, 1., 3., 1.), y0.to(dev), torch.linspace(0., 10., 1001, dtype=torch.float64), dict(method='rk4'),
                'config 1: Lotka-Volterra (1.5, 1, 3, 1), y0 = [1, 1], RK4 (3/8 rule), 1000 steps on t = linspace(0, 10, 1001)')
    if cfg == 2:
        rng = np.random.default_rng(0)
        y0 = torch.tensor(rng.uniform(-2, 2, size=(4096, 2)))
        f = rhs.CubicLinear(torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64))
        return (f, y0.to(dev), torch.linspace(0., 25., 10, dtype=torch.float64), dict(method='dopri5'),
                'config 2: spiral f=(y^3)A, batch 4096 x 2, Dopri5 fp64, rtol 1e-7 atol 1e-9, t = linspace(0, 25, 10)')
    if cfg == 3:
        rng = np.random.default_rng(1)
        y0 = torch.tensor(np.array([1., 1., 1.]) + 1e-3 * rng.standard_normal((65536, 3)))
        return (rhs.Lorenz(), y0.to(dev), torch.tensor([0., 1.], dtype=torch.float64), dict(rtol=1e-6, atol=1e-9, method='tsit5'),
                'config 3: Lorenz (10, 8/3, 28), batch 65536 x 3, Tsit5 (published coefficients - the reference\'s own tableau is defective, SURVEY F6, so '
                'this configuration has NO reference anchor: parity is against the oracle\'s corrected tableau and DOP853) fp64, rtol 1e-6 atol 1e-9, t=[0,1]')
    if cfg == 5:
        gm = torch.Generator().manual_seed(4)

        def glorot(i, o):
            lim = (6.0 / (i + o)) ** 0.5
            return ((torch.rand(i, o, generator=gm) * 2 - 1) * lim).to(dev)
        mlp = rhs.MLPTanh(glorot(64, 128), torch.zeros(128, device=dev), glorot(128, 128), torch.zeros(128, device=dev),
                          glorot(128, 64), torch.zeros(64, device=dev))
        y0 = torch.randn(32768, 64, generator=torch.Generator().manual_seed(5))
        return (mlp, y0.to(dev), torch.tensor([0., 1.], dtype=torch.float64),
                dict(rtol=1e-3, atol=1e-3, method='dopri5', options={'max_num_steps': 1000}),
                'config 5: ODEFunc MLP 64-128-128-64 tanh, batch 32768 x 64 fp32, Dopri5 rtol=atol=1e-3, t=[0,1]')
    raise SystemExit('unknown --config %r' % cfg)


def published(args):
    """The reference's own published workloads (`%%time` outputs of examples/ode_usage.ipynb, the Lorenz script) - single
    trajectories with 1000 - 10 000 output times, thousands of dependent attempts on one lane.  Every system is handed to `odeint` as the
    PYTHON CALLABLE the notebook writes (examples/reference_systems.py); one JSON line per system:
      gpu_lowered_s    the call as a user makes it: traced, lowered onto the fused kernels, ONE launch (median of --steps calls)
      gpu_callable_s   options={'lower': False}: the same callable evaluated by torch between library kernels (device-controlled engine)
      gpu_catalogue_s  the hand-written device right-hand side where the catalogue has one (rhs.Lorenz, rhs.LotkaVolterra), else null
      cpu_restatement_s  the op-for-op torch-CPU eager restatement of the reference's Dopri5 path (the oracle) on this host, same callable
      published_s      the wall time the reference's notebook records (Colab CPU, TensorFlow eager) - another machine, quoted not measured
    attempts / accepted are the GPU run's, oracle_attempts / oracle_accepted the restatement's (must be equal)."""
    import statistics
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    import reference_systems as RS
    from oracle import ode_torch_cpu as TC                 # the checker / CPU baseline leg only
    from tfdiffeq_amd import odeint, rhs
    if not torch.cuda.is_available():
        sys.stderr.write('bench.py: no GPU visible (there is no CPU path)\n')
        sys.exit(3)
    dev = torch.device('cuda:0')
    gpu, cpu = RS.systems(dev), RS.systems('cpu')
    names = list(gpu) if args.published_all else list(RS.PUBLISHED)
    catalogue = {'lorenz': lambda: rhs.Lorenz(10., 8. / 3., 28.), 'predator_prey': lambda: rhs.LotkaVolterra(1.5, 1., 3., 1.)}

    def timed(fn, reps):
        out = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t0)
        return out
    for name in names:
        s = gpu[name]
        f, y0, t = s['func'], s['y0'], s['t']
        first = timed(lambda: odeint(f, y0, t), 1)[0]                       # includes tracing + loading the prebuilt kernel
        sol = odeint(f, y0, t)
        st = dict(odeint.last_stats)
        low = timed(lambda: odeint(f, y0, t), max(args.steps, 3))
        call = timed(lambda: odeint(f, y0, t, method='dopri5', options={'lower': False}), 2)
        cst = dict(odeint.last_stats)
        cat = None
        if name in catalogue:
            r = catalogue[name]()
            odeint(r, y0.reshape(1, -1), t)
            cat = statistics.median(timed(lambda: odeint(r, y0.reshape(1, -1), t), max(args.steps, 3)))
        c = cpu[name]
        t0 = time.perf_counter()
        ref, rst = TC.odeint_dopri5(c['func'], c['y0'], c['t'])
        cpu_s = time.perf_counter() - t0
        err = float((sol.cpu() - ref).abs().max())
        line = {'metric': 'wall seconds per odeint call (reference-published workload)', 'workload': name, 'source': s['source'],
                'outputs': int(t.shape[0]), 'state_elements': int(y0.numel()), 'dtype': 'f64', 'method': 'dopri5', 'rtol': 1e-7, 'atol': 1e-9,
                'published_s': s['published_s'], 'published_on': 'Colab CPU, TensorFlow eager (examples/ode_usage.ipynb %%time output; not this host)',
                'gpu_lowered_s': statistics.median(low), 'gpu_lowered_first_call_s': first, 'gpu_callable_s': min(call), 'gpu_catalogue_s': cat,
                'cpu_restatement_s': cpu_s, 'cpu': cpu_model(), 'lower': st.get('lower'), 'engine': st.get('engine'),
                'n_launches': int(st.get('n_launches', -1)), 'attempts': int(st['n_attempts']), 'accepted': int(st['n_accepted']), 'nfe': int(st.get('nfe', -1)),
                'oracle_attempts': rst.n_attempts, 'oracle_accepted': rst.n_accepted, 'oracle_nfe': rst.nfe,
                'us_per_attempt_lowered': 1e6 * statistics.median(low) / max(int(st['n_attempts']), 1),
                'us_per_attempt_callable': 1e6 * min(call) / max(int(cst.get('n_attempts', st['n_attempts'])), 1),
                'callable_engine': cst.get('engine'), 'max_abs_diff_vs_oracle': err,
                'speedup_vs_published': s['published_s'] / statistics.median(low) if s['published_s'] else None,
                'speedup_vs_cpu_restatement': cpu_s / statistics.median(low), 'data': 'the notebook\'s own initial state and output grid'}
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=BATCH, help='config 4: rows per GPU (weak) / global rows (strong)')
    ap.add_argument('--dim', type=int, default=DIM, help='config 4 at another state width (default 128 = BASELINE config 4; 129 .. 256: the 256-wide tile kernels)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--config', default='4', choices=['1', '2', '3', '4', '5', 'published'],
                    help="BASELINE.json configuration (default: the headline, 4); 'published': the workloads the reference publishes wall times "
                         "for (examples/ode_usage.ipynb, lorenz_attractor.py), as the Python callables it writes them")
    ap.add_argument('--published-all', action='store_true', help='--config published: every system of the notebook, not only the ones BASELINE.md lists')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-worker', type=int, default=0, help='internal: run the pinned torch-CPU leg with this many threads and exit')
    ap.add_argument('--cpu-runs', type=int, default=3)
    ap.add_argument('--linear-variant', type=int, default=0)
    ap.add_argument('--transport', default='auto', choices=['auto', 'peer', 'host', 'rccl', 'hook'],
                    help="N > 1: how the per-attempt record crosses ranks - 'peer' mailboxes in peer device memory (xGMI, one launch per "
                         "call), 'host' a shared host segment (one launch per call), 'rccl' ncclAllGather enqueued by libmi_ode per attempt, "
                         "'hook' torch.distributed through the callback; 'auto' tries them in that order.  What ran is in config.cross_rank")
    ap.add_argument('--fusion', default='auto', choices=['auto', 'stage', 'step', 'whole'],
                    help="'stage': one kernel per RK stage (34 planes/attempt, HBM-bound); 'step': whole attempt in one kernel; "
                         "'whole'/'auto': the whole call in one launch")
    args = ap.parse_args()
    if args.config == 'published':
        published(args)
        return
    args.config = int(args.config)
    if args.dim != DIM:                                            # (not the BASELINE configuration: no CPU leg, the width in every label)
        global TILE_W
        globals()['DIM'] = int(args.dim)
        TILE_W = 128 if args.dim <= 128 else 256
        args.no_cpu_baseline = True
    if args.cpu_worker > 0:
        cpu_worker(args.cpu_worker, args.cpu_runs)
        return
    if args.transport != 'auto':
        os.environ['TFDIFFEQ_AMD_XRANK'] = args.transport          # (read when the engine is created, on every rank)
    # A run that cannot be what was asked for ends at once, with ONE line that says why (non-zero exit status): fewer devices than ranks
    # would otherwise surface as N stack traces from the launcher, or as a hang inside the first collective.
    share_gpu = os.environ.get('BENCH_SHARE_GPU') == '1'
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < 1 or (args.gpus > n_dev and not share_gpu):
        sys.stderr.write('bench.py: --gpus %d but %d device(s) visible (there is no CPU path; BENCH_SHARE_GPU=1 lets several ranks share device 0 '
                         'for plumbing tests)\n' % (args.gpus, n_dev))
        sys.exit(3)
    # ... and a run that hangs (a peer mailbox that never answers, a collective with a dead rank) ends after BENCH_WATCHDOG_S seconds
    # (default 900: half of the driver's own limit) instead of taking the driver's whole time-out with it.  The in-kernel waits are bounded
    # already (~10 s per hand-off across ranks, then the next transport); this is the last line of defence.
    import signal

    def _watchdog(signum, frame):
        sys.stderr.write('bench.py: watchdog - no result after %s s (rank %s of %s); giving up\n' % (
            os.environ.get('BENCH_WATCHDOG_S', '900'), os.environ.get('RANK', '0'), os.environ.get('WORLD_SIZE', '1')))
        os._exit(4)
    signal.signal(signal.SIGALRM, _watchdog)
    signal.alarm(int(os.environ.get('BENCH_WATCHDOG_S', '900')))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn(args)
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
    if args.gpus != world:
        sys.stderr.write('bench.py: --gpus %d but %d rank(s) were launched; refusing to print a line whose n_gpus is not the '
                         'number of ranks that ran\n' % (args.gpus, world))
        sys.exit(2)
    if args.config != 4 and world > 1:
        raise SystemExit('configs 1, 2, 3, 5 are single-GPU workloads (BASELINE.json)')
    # BENCH_SHARE_GPU=1 (test hook for one-GPU boxes): every rank computes on cuda:0 and the group is gloo - the whole N > 1 code
    # path of this file (sharding, per-rank logs, the transport survey) runs, the kernels of the ranks share the device
    share = os.environ.get('BENCH_SHARE_GPU') == '1'
    torch.cuda.set_device(0 if share else local_rank)
    dev = torch.device('cuda', 0 if share else local_rank)
    cdev = torch.device('cpu') if share else dev                           # where collective scalars live (gloo: host tensors)
    group = None
    use_dist = world > 1 or os.environ.get('BENCH_FORCE_DIST') == '1'     # the latter: exercise the N>1 plumbing on one GPU
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get('BENCH_PG_TIMEOUT_S', '180')))   # (default 3 min: a dead rank should cost minutes, not the run)
        if share:
            dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=pg_timeout)
        else:
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        group = dist.group.WORLD
    n_gpus = world

    from tfdiffeq_amd import odeint
    f, y0, t, kw, desc = workload(args.config, args, rank, world, dev)
    opts = dict(kw.pop('options', None) or {})
    opts.update({'profile': True, 'fusion': args.fusion})
    if args.config == 4:
        opts['linear_variant'] = args.linear_variant
    if kw['method'] in ('rk4', 'euler'):
        opts = {'fusion': args.fusion}
    if group is not None:
        opts['process_group'] = group

    def step():
        out = odeint(f, y0, t, options=opts, **kw)
        return out, dict(odeint.last_stats)

    # engine / handle creation and module load happen in the first call
    step()
    torch.cuda.synchronize()
    # A full collection of the interpreter's object graph (torch alone keeps ~1e6 objects alive) takes 40-50 ms; when one lands inside
    # a ten-call timed region it shows up as +4 ms per step on the latency-bound configurations (measured: config 2 at 5.5 instead of
    # 1.6 ms).  Collect now and keep the collector out of the timed region, as `timeit` does.  NOW means: BEFORE the clock ramp and the
    # warm-up steps, not between them and the timed steps - round 5 (scripts/power_trace.py, profiles/r05_power_trace_config4.txt): the part
    # drops its engine clock within milliseconds of going idle and needs tens of milliseconds of work to get it back, and rounds 1-4 put
    # this 40-50 ms host-side pause right in front of a 13 ms timed region.  The same box then reads 1.70 ms per step in the timed steps
    # (kernel clock 2167 MHz) and 1.29 ms (2394 MHz) in a 1.5 s loop of the same call seconds later.
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    hw = Hwmon()
    smi = {'source': 'amdgpu hwmon files: ' + hw.note, 'before_warmup': hw.read()}
    # clock ramp (reported as `preheat_calls`), then the W warm-up steps, then - with nothing in between but the barrier and the
    # synchronisation the contract asks for - the timed steps
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
    clk_sum, clk_n = 0.0, 0
    stats = {}
    out = None
    for _ in range(args.steps):
        out, stats = step()
        p = stats.get('profile', [0, 0, 0, 0])
        prof_last_ms += p[0]
        prof_all_ms += p[2]
        prof_n += int(p[1])
        if stats.get('clock_mhz', 0) > 0:
            clk_sum += stats['clock_mhz']
            clk_n += 1
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t_start                # this rank's own clock, before the closing barrier (config.per_rank)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    smi['after_timed_steps'] = hw.read()                      # (six file reads, ~0.5 ms: after the clock is stopped)
    if gc_was_enabled:
        gc.enable()
    per_rank = None
    if use_dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        per_rank = [None] * world
        att = max(int(stats.get('n_attempts', 0)), 1)
        dist.all_gather_object(per_rank, {'rank': rank, 'rows': int(y0.shape[0]), 'attempts': int(stats.get('n_attempts', 0)),
                                          'launches': int(stats.get('n_launches', 0)), 'cross_rank': stats.get('cross_rank', '?'),
                                          'ms_per_step': 1e3 * my_elapsed / args.steps,
                                          'us_per_attempt': 1e6 * my_elapsed / args.steps / att,
                                          'kernel_ms_per_step': prof_last_ms / max(prof_n, 1),
                                          # time workgroup 0 of this rank's whole-call kernel spent in the grid hand-offs of the last call
                                          # (the cross-rank exchange included): what the sharding costs per call, seen from inside
                                          'handoff_us': stats.get('handoff_us'),
                                          # what this rank tried, in order, to carry the per-attempt record, and why each ended as it did
                                          'transport_log': stats.get('cross_rank_log')})
    # The same call in a loop for 1.5 s with the files sampled every 10 ms, OUTSIDE the timed region (after it): power against its cap,
    # the clock the driver reports and the temperatures while the kernel of the timed steps is all the part does.
    if len(hw.dirs) > 0 and os.environ.get('BENCH_NO_SMI_LOOP') != '1':
        clk2 = []

        def _one():
            _, st_ = step()
            if st_.get('clock_mhz', 0) > 0:
                clk2.append(st_['clock_mhz'])
        smi['sustained_loop'] = hw.sample_while(_one, 1.5)
        smi['sustained_loop']['kernel_clock_mhz'] = round(sum(clk2) / len(clk2), 1) if clk2 else None
        if use_dist:
            dist.barrier()
    survey = None
    if use_dist and args.config == 4 and args.transport == 'auto' and os.environ.get('BENCH_NO_SURVEY') != '1':
        # First contact with a multi-GPU node should explain itself: OUTSIDE the timed region, every transport in turn (pinned, so
        # nothing falls through silently) runs 2 + 3 calls; the line then carries what each one did - or why it could not run -
        # next to the one the timed steps used.  Collective-safe: every rank walks the same list.
        from tfdiffeq_amd import solvers as _solvers
        survey = []
        for tr in ('peer', 'host', 'rccl', 'hook'):
            os.environ['TFDIFFEQ_AMD_XRANK'] = tr
            _solvers.clear_engine_cache()                      # (the engine key does not name the transport)
            entry = {'requested': tr}
            try:
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                dist.barrier()
                t_s = time.perf_counter()
                st_s = {}
                for _ in range(3):
                    _, st_s = step()
                torch.cuda.synchronize()
                el_s = torch.tensor([time.perf_counter() - t_s], dtype=torch.float64, device=cdev)
                dist.all_reduce(el_s, op=dist.ReduceOp.MAX)
                ran = str(st_s.get('cross_rank', '?'))
                entry.update({'ran': ran, 'pinned_transport_ran': {'peer': 'peer device memory', 'host': 'host segment', 'rccl': 'ncclAllGather',
                                                                   'hook': 'allgather hook'}[tr] in ran,
                              'ms_per_step': 1e3 * float(el_s.item()) / 3, 'launches': int(st_s.get('n_launches', 0)),
                              'attempts': int(st_s.get('n_attempts', 0)), 'log_rank0': st_s.get('cross_rank_log')})
            except Exception as e:                             # pragma: no cover
                entry['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
            survey.append(entry)
        os.environ.pop('TFDIFFEQ_AMD_XRANK', None)
        _solvers.clear_engine_cache()

    if rank == 0:
        n_elem_rank = int(y0.numel())
        rows_global = int(y0.shape[0]) * n_gpus if (args.config == 4 and args.scaling == 'weak') else \
            (args.batch if args.config == 4 else int(y0.shape[0]))
        n_elem_global = rows_global * int(y0.shape[1])
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_elem_global * args.steps / elapsed
        attempts = int(stats.get('n_attempts', 0))
        launches = int(stats.get('n_launches', 0))
        last_ms = prof_last_ms / max(prof_n, 1)
        all_ms = prof_all_ms / max(prof_n, 1)
        cfg = {'workload': desc + '; one odeint call per step; rows per GPU %d, global %d' % (int(y0.shape[0]), rows_global),
               'parallelism': 'batch-sharded x%d (%s scaling)' % (n_gpus, args.scaling) if args.config == 4 else 'single GPU',
               # what carried the per-attempt records in the TIMED steps, and how many ranks the RCCL communicator that did it had (0: the
               # records did not travel through RCCL - peer mailboxes, the host segment or, under BENCH_SHARE_GPU, a gloo group)
               'records_transport': stats.get('cross_rank', 'single rank'),
               'rccl_ranks': n_gpus if (use_dist and 'ncclAllGather' in str(stats.get('cross_rank', ''))) else 0,
               'process_group_backend': (dist.get_backend(group) if group is not None else None),
               'cross_rank': stats.get('cross_rank', 'single rank'), 'transport_requested': args.transport,
               'preheat_calls': PREHEAT_CALLS, 'attempts_per_step': attempts, 'accepted': int(stats.get('n_accepted', 0)),
               'nfe': int(stats.get('nfe', 0)), 'host_polls': int(stats.get('n_polls', 0)), 'kernel_launches': launches,
               'element_steps_per_s': n_elem_global * max(attempts, 1) * args.steps / elapsed,
               'us_per_attempt': 1e3 * ms_per_step / max(attempts, 1), 'attempt_kernels_ms': all_ms,
               # shader clock the whole-call kernel itself observed (its cycle counter against the 100 MHz constant clock), mean
               # over the timed steps: latency-bound configurations move with it (a one-wavefront kernel is granted whatever the
               # governor leaves it at)
               'clock_mhz': (clk_sum / clk_n) if clk_n else None,
               'handoff_us': stats.get('handoff_us'),           # workgroup 0's time in the grid hand-offs of the last timed call
               # what the driver reports beside it (class Hwmon): a reading right before and right after the timed steps, and the same call
               # in a 1.5 s loop afterwards with power / sclk / temperatures sampled every 10 ms
               'smi': smi}
        if per_rank is not None:
            cfg['per_rank'] = per_rank
        if survey is not None:
            cfg['transport_survey'] = survey
        elt = y0.element_size()
        nfe = int(stats.get('nfe', 0))
        roof = None
        stage_roof = None
        if args.config == 4:
            step_fused = launches < 6 * max(attempts, 1)        # whole-attempt kernel in use
            whole = launches == 1                                # the whole call ran in one launch (k_persist_linear_mfma)
            bytes_attempt = 34 * n_elem_rank * 8                 # 34 planes per Dopri5 attempt (per-stage structure)
            peak_src = 'AMD MI355X datasheet FP64 matrix 78.6 TFLOP/s (not listed in MI355X_MICROARCH.md)'
            if whole:
                # One launch = before_integrate (2 RHS passes) + every attempt (6 stages each): nfe RHS evaluations of
                # 2*dim flop per element on the fp64 matrix pipe, plus the in-kernel grid hand-offs and controllers.
                # Algorithmic HBM bytes: f0 pass 3 planes (y0 in, f0 + solution[0] out), initial-step pass 2, each attempt 4
                # (y0, f0 in; y1, f1 out), one plane per output row.
                flops = nfe * 2 * DIM * n_elem_rank
                planes = 3 + 2 + 4 * attempts + (len(t) - 1)
                ach = flops / (last_ms * 1e-3) / 1e12 if last_ms > 0 else 0.0
                cfg['fusion'] = 'whole (the whole odeint call in one kernel launch)'
                traffic, src = pmc_traffic('k_persist_linear_mfma<double, %d, 6' % TILE_W)
                roof = {'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP64_MFMA_PEAK_TFLOPS,
                        'traffic': traffic, 'traffic_source': src,
                        'kernel': 'k_persist_linear_mfma<double,%d,6> (before_integrate + all attempts: %d RHS evaluations on '
                                  'v_mfma_f64_16x16x4_f64, error norms, in-kernel controller and dense output)' % (TILE_W, nfe),
                        'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': planes * n_elem_rank * 8,
                        'hbm_GBps_at_algorithmic_bytes': (planes * n_elem_rank * 8) / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0,
                        'avg_launch_ms': last_ms, 'launches_timed': prof_n, 'peak_source': peak_src,
                        'frac_source': 'HIP events around the launch, inside this run (achieved = algorithmic flops / avg_launch_ms)'}
                rp_us, rp_src = rocprof_avg_us('k_persist_linear_mfma<double, %d, 6' % TILE_W)
                if rp_us:                                       # NOT measured in this run: the committed profiler pass of the same command
                    roof['frac_rocprof_committed'] = flops / (rp_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS   # (clocks ~2.5 % lower under
                    roof['rocprof_committed_avg_launch_ms'] = rp_us * 1e-3                                    # the profiler); a kernel change
                    roof['rocprof_committed_source'] = rp_src                                                 # makes it stale until re-profiled
            elif step_fused:
                flops = 6 * 2 * DIM * n_elem_rank
                ach = flops / (last_ms * 1e-3) / 1e12 if last_ms > 0 else 0.0
                cfg['fusion'] = 'step (whole attempt in one kernel)'
                traffic, src = pmc_traffic('k_step_linear_mfma<double, 128, 6')
                roof = {'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP64_MFMA_PEAK_TFLOPS,
                        'traffic': traffic, 'traffic_source': src,
                        'kernel': 'k_step_linear_mfma<double,128,6> (all 6 Dopri5 stages + error norms + dense output, v_mfma_f64_16x16x4_f64)',
                        'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': 4 * n_elem_rank * 8,
                        'hbm_GBps_at_algorithmic_bytes': (4 * n_elem_rank * 8) / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0,
                        'avg_launch_ms': last_ms, 'launches_timed': prof_n, 'peak_source': peak_src}
            else:
                bytes_last = 9 * n_elem_rank * 8                # y0,k1..k6 in; k7,y1 out
                ach = bytes_last / (last_ms * 1e-3) / 1e9 if last_ms > 0 else 0.0
                cfg['fusion'] = 'stage (one kernel per RK stage)'
                cfg['all_stage_kernels_GBps'] = (bytes_attempt / (all_ms * 1e-3) / 1e9) if all_ms > 0 else 0.0
                traffic, src = pmc_traffic('k_stage_linear_mfma<double, 128, 6, 1, false>')
                roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                        'traffic': traffic, 'traffic_source': src,
                        'kernel': 'k_stage_linear_mfma<double,128,6,LAST_FSAL> (Dopri5 stage 6 + error norms)',
                        'algorithmic_bytes_per_launch': bytes_last, 'avg_launch_ms': last_ms, 'launches_timed': prof_n}
            if step_fused and n_gpus == 1 and not use_dist:
                # The north star names the per-stage structure's fused "stage + error" kernel and an HBM target (>= 60 %).
                # The default schedule above replaced it (it is 1.8x faster end to end); measure that kernel too, OUTSIDE
                # the timed region, so one bench line carries both rooflines.
                sopts = dict(opts, fusion='stage')
                s_last = s_all = t_s = 0.0
                s_n = 0
                for i in range(2 + 5):
                    torch.cuda.synchronize()
                    t_a = time.perf_counter()
                    odeint(f, y0, t, options=sopts, **kw)
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
                traffic, src = pmc_traffic('k_stage_linear_mfma<double, 128, 6, 1, false>')
                stage_roof = {'bound': 'hbm', 'achieved': s_ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': s_ach / HBM_PEAK_GBS,
                              'traffic': traffic, 'traffic_source': src,
                              'kernel': 'k_stage_linear_mfma<double,128,6,LAST_FSAL> (Dopri5 stage 6 + error norms), options fusion=stage',
                              'measured': 'OUTSIDE the timed region: 2 + 5 extra odeint calls with the per-stage schedule after the timed steps',
                              'algorithmic_bytes_per_launch': bytes_last, 'avg_launch_ms': s_last_ms, 'launches_timed': s_n,
                              'all_stage_kernels_GBps': (bytes_attempt / (s_all_ms * 1e-3) / 1e9) if s_all_ms > 0 else 0.0,
                              'ms_per_step_with_this_schedule': 1e3 * t_s / 5}
        elif args.config == 5:
            # k_persist_mlp: nfe evaluations of the 64-128-128-64 MLP = 2*(64*128 + 128*128 + 128*64) flop per row each, fp32 MFMA
            rows = int(y0.shape[0])
            flops = nfe * 2 * (64 * 128 + 128 * 128 + 128 * 64) * rows
            dur_ms = last_ms if last_ms > 0 else ms_per_step
            ach = flops / (dur_ms * 1e-3) / 1e12
            planes = 3 + 2 + 4 * attempts + (len(t) - 1)
            traffic, src = pmc_traffic('k_persist_mlp')
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / FP32_MFMA_PEAK_TFLOPS,
                    'traffic': traffic, 'traffic_source': src,
                    'kernel': 'k_persist_mlp<64,128,6> (whole call in one launch: %d MLP evaluations on v_mfma_f32_16x16x4_f32 + tanh)' % nfe,
                    'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': planes * n_elem_rank * elt,
                    'avg_launch_ms': dur_ms, 'launches_timed': prof_n}
        else:
            # configs 1-3: state << cache; the call is bound by per-attempt latency (kernel stages + in-kernel grid hand-off +
            # controller), not by bytes.  The HBM figure is reported for the contract's sake; the meaningful one is us_per_attempt.
            T = len(t)
            if kw['method'] == 'rk4':
                alg = (1 + T) * n_elem_rank * elt
                kern = 'k_fixed_rowlocal (whole fixed-grid integration in one launch: y0 in, T solution rows out)'
                sub = 'k_fixed_rowlocal'
            else:
                alg = (1 + T) * n_elem_rank * elt
                kern = 'k_persist_rowlocal (whole adaptive integration in one launch: y0 in, T solution rows out; state in registers)'
                sub = 'k_persist_rowlocal'
            dur_ms = last_ms if last_ms > 0 else ms_per_step
            ach = alg / (dur_ms * 1e-3) / 1e9
            traffic, src = pmc_traffic(sub)
            roof = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': traffic,
                    'traffic_source': src, 'kernel': kern, 'algorithmic_bytes_per_launch': alg, 'avg_launch_ms': dur_ms,
                    'launches_timed': prof_n,
                    'note': 'latency-bound (state fits in cache): see config.us_per_attempt; DESIGN.md section 5 has the per-attempt budget'}
        dtype_name = {torch.float64: 'f64', torch.float32: 'f32'}[y0.dtype]
        res = {
            'metric': 'state-elements/sec (batch x dim / wall-s) %s' % ('Dopri5 float64' if args.config == 4 else 'config %d' % args.config),
            'value': value, 'unit': 'state-elements/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': dtype_name, 'data': 'synthetic', 'config': cfg, 'roofline': roof,
        }
        if stage_roof is not None:
            res['roofline_per_stage_schedule'] = stage_roof
        # The pool's boxes grant the kernels that keep every matrix pipe busy different clocks (2122 ... 2398 MHz seen for the same
        # binary, profiles/r04_clock_variation.txt); `frac` stays what the contract defines (against the nominal peak at 2400 MHz),
        # this is the same figure against the peak at the clock THIS run was granted (counted by the kernel itself)
        clk = cfg.get('clock_mhz')
        if roof.get('bound') == 'mfma' and clk:
            roof['nominal_clock_mhz'] = NOMINAL_CLOCK_MHZ
            roof['granted_clock_mhz'] = clk
            roof['frac_at_granted_clock'] = roof['frac'] * NOMINAL_CLOCK_MHZ / clk
        res['cpu_baseline'] = None
        if not args.no_cpu_baseline and n_gpus == 1 and args.config == 4 and args.batch == BATCH:
            res['cpu_baseline'], res['parity_max_abs_diff'], ref_attempts = cpu_baseline(out)
            res['parity_attempts'] = {'gpu': attempts, 'oracle': ref_attempts}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
