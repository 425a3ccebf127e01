#!/usr/bin/env python3
"""The reference's DETEST harness (tests/DETEST/run.py:25-60; problems tests/DETEST/detest.py:9-351) on the ONE-LAUNCH kernels,
with the CPU restatement of the reference path beside it.

Every problem A1..E5 from t = 0 to 20 with dopri5 AND adams (the two methods of the reference's harness, run.py:28) at tol = rtol =
atol in {1e-3, 1e-6, 1e-9}; per problem
`NFE | wall | RMS error` for
  * GPU: the problem as a device right-hand side - `rhs.CustomRowLocal` source (classes A, B, D, E: dim <= 4) or `rhs.Linear`
    (B2, C1-C4: y' = A y, dim 3 / 10 / 51 on the MFMA tile kernels, zero padded) - i.e. the whole `odeint` call is one launch of
    the whole-integration kernel; C5 (five-body problem, a [2, 3, 5] state) has no row-local kernel and runs as a Python callable
    on the plane-kernel engine (marked *);
  * CPU: oracle/ode_torch_cpu.odeint_dopri5 - the op-for-op torch-CPU eager restatement of the reference's Dopri5 path - on the
    SAME problem definitions (oracle/detest_problems.py with xp = torch), on this host; for adams the numpy restatement
    (oracle/adams_numpy.py, one core).  adams on the GPU: the variable-order kernel (csrc/mi_ode_adams_vc.h, one launch) for the
    row-local problems; the linear systems and C5 take the per-step loop over plane kernels (marked +).
The error is against a tol 1e-12 solution of the numpy oracle, as the reference's harness measures against its own tight solve.
Single trajectories: these runs are latency-bound (one thread of one wavefront per problem) - the table is the reference's own
timing harness, not a throughput claim.

    python scripts/detest_fused.py [--tols 1e-3,1e-6,1e-9] [--no-cpu]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# device code of the row-local problems: the arithmetic of oracle/detest_problems.py, operation for operation
BODIES = {
    'A1': (1, 'k[0] = -y[0];'),
    'A2': (1, 'k[0] = -(y[0] * y[0] * y[0]) / 2;'),
    'A3': (1, 'k[0] = y[0] * cos(t);'),
    'A4': (1, 'k[0] = y[0] / 4 * (1 - y[0] / 20);'),
    'A5': (1, 'k[0] = (y[0] - t) / (y[0] + t);'),
    'B1': (2, 'k[0] = 2 * (y[0] - y[0] * y[1]); k[1] = -(y[1] - y[0] * y[1]);'),
    'B3': (3, 'k[0] = -y[0]; k[1] = y[0] - y[1] * y[1]; k[2] = y[1] * y[1];'),
    'B4': (3, 'const T a = sqrt(y[0] * y[0] + y[1] * y[1]); k[0] = -y[1] - y[0] * y[2] / a; k[1] = y[0] - y[1] * y[2] / a; k[2] = y[0] / a;'),
    'B5': (3, 'k[0] = y[1] * y[2]; k[1] = -y[0] * y[2]; k[2] = (T)-0.51 * y[0] * y[1];'),
    'E1': (2, 'k[0] = y[1]; k[1] = -(y[1] / (t + 1) + (1 - (T)0.25 / ((t + 1) * (t + 1))) * y[0]);'),
    'E2': (2, 'k[0] = y[1]; k[1] = (1 - y[0] * y[0]) * y[1] - y[0];'),
    'E3': (2, 'k[0] = y[1]; k[1] = y[0] * y[0] * y[0] / 6 - y[0] + 2 * sin((T)2.78535 * t);'),
    'E4': (2, 'k[0] = y[1]; k[1] = (T).32 - (T).4 * (y[1] * y[1]);'),
    'E5': (2, 'k[0] = y[1]; k[1] = sqrt(1 + y[1] * y[1]) / (25 - t);'),
}
for _n in ('D1', 'D2', 'D3', 'D4', 'D5'):
    BODIES[_n] = (4, 'const T r = pow(y[0] * y[0] + y[1] * y[1], (T)1.5); k[0] = y[2]; k[1] = y[3]; k[2] = -y[0] / r; k[3] = -y[1] / r;')
LINEAR = ('B2', 'C1', 'C2', 'C3', 'C4')


def linear_matrix(name):
    from oracle import detest_problems as DP
    if name == 'B2':
        return np.array([[-1., 1., 0.], [1., -2., 1.], [0., 1., -1.]])
    if name == 'C1':
        return DP._band(10, [-1.] * 9 + [0.], 1.)
    if name == 'C2':
        return DP._band(10, list(np.linspace(-1., -9., 9)) + [0.], np.linspace(1., 9., 9))
    return DP._band(10 if name == 'C3' else 51, -2., 1., 1.)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tols', default='1e-3,1e-6,1e-9')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--methods', default='dopri5,adams')
    ap.add_argument('--build-only', action='store_true', help='compile the plugins (no GPU needed) and exit')
    ap.add_argument('--callable', action='store_true', help='every problem as a PYTHON callable (the reference\'s call shape): the GPU column '
                    'is then the device-controlled callable path (graph_step.DeviceControlledRK), not the one-launch kernels')
    ap.add_argument('--graph-mode', default='auto', help="--callable: options={'graph': auto|host|True|False}")
    args = ap.parse_args()
    from tfdiffeq_amd import rhs
    from oracle import adams_numpy as OA, detest_problems as DP, ode_numpy as O, ode_torch_cpu as TC   # problem definitions / checker / CPU baseline
    dev_rhs = {}
    for name, (dim, body) in ({} if args.callable else BODIES).items():
        dev_rhs[name] = rhs.CustomRowLocal(dim, body)
    if args.build_only:
        for name, f in dev_rhs.items():
            f._plugin(torch.float64)
            print('built', name)
        return
    from tfdiffeq_amd import odeint
    dev = torch.device('cuda:0')
    for name in (() if args.callable else LINEAR):
        dev_rhs[name] = rhs.Linear.from_matrix(torch.tensor(linear_matrix(name)))
    gmode = {'True': True, 'False': False}.get(args.graph_mode, args.graph_mode)
    like = torch.zeros(1, device=dev, dtype=torch.float64)
    tt = torch.tensor([0., DP.T_END], dtype=torch.float64)
    tn = np.array([0., DP.T_END])
    tols = [float(x) for x in args.tols.split(',')]
    ref = {}
    for name in DP.NAMES:
        f_np, y0_np = DP.problem(name, np)
        ref[name] = O.odeint(f_np, y0_np, tn, rtol=1e-12, atol=1e-12, method='dopri5')[1]
    print('host: %d logical CPUs, torch threads %d; GPU: %s' % (os.cpu_count(), torch.get_num_threads(), torch.cuda.get_device_name(0)))
    for method in args.methods.split(','):
      for tol in tols:
        print('======= %s | tol=%e =======' % (method, tol))
        print('%-4s | %22s | %28s | %28s' % ('', 'NFE  gpu / cpu', 'wall ms  gpu / cpu (speed-up)', 'RMS error  gpu / cpu'))
        tot = {'gn': 0, 'cn': 0, 'gt': 0.0, 'ct': 0.0, 'ge': [], 'ce': []}
        for name in DP.NAMES:
            f_t, y0_t = DP.problem(name, torch, like=like)
            calls = [0]
            if name in dev_rhs:
                f_gpu, y0_gpu, mark = dev_rhs[name], y0_t.reshape(1, -1).contiguous(), ' '
            else:
                def f_gpu(t_, y_, f_t=f_t, calls=calls):
                    calls[0] += 1
                    return f_t(t_, y_)
                y0_gpu, mark = y0_t, '*'
            walls = []
            reps = 4 if method == 'dopri5' else 2                 # 1 warm-up (engine creation, module load) + timed runs
            for rep in range(reps):
                if rep == reps - 1 and mark == ' ':
                    f_gpu.nfe = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                opts = {'graph': gmode} if (args.callable and method == 'dopri5') else None
                est = odeint(f_gpu, y0_gpu, tt, rtol=tol, atol=tol, method=method, options=opts)
                torch.cuda.synchronize()
                if rep:
                    walls.append(time.perf_counter() - t0)
            st = dict(odeint.last_stats)
            g_wall = float(np.median(walls))
            if mark == ' ':
                g_nfe = int(st['nfe']) if st.get('nfe') else int(f_gpu.nfe)      # (the per-step loop counts calls of the device RHS)
                if method == 'adams' and not str(st.get('engine', '')).startswith('fused'):
                    mark = '+'
            else:
                g_nfe = calls[0] // reps
                if 'device-controlled' in str(st.get('engine')):  # replayed evaluations do not run Python: the kernel-side count + the
                    g_nfe = 2 + int(st.get('nfe', 0))             # two evaluations of before_integrate
            g_err = float(np.sqrt(np.mean((est[1].cpu().numpy().reshape(ref[name].shape) - ref[name]) ** 2)))
            c_wall = c_err = float('nan')
            c_nfe = 0
            if not args.no_cpu:
                walls = []
                if method == 'dopri5':
                    f_c, y0_c = DP.problem(name, torch)
                    for rep in range(2):
                        t0 = time.perf_counter()
                        sol_c, st_c = TC.odeint_dopri5(f_c, y0_c, [0., DP.T_END], rtol=tol, atol=tol)
                        if rep:
                            walls.append(time.perf_counter() - t0)
                    c_nfe = int(st_c.nfe)
                    c_err = float(np.sqrt(np.mean((sol_c[1].numpy() - ref[name]) ** 2)))
                else:
                    f_np, y0_np = DP.problem(name, np)
                    t0 = time.perf_counter()
                    sol_c, st_c = OA.odeint(f_np, y0_np, tn, rtol=tol, atol=tol, method='adams', return_stats=True)
                    walls.append(time.perf_counter() - t0)
                    c_nfe = int(st_c.nfe)
                    c_err = float(np.sqrt(np.mean((np.asarray(sol_c)[1] - ref[name]) ** 2)))
                c_wall = float(np.median(walls))
            print('%-3s%s | %10d / %-9d | %9.3f / %9.3f (%5.1fx) | %12.3e / %-12.3e' % (
                name, mark, g_nfe, c_nfe, 1e3 * g_wall, 1e3 * c_wall, c_wall / g_wall if g_wall > 0 else float('nan'), g_err, c_err))
            tot['gn'] += g_nfe; tot['cn'] += c_nfe; tot['gt'] += g_wall; tot['ct'] += c_wall
            tot['ge'].append(max(g_err, 1e-300)); tot['ce'].append(max(c_err, 1e-300))
        print('Total NFE %d / %d | Total time %.3f ms / %.3f ms | GeomAvg error %.3e / %.3e' % (
            tot['gn'], tot['cn'], 1e3 * tot['gt'], 1e3 * tot['ct'], float(np.exp(np.mean(np.log(tot['ge'])))),
            float(np.exp(np.mean(np.log(tot['ce'])))) if not args.no_cpu else float('nan')))
    print('(* Python callable on the plane-kernel engine: no row-local kernel for a [2, 3, 5] state;  + adams: per-step loop over plane kernels)')


if __name__ == '__main__':
    main()
