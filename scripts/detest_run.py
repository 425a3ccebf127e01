#!/usr/bin/env python3
"""The reference's DETEST harness (tests/DETEST/run.py:25-60) on this package: every problem A1..E5 from t = 0 to 20,
per method and tolerance the table `NFE | time | error` against a tight (tol 1e-12) dopri5 solution, then the totals and
the geometric-mean error.  The right-hand sides are Python callables over torch ops (oracle/detest_problems.py with
xp = torch), i.e. the plane-kernel engine - the reference's only timing harness, so these are its numbers to compare with.

    python scripts/detest_run.py [--methods dopri5,adams] [--tols 1e-3,1e-6,1e-9] [--graph]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--methods', default='dopri5')
    ap.add_argument('--tols', default='1e-3,1e-6,1e-9')
    ap.add_argument('--graph', action='store_true', help="options={'graph': True}: record after the first attempt")
    ap.add_argument('--graph-mode', default=None, help="options={'graph': <auto|host|True|False>} for the adaptive RK methods")
    args = ap.parse_args()
    from tfdiffeq_amd import odeint
    from oracle import detest_problems as DP          # test infrastructure: problem definitions only
    dev = torch.device('cuda:0')
    like = torch.zeros(1, device=dev, dtype=torch.float64)
    tgrid = torch.tensor([0., DP.T_END], dtype=torch.float64)
    sol = {}
    for method in args.methods.split(','):
        for tol in [float(x) for x in args.tols.split(',')]:
            print('======= {} | tol={:e} ======='.format(method, tol))
            nfes, times, errs = [], [], []
            for name in DP.NAMES:
                f, y0 = DP.problem(name, torch, like=like)
                cnt = [0]

                def counted(t, y, f=f):
                    cnt[0] += 1
                    return f(t, y)
                if name not in sol:
                    sol[name] = odeint(counted, y0, tgrid, atol=1e-12, rtol=1e-12, method='dopri5')[1]
                cnt[0] = 0
                opts = {'graph': True} if (args.graph and method in ('dopri5', 'bosh3', 'tsit5')) else None
                if args.graph_mode is not None and method in ('dopri5', 'bosh3', 'tsit5', 'dopri8', 'adaptive_heun'):
                    opts = {'graph': {'True': True, 'False': False}.get(args.graph_mode, args.graph_mode)}
                torch.cuda.synchronize()
                t0 = time.time()
                est = odeint(counted, y0, tgrid, atol=tol, rtol=tol, method=method, options=opts)
                torch.cuda.synchronize()
                spent = time.time() - t0
                err = float(torch.sqrt(torch.mean((sol[name] - est[1]) ** 2)))
                st = dict(odeint.last_stats)
                nfe = max(cnt[0], 2 + int(st.get('nfe', 0))) if 'device-controlled' in str(st.get('engine')) else cnt[0]   # (replayed
                nfes.append(nfe); times.append(spent); errs.append(max(err, 1e-300))                 # evaluations do not run Python)
                print('{}: NFE {} | Time {:.4f} | Err {:e}'.format(name, nfe, spent, err))
            print('Total NFE {} | Total Time {:.3f} | GeomAvg Error {:e}'.format(int(np.sum(nfes)), float(np.sum(times)),
                                                                               float(np.exp(np.mean(np.log(errs))))))


if __name__ == '__main__':
    main()
