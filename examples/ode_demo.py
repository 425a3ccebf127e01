#!/usr/bin/env python3
"""Neural-ODE fit of the cubic spiral on an MI355X: what the reference's examples/ode_demo.py does, with this package.

Data: the spiral y' = (y**3) @ A from y0 = [2, 0] over t in [0, 25] - one `odeint` call on the fused engine (the
`rhs.CubicLinear` device right-hand side; the whole adaptive integration is one kernel launch).
Model: ODEFunc, a 2-50-2 tanh MLP on y**3 (the reference's architecture); trained on random sub-trajectories with
`odeint_adjoint` (O(1) memory: the backward pass integrates the augmented system through the plane kernels).

    python examples/ode_demo.py --niters 200
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfdiffeq_amd import odeint, odeint_adjoint, rhs  # noqa: E402


class ODEFunc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(2, 50), torch.nn.Tanh(), torch.nn.Linear(50, 2)).double()
        for m in self.net.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.normal_(m.weight, mean=0.0, std=0.1)
                torch.nn.init.zeros_(m.bias)

    def forward(self, t, y):
        return self.net(y ** 3)


def main(argv=None):
    ap = argparse.ArgumentParser('ODE demo')
    ap.add_argument('--method', default='dopri5', choices=['dopri5', 'tsit5', 'adams'])
    ap.add_argument('--data_size', type=int, default=1000)
    ap.add_argument('--batch_time', type=int, default=10)
    ap.add_argument('--batch_size', type=int, default=20)
    ap.add_argument('--niters', type=int, default=200)
    ap.add_argument('--test_freq', type=int, default=20)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('this example needs an MI355X (tfdiffeq_amd has no CPU path)')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(args.seed)
    torch.manual_seed(args.seed)

    true_y0 = torch.tensor([[2., 0.]], dtype=torch.float64, device=dev)
    t = torch.linspace(0., 25., args.data_size, dtype=torch.float64)
    true_A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    # the reference's `Lambda` (examples/ode_demo.py:32-35) as it is written: a Python callable - traced and lowered by odeint
    true_y = odeint(lambda t_, y: torch.matmul(y ** 3, true_A), true_y0, t, method='dopri5')       # [data_size, 1, 2]
    torch.cuda.synchronize()
    print('ground truth: %d points in %.2f ms (one kernel launch)' % (args.data_size, 1e3 * (time.perf_counter() - t0)))

    def get_batch():
        s = rng.choice(np.arange(args.data_size - args.batch_time), args.batch_size, replace=False)
        batch_y0 = true_y[s]                                                            # (M, 1, D)
        batch_t = t[:args.batch_time]                                                   # (T)
        batch_y = torch.stack([true_y[s + i] for i in range(args.batch_time)], dim=0)   # (T, M, 1, D)
        return batch_y0, batch_t, batch_y

    func = ODEFunc().to(dev)
    opt = torch.optim.RMSprop(func.parameters(), lr=1e-3)
    losses = []
    for itr in range(1, args.niters + 1):
        opt.zero_grad()
        batch_y0, batch_t, batch_y = get_batch()
        pred_y = odeint_adjoint(func, batch_y0, batch_t, method=args.method)
        loss = (pred_y - batch_y).abs().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if itr % args.test_freq == 0 or itr == 1:
            with torch.no_grad():
                pred = odeint(func, true_y0, t, method=args.method)
                total = float((pred - true_y).abs().mean())
            print('Iter %04d | batch loss %.6f | total loss %.6f' % (itr, losses[-1], total))
    return losses


if __name__ == '__main__':
    main()
