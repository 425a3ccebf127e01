import numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tfdiffeq_amd import odeint, rhs
from tfdiffeq_amd import plugin_examples as PE
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
mk = lambda *s: (0.7 * torch.randn(*s, generator=g, dtype=torch.float64) / s[0] ** 0.5)
f = rhs.MLP(mk(6, 24), None, mk(24, 24), None, mk(24, 6), None, activation='tanh')
ring = PE.reaction_diffusion_ring(100)
for name, func, dim, batches in (('mlp', f, 6, (3, 50000)), ('ring', ring, 100, (3, 4000))):
    for batch in batches:
        y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dev)
        for t in (torch.linspace(0, 2, 101, dtype=torch.float64), -torch.linspace(0, 1, 40, dtype=torch.float64)):
            for opts in (None, {'first_step': 0.01}, {'max_num_steps': 3}):
                try:
                    a = odeint(func, y0, t, method='dopri5', rtol=1e-6, atol=1e-8, options=opts)
                    sa = dict(odeint.last_stats)
                    b = odeint(lambda t_, y: func.forward(t_, y), y0, t, method='dopri5', rtol=1e-6, atol=1e-8, options=opts)
                    sb = dict(odeint.last_stats)
                    print(name, batch, len(t), opts, 'launches', sa.get('n_launches'), 'attempts', sa['n_attempts'], sb['n_attempts'], 'dev %.2e' % float((a - b).abs().max()))
                except AssertionError as e:
                    try:
                        odeint(lambda t_, y: func.forward(t_, y), y0, t, method='dopri5', rtol=1e-6, atol=1e-8, options=opts)
                        print(name, batch, len(t), opts, 'KERNEL RAISED BUT CALLABLE DID NOT:', e)
                    except AssertionError as e2:
                        print(name, batch, len(t), opts, 'both raise:', str(e)[:60], '|', str(e2)[:60])
