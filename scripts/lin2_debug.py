"""Which tiles differ between the two-tile whole-call kernel and fusion='step'? (debug aid, round 3)"""
import sys, numpy as np, torch
from tfdiffeq_amd import odeint, rhs
rng = np.random.default_rng(13)
D = 128
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S_ = rng.standard_normal((D, D))
A = -0.5 * np.eye(D) + 0.5 * (S_ - S_.T) / np.sqrt(D)
f = rhs.Linear.from_matrix(torch.tensor(A))
y0 = torch.tensor(rng.standard_normal((batch, D)), dtype=torch.float64, device='cuda')
for t in ([0., 1.0], [0., 0.3, 0.35, 1.0]):
    tt = torch.tensor(t, dtype=torch.float64)
    a = odeint(f, y0, tt, method='dopri5', options={'fusion': 'step'}, rtol=1e-6, atol=1e-9)
    sa = dict(odeint.last_stats)
    b = odeint(f, y0, tt, method='dopri5', options={'fusion': 'whole'}, rtol=1e-6, atol=1e-9)
    sb = dict(odeint.last_stats)
    print('t', t, 'attempts', sa['n_attempts'], sb['n_attempts'], 'launches', sa['n_launches'], sb['n_launches'], 'dt', sa['dt'], sb['dt'])
    for j in range(1, len(t)):
        diff = (a[j] != b[j]).any(dim=1).cpu().numpy()
        rows = np.nonzero(diff)[0]
        tiles = np.unique(rows // 16)
        print('  out', j, 'rows differing', len(rows), 'tiles', len(tiles), 'max abs', float((a[j] - b[j]).abs().max()))
        if len(tiles):
            print('   tiles % 256 :', np.unique(tiles % 256)[:20], ' tiles // 256:', np.unique(tiles // 256))
            print('   first rows', rows[:10], 'cols of first row', np.nonzero((a[j][rows[0]] != b[j][rows[0]]).cpu().numpy())[0][:16])
