import sys
import torch
sys.path.insert(0, '.')
from scripts.linadj_check import grads, dev
from tfdiffeq_amd import models
torch.manual_seed(3)
dim, batch = 128, 512
func = models.LinearODEFunc(dim, bias=False, dtype=torch.float64).to(dev)
g = torch.Generator().manual_seed(1)
y0 = torch.randn(batch, dim, generator=g, dtype=torch.float64).to(dev)
t = torch.tensor([0.0, 1.0], dtype=torch.float64)
w = torch.randn(2, batch, dim, generator=g, dtype=torch.float64).to(dev)
tol = dict(rtol=1e-6, atol=1e-9, method='dopri5')
a = grads(func, y0, t, w, True, **tol)
b = grads(func, y0, t, w, False, **tol)
print(a[4]['segments'][0]['n_attempts'], b[4].get('last_segment', {}).get('n_attempts'))
d = (a[2][0] - b[2][0]).abs()
print('max err', float(d.max()), 'ref max', float(b[2][0].abs().max()))
blk = d.reshape(8, 16, 8, 16).amax(dim=(1, 3))
torch.set_printoptions(linewidth=200, precision=2, sci_mode=True)
print(blk)
print('row max', d.amax(1)[:16], d.amax(1)[112:])
