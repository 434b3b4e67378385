import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib
avail = os.environ.get("GPVAE_DEV_SYMS")
if avail:
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in avail.split(",")}
from gp_vae_amd import ops
dev = "cuda"; n = 4096
g = np.arange(0, 1, 1 / 64); u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1).astype(np.float32), device=dev)
K = torch.empty((n, n), device=dev)
var = torch.ones(1, device=dev); ls = torch.full((1,), 0.3, device=dev)
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
for _ in range(3):
    ops.gram_se(x, x, var, ls, 1e-3, out=K)
    ops.potrf_(K, zero_upper=False, nan_on_fail=False)
    ops.draw(K, E, E2, 0.1)
torch.cuda.synchronize()
nb = int(os.environ.get("NBATCH", "16"))
Kb = torch.empty((nb, n, n), device=dev)
lsb = torch.linspace(0.05, 1.0, nb, device=dev); vb = torch.ones(nb, device=dev)
for _ in range(2):
    ops.gram_se(x, x, vb, lsb, 1e-3, out=Kb)
    ops.potrf_(Kb, zero_upper=False, nan_on_fail=False)
torch.cuda.synchronize()
