import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
from oracle import gpvae_oracle as orc
dev = "cuda"
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
dt = np.float64
for n, nx in ((10, 291), (100, 526), (700, 801), (64, 300), (128, 257)):
    rng = np.random.default_rng(0)
    x = np.sort(rng.random(n)).astype(dt)
    K = orc.exp_kernel2(x, x, 1.0, 0.2, 1e-5) + 0.05 * np.eye(n)
    L = orc.cholesky_lower(K)
    Xh = rng.standard_normal((nx, n)).astype(dt)
    A = torch.tensor(K, device=dev); Xd = torch.tensor(Xh, device=dev)
    info = ops.potrf_(A, Xd)
    Xo = orc.right_solve_lt(L, Xh)
    Xg = Xd.cpu().numpy()
    rowerr = np.linalg.norm(Xg - Xo, axis=1) / np.linalg.norm(Xo, axis=1)
    bad = np.where(rowerr > 1e-8)[0]
    print(n, nx, "L rel", rel(np.tril(A.cpu().numpy()), L), "X rel", rel(Xg, Xo), "bad rows", bad[:10], len(bad), flush=True)
