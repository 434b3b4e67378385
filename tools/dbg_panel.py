import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
for n in (512, 1024):
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.random((n, 2)), dtype=torch.float32, device=dev)
    K = ops.gram_se(x, x, 1.0, 0.3, 1e-2).reshape(1, n, n)
    A = K.clone()
    info = ops.potrf_(A, zero_upper=True, nan_on_fail=False)
    L = np.linalg.cholesky(K[0].cpu().numpy().astype(np.float64))
    G = A[0].cpu().numpy().astype(np.float64)
    print("n", n, "info", info.tolist())
    nb = n // 64
    E = np.zeros((nb, nb))
    for i in range(nb):
        for j in range(i + 1):
            d = G[64*i:64*i+64, 64*j:64*j+64] - L[64*i:64*i+64, 64*j:64*j+64]
            E[i, j] = np.nanmax(np.abs(d)) if np.isfinite(d).any() else np.nan
    np.set_printoptions(linewidth=250, precision=1)
    print(E[:8, :8])
