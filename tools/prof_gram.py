"""6 launches of the Gram builder at the training factor group's shape (16 x 4096 x 4096 f32) for PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"
g = np.arange(0, 1, 1 / 64)
u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1), dtype=torch.float32, device=dev)
ls = torch.linspace(0.05, 1.0, 16, device=dev); var = torch.ones(16, device=dev)
K = torch.empty((16, 4096, 4096), device=dev)
for _ in range(6):
    ops.gram_se(x, x, var, ls, 1e-3, out=K)
torch.cuda.synchronize()
