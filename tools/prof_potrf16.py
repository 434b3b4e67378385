import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096; batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = int(np.sqrt(n))
xs = torch.linspace(0, 1, g, device=dev)
X = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2).contiguous()
ls = torch.linspace(0.1, 0.5, batch, device=dev); var = torch.ones(batch, device=dev)
K = ops.gram_se(X, X, var, ls, jitter=1e-3)
A = K.clone()
for _ in range(3):
    A.copy_(K); ops.potrf_(A)
torch.cuda.synchronize()
