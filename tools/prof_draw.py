import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
L = torch.tril(ops.randn((n, n), seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev)
for _ in range(4):
    ops.draw(L, E, E2, sig)
torch.cuda.synchronize()
