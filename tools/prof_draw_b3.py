"""12 back-to-back launches of the C3 bf16 x 3 draw (for rocprofv3 --pmc / --kernel-trace passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096
E3 = ops.randn_b3((4096, n), dev, seed=1); E2 = ops.randn((4096, n), seed=2)
L3 = ops.split_b3(torch.tril(ops.randn((n, n), seed=3)).contiguous(), lower=True)
sig = torch.full((1,), 0.1, device=dev)
Y = torch.empty((4096, n), device=dev)
for _ in range(12):
    ops.draw_b3(L3, E3, E2, sig, out=Y)
torch.cuda.synchronize()
