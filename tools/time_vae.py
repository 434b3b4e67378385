"""Dev tool: standalone timing of the VAE ELBO fwd/bwd entry point at the C3 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096; B = 4096
h0, h1, z = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 30, 15)
P = ops.vae_param_count(n, h0, h1, z)
params = (ops.randn((P,), seed=5) * 0.05).contiguous()
Y = ops.randn((B, n), seed=1); eps = ops.randn((B, z), seed=2)
ws = ops.VaeWorkspace(B, n, h0, h1, z, torch.float32, dev)
grads = torch.empty_like(params)
def f():
    ops.vae_elbo(params, Y, eps, (h0, h1, z), 1.0, grads=grads, work=ws)
for _ in range(3): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"vae_elbo fwd+bwd B={B} n={n} dims=({h0},{h1},{z}): {e0.elapsed_time(e1)/20*1e3:.1f} us")
