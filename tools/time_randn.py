"""Dev tool: timing of gpvae_randn at the C3 size (its accuracy is tests/test_gpu_parity.py::test_randn)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from gp_vae_amd import ops
n = 4096 * 4096
out = torch.empty(n, device="cuda")
f = lambda: ops.randn(None, out=out, seed=7, stream_id=3)
for _ in range(3): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print(f"randn f32 {n} normals: {t*1e6:.1f} us  {n/t/1e12:.2f} T normals/s  {4*n/t/1e12:.2f} TB/s written")
