"""Dev tool: stand-alone timing of the batched Gram kernel (HBM-write bound) at the C3 shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096
g = int(np.sqrt(n)); xs = torch.arange(g, device=dev, dtype=torch.float32) / g
X = torch.stack(torch.meshgrid(xs, xs, indexing="xy"), -1).reshape(-1, 2).contiguous()
for dt in (torch.float32, torch.float64):
    for batch in (1, 16):
        ls = torch.linspace(0.1, 0.5, batch, device=dev, dtype=dt); var = torch.ones(batch, device=dev, dtype=dt)
        K = torch.empty((batch, n, n), dtype=dt, device=dev)
        Xd = X.to(dt)
        f = lambda: ops.gram_se(Xd, Xd, var, ls, jitter=1e-3, out=K)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"gram {dt} batch {batch}: {t*1e6:.1f} us  {K.numel()*K.element_size()/t/1e12:.2f} TB/s written")
