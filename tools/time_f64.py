"""Dev tool: f64 batched Cholesky timing (16 x 4096) and one 16384^2 factorisation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"
for n, batch in ((4096, 16), (16384, 1)):
    g = int(np.sqrt(n)); xs = torch.linspace(0, 1, g, device=dev, dtype=torch.float64)
    X = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2).contiguous()
    ls = torch.linspace(0.05, 0.3, batch, device=dev, dtype=torch.float64); var = torch.ones(batch, device=dev, dtype=torch.float64)
    K = ops.gram_se(X, X, var, ls, jitter=1e-3)
    A = K.clone(); ts = []
    for it in range(3):
        A.copy_(K); torch.cuda.synchronize(); t0 = time.perf_counter()
        info = ops.potrf_(A, zero_upper=False, nan_on_fail=False); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"f64 potrf batch {batch} n {n}: {t*1e3:.2f} ms  {batch*n**3/3/t/1e12:.1f} TF  pd={bool((info==0).all())}", flush=True)
n = 4096
E = ops.randn((4096, n), torch.float64, dev, seed=1); E2 = ops.randn((4096, n), torch.float64, dev, seed=2)
L = torch.tril(ops.randn((n, n), torch.float64, dev, seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev, dtype=torch.float64)
for _ in range(2): ops.draw(L, E, E2, sig)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ops.draw(L, E, E2, sig)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
print(f"f64 draw 4096^3: {t*1e3:.3f} ms  {n*n*4096/t/1e12:.1f} TF")
