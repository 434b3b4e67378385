"""Dev tool: standalone timings (no concurrent stream) of draw, batched potrf and its kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops, gp
dev = "cuda"; n = 4096
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
L = torch.tril(ops.randn((n, n), seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev)
F = torch.empty_like(E); Y = torch.empty_like(E)
ms = timeit(lambda: ops.draw(L, E, E2, sig))
print(f"draw 4096x4096x4096 f32: {ms:.3f} ms  {n*n*4096/ms/1e9:.1f} TF")
g = int(np.sqrt(n))
xs = torch.linspace(0, 1, g, device=dev)
X = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2).contiguous()
for batch in (1, 16, 32, 64):
    ls = torch.linspace(0.1, 0.5, batch, device=dev); var = torch.ones(batch, device=dev)
    K = ops.gram_se(X, X, var, ls, jitter=1e-3)
    A = K.clone()
    def f():
        A.copy_(K)
        ops.potrf_(A)
    def c():
        A.copy_(K)
    t = timeit(f, 5) - timeit(c, 5)
    print(f"potrf batch {batch} n {n}: {t:.3f} ms  {batch*n**3/3/t/1e9:.1f} TF")

# stand-alone solve against an existing factor (not on the hot paths)
A = ops.gram_se(X, X, 1.0, 0.3, jitter=1e-3); A.diagonal().add_(0.1)
ops.potrf_(A)
Xr = ops.randn((4096, n), seed=5); Xw = Xr.clone()
for tr in (True, False):
    def g():
        Xw.copy_(Xr); ops.trsm_(A, Xw, trans=tr)
    def c2():
        Xw.copy_(Xr)
    t = timeit(g, 5) - timeit(c2, 5)
    print(f"trsm trans={int(tr)} n {n} m 4096: {t:.3f} ms  {4096*n*n/t/1e9:.1f} TF")
