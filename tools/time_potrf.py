"""Batched Cholesky timings (C3 micro-benchmark shapes) + backward error of the factors."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
N = 4096
g = np.arange(0, 1, 1 / 64)
u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1), dtype=torch.float32, device=dev)
out = {}
for batch, dt in ((32, torch.float32), (16, torch.float32), (1, torch.float32), (16, torch.float64)):
    xx = x.to(dt)
    ls = torch.linspace(0.05, 1.0, batch, device=dev, dtype=dt)
    var = torch.ones(batch, device=dev, dtype=dt)
    Kb = torch.empty((batch, N, N), device=dev, dtype=dt)
    ts = []
    for it in range(5):
        ops.gram_se(xx, xx, var, ls, 1e-3, out=Kb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); info = ops.potrf_(Kb, zero_upper=False, nan_on_fail=False); e1.record(); torch.cuda.synchronize()
        if it: ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts))
    K0 = ops.gram_se(xx, xx, var[:1], ls[:1], 1e-3).reshape(N, N).double()
    L = torch.tril(Kb[0]).double()
    be = (torch.linalg.norm(L @ L.T - K0) / torch.linalg.norm(K0)).item()
    out["b%d_%s" % (batch, "f32" if dt == torch.float32 else "f64")] = {"ms": t, "tflops": batch * N ** 3 / 3 / t / 1e9, "pd": bool((info == 0).all().item()), "backward_err": be}
print(json.dumps(out))
