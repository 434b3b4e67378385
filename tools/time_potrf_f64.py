"""Batched Cholesky on the f64 / f32 MFMA trailing update (not the bf16 x 3 route): f64 16 x 4096, f32 with extra rows."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
out = {}
for N, batch, dt, mx in ((4096, 16, torch.float64, 0), (2048, 16, torch.float64, 0), (1024, 16, torch.float64, 0), (4000, 16, torch.float32, 0),
                         (4096, 4, torch.float32, 512), (8192, 2, torch.float64, 0)):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand((N, 2), generator=g, dtype=torch.float64).to(dev).to(dt)
    ls = torch.linspace(0.05, 1.0, batch, device=dev, dtype=dt)
    var = torch.ones(batch, device=dev, dtype=dt)
    Kb = torch.empty((batch, N, N), device=dev, dtype=dt)
    X0 = torch.randn((batch, mx, N), device=dev, dtype=dt) if mx else None
    ts = []
    for it in range(5):
        ops.gram_se(x, x, var, ls, 1e-3, out=Kb)
        X = X0.clone() if mx else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); info = ops.potrf_(Kb, X, zero_upper=False, nan_on_fail=False) if mx else ops.potrf_(Kb, zero_upper=False, nan_on_fail=False); e1.record(); torch.cuda.synchronize()
        if it: ts.append(e0.elapsed_time(e1))
    K0 = ops.gram_se(x, x, var[:1], ls[:1], 1e-3).reshape(N, N).double()
    L = torch.tril(Kb[0]).double()
    be = (torch.linalg.norm(L @ L.T - K0) / torch.linalg.norm(K0)).item()
    r = {"ms": round(float(np.median(ts)), 4), "tflops": round(batch * N ** 3 / 3 / float(np.median(ts)) / 1e9, 1), "pd": bool((info == 0).all().item()), "bw": be}
    if mx:
        r["solve_err"] = (torch.linalg.norm(X[0].double() @ L.T - X0[0].double()) / torch.linalg.norm(X0[0].double())).item()
    out["n%d_b%d_%s_mx%d" % (N, batch, "f64" if dt == torch.float64 else "f32", mx)] = r
print(json.dumps(out))
