"""Batched f32 Cholesky through plain gpvae_potrf at the smaller bf16 x 3 sizes (C2: n = 1024)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
out = {}
for N, batch in ((512, 16), (768, 16), (1024, 16), (1024, 64), (1280, 16), (2048, 16), (4096, 16), (4096, 32)):
    x = torch.linspace(0, 1, N, device=dev, dtype=torch.float32)
    ls = torch.linspace(0.05, 1.0, batch, device=dev)
    var = torch.ones(batch, device=dev)
    Kb = torch.empty((batch, N, N), device=dev)
    ts = []
    for it in range(6):
        ops.gram_se(x, x, var, ls, 1e-3, out=Kb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); info = ops.potrf_(Kb, zero_upper=False, nan_on_fail=False); e1.record(); torch.cuda.synchronize()
        if it: ts.append(e0.elapsed_time(e1))
    K0 = ops.gram_se(x, x, var[:1], ls[:1], 1e-3).reshape(N, N).double()
    L = torch.tril(Kb[0]).double()
    be = (torch.linalg.norm(L @ L.T - K0) / torch.linalg.norm(K0)).item()
    out["n%d_b%d" % (N, batch)] = {"ms": round(float(np.median(ts)), 4), "pd": bool((info == 0).all().item()), "bw": be}
print(json.dumps(out))
