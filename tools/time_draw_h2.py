"""C3 draw: bf16 x 3 against fp16 x 2 (time, error against a float64 product on sampled rows)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
n = B = 4096
def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = np.arange(0, 1, 1 / 64)
u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1), dtype=torch.float32, device=dev)
K = ops.gram_se(x, x, 1.0, 0.2, 1e-3)
ops.potrf_(K, zero_upper=False)
E = ops.randn((B, n), torch.float32, dev, seed=1, stream_id=1)
E2 = ops.randn((B, n), torch.float32, dev, seed=1, stream_id=2)
E3 = ops.randn_b3((B, n), dev, seed=1, stream_id=1)
Eh = ops.randn_h2((B, n), dev, seed=1, stream_id=1)
L3 = ops.split_b3(K, lower=True)
L2, inv = ops.split_h2(K, 1.01, lower=True)
Y = torch.empty((B, n), device=dev)
sig = torch.tensor([0.05], device=dev)
out = {"draw_b3_ms": timed(lambda: ops.draw_b3(L3, E3, E2, sig, out=Y)),
       "draw_h2_ms": timed(lambda: ops.draw_h2(L2, inv, Eh, E2, sig, out=Y)),
       "split_h2_ms": timed(lambda: ops.split_h2(K, 1.01, lower=True, out=L2, inv_scale=inv)),
       "randn_h2_ms": timed(lambda: ops.randn_h2((B, n), dev, seed=1, stream_id=1, out=Eh)),
       "randn_b3_ms": timed(lambda: ops.randn_b3((B, n), dev, seed=1, stream_id=1, out=E3))}
rows = torch.tensor(np.random.default_rng(0).integers(0, B, 64), device=dev)
want = E[rows].double() @ torch.tril(K).double().T + 0.05 * E2[rows].double()
for tag, y in (("b3", ops.draw_b3(L3, E3, E2, sig)), ("h2", ops.draw_h2(L2, inv, Eh, E2, sig)), ("f32", ops.draw(K, E, E2, 0.05))):
    out["rel_err_" + tag] = (torch.linalg.norm(y[rows].double() - want) / torch.linalg.norm(want)).item()
    out["max_err_" + tag] = ((y[rows].double() - want).abs().max() / want.abs().max()).item()
out["f32_equiv_tflops_h2"] = n * n * B / out["draw_h2_ms"] / 1e9
print(json.dumps(out))
