"""C3 draw: f32-MFMA kernel vs the bf16 x 3 kernel, and the producers (split of L, Philox planes)."""
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
L3 = ops.split_b3(K, lower=True)
Y = torch.empty((B, n), device=dev)
sig = torch.tensor([0.05], device=dev)
out = {"draw_f32_ms": timed(lambda: ops.draw(K, E, E2, sig)),
       "draw_b3_ms": timed(lambda: ops.draw_b3(L3, E3, E2, sig, out=Y)),
       "draw_b3_noE2_ms": timed(lambda: ops.draw_b3(L3, E3, out=Y)),
       "split_L_ms": timed(lambda: ops.split_b3(K, lower=True, out=L3)),
       "randn_f32_ms": timed(lambda: ops.randn(None, out=E, seed=1, stream_id=1)),
       "randn_b3_ms": timed(lambda: ops.randn_b3(None, out=E3, seed=1, stream_id=1))}
out["f32_equiv_tflops_b3"] = n * n * B / out["draw_b3_ms"] / 1e9
print(json.dumps(out))
