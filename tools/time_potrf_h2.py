"""gpvae_potrf_b3 against gpvae_potrf_h2 at the C3 shapes (batch 16 / 32 x 4096, float32): ms per call and
float-equivalent TFLOP/s (n^3/3 per matrix).  (dev tool)"""
import sys
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd import ops

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev, dtype=torch.float32)
n = x.shape[0]
for batch in (16, 32):
    ls = torch.linspace(0.05, 1.0, batch, device=dev)
    K = ops.gram_se(x, x, torch.ones(batch, device=dev), ls, bench.JITTER).reshape(batch, n, n)
    A = torch.empty_like(K)
    L3 = torch.empty((batch, 3, n, n), dtype=torch.bfloat16, device=dev)
    L2 = L3.view(torch.float16)[:, :2]
    inv = torch.empty(batch, device=dev)
    info = torch.empty(batch, dtype=torch.int32, device=dev)
    wb = torch.empty(ops._lib.load().gpvae_potrf_workspace_bytes(0, n, 0, batch), dtype=torch.uint8, device=dev)
    for name, f in (("potrf_b3", lambda: ops.potrf_b3_(A, L3, work=wb, info=info)),
                    ("potrf_h2", lambda: ops.potrf_h2_(A, L2, inv, work=wb, info=info))):
        ts = []
        for rep in range(6):
            A.copy_(K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts[1:])[len(ts[1:]) // 2]
        print("batch %d %s: %.3f ms  %.1f TFLOP/s (n^3/3)  ok %s" % (batch, name, t, batch * n ** 3 / 3 / t / 1e9, bool((info == 0).all())))
