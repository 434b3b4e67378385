"""gpvae_trsm at m = n = 4096 f32 (both directions): ms per call; run under rocprofv3 --kernel-trace --stats for
the split between the diagonal solves and the trailing updates.  (dev tool)"""
import sys
import torch
sys.path.insert(0, ".")
from gp_vae_amd import ops
n = m = 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
A = torch.randn((n, n), device="cuda", generator=g)
K = A @ A.T / n + torch.eye(n, device="cuda")
ops.potrf_(K)
X0 = torch.randn((m, n), device="cuda", generator=g)
for tr in (True, False):
    X = X0.clone(); ops.trsm_(K, X, trans=tr); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    X = X0.clone(); e0.record(); ops.trsm_(K, X, trans=tr); e1.record(); torch.cuda.synchronize()
    print("trsm trans=%s: %.3f ms  %.1f TFLOP/s" % (tr, e0.elapsed_time(e1), m * n * n / e0.elapsed_time(e1) / 1e9))
