"""Backward error of gpvae_potrf_h2 / gpvae_potrf_b3 on the C3 grid for chosen length scales and batch sizes (dev tool)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd import ops

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev, dtype=torch.float32)
n = x.shape[0]
for batch in (1, 2, 16):
    ls = torch.tensor(([1.1259, 4.4, 54.6, 0.2] * 4)[:batch], device=dev)
    var = torch.tensor(([1.25, 0.995, 1.07, 1.0] * 4)[:batch], device=dev)
    K = ops.gram_se(x, x, var, ls, bench.JITTER).reshape(batch, n, n)
    for name in ("b3", "h2"):
        A = K.clone()
        if name == "b3":
            info, P = ops.potrf_b3_(A)
        else:
            info, P, inv = ops.potrf_h2_(A)
        for b in sorted(set((0, batch // 2, batch - 1))):
            L = torch.tril(A[b]).double()
            err = ((L @ L.T - K[b].double()).abs().max() / K[b].diagonal().max()).item()
            print("batch %2d %s b=%2d ls=%.3g info=%d  max|LL^T-K|/maxdiag %.3e  max|L| %.3e  min diag L %.3e" %
                  (batch, name, b, ls[b].item(), info[b].item(), err, L.abs().max().item(), L.diagonal().min().item()))
        if name == "h2":
            for b in sorted(set((0, batch // 2, batch - 1))):
                L = torch.tril(A[b]).double()
                back = torch.tril(P[b].float().sum(0).double() * inv[b].double())
                E2h = ops.randn_h2((256, n), dev, seed=b, stream_id=2)
                E = ops.randn((256, n), torch.float32, dev, seed=b, stream_id=2)
                F = ops.draw_h2(P[b], inv[b:b + 1], E2h)
                Fo = E.double() @ L.T
                Ff = ops.draw(torch.tril(A[b]).contiguous(), E)                                  # f32 MFMA
                Fb = ops.draw_b3(ops.split_b3(A[b], lower=True), ops.randn_b3((256, n), dev, seed=b, stream_id=2))
                top = Fo.abs().max()
                print("   planes b=%d: max|planes/s - L| %.3e  inv %.4e  draw max err / max|F|: fp16x2 %.3e  bf16x3 %.3e  f32 MFMA %.3e" %
                      (b, (back - L).abs().max().item(), inv[b].item(), ((F.double() - Fo).abs().max() / top).item(),
                       ((Fb.double() - Fo).abs().max() / top).item(), ((Ff.double() - Fo).abs().max() / top).item()))
