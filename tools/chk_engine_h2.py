"""After a pipelined f16x2 run: do the fp16 planes of the last two factor groups match the f32 factors in Kbuf? (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                 hyper_sampler=bench.bench_hyper, group=16, mfma="f16x2")
st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if warm:
    eng.run([(0, i) for i in range(warm)], st, train=True)
eng.run([(1, i) for i in range(steps)], st, train=True)
torch.cuda.synchronize()
print("losses", eng.loss_steps.cpu().numpy())
n = bench.N
for slot in (0, 1):
    K, P, inv, hd = eng.Kbuf[slot], eng.L2buf[slot], eng.L2inv[slot], eng.hyper_dev[slot]
    for i in range(16):
        L = torch.tril(K[i][:, :n])
        back = torch.tril(P[i].float().sum(0) * inv[i])
        d = (back - L).abs()
        print("slot %d i=%2d ls=%.3g var=%.3g inv=%.4e  max|planes - L| %.3e at %s  max|L| %.3e  planes finite %s" %
              (slot, i, hd[1, i].item(), hd[0, i].item(), inv[i].item(), d.max().item(),
               np.unravel_index(int(d.argmax().item()), (n, n)), L.abs().max().item(), bool(torch.isfinite(P[i].float()).all())))
from gp_vae_amd import ops
for slot, i in ((1, 9), (1, 8), (1, 5), (0, 0), (1, 0)):
    L = torch.tril(eng.Kbuf[slot][i][:, :n]).double()
    E2h = ops.randn_h2((256, n), dev, seed=7, stream_id=2)
    E = ops.randn((256, n), torch.float32, dev, seed=7, stream_id=2)
    F = ops.draw_h2(eng.L2buf[slot][i], eng.L2inv[slot][i:i + 1], E2h)
    Fo = E.double() @ L.T
    up = torch.triu(eng.L2buf[slot][i].float().abs().sum(0), 1)
    print("slot %d i=%d draw max err %.3e max|F| %.3e  upper max %.3e nonzeros in upper %d" %
          (slot, i, (F.double() - Fo).abs().max().item(), Fo.abs().max().item(), up.max().item(), int((up != 0).sum().item())))
    blk = up.reshape(n // 64, 64, n // 64, 64).amax(dim=(1, 3))
    bad = torch.nonzero(blk)
    print("   nonzero 64x64 upper blocks (first 12):", bad[:12].tolist(), " count", bad.shape[0])
