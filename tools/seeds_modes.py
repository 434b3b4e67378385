"""C3 training, 5 warm-up + 300 pipelined steps, several seeds and draw modes: how often does a run reach the
regime where the ELBO explodes on smooth (large length-scale) draws?  (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
summary = {}
for seed in range(1, int(sys.argv[2]) + 1 if len(sys.argv) > 2 else 9):
    row = []
    for mode in (sys.argv[3:] or ["f32", "bf16x3", "f16x2"]):
        eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=seed,
                         hyper_sampler=bench.bench_hyper, group=16, mfma=mode)
        st = SVIState(init_params(bench.N, *bench.DIMS, seed, torch.float32, dev))
        eng.run([(0, i) for i in range(5)], st, train=True)
        eng.run([(1, i) for i in range(steps)], st, train=True)
        l = eng.loss_steps.cpu().numpy().astype(np.float64)
        ll = np.log10(np.maximum(l, 1.0))
        summary.setdefault(mode, []).append((np.median(ll), l.max(), int((l > 1e11).sum())))
        row.append("%s: mean log10 %.3f max %.2e n(>1e11) %d" % (mode, ll.mean(), l.max(), int((l > 1e11).sum())))
    print("seed %d  " % seed + "  |  ".join(row), flush=True)
for mode, v in summary.items():
    v = np.array(v)
    print("%s: over %d seeds  median of (median log10 loss) %.3f   runs with a loss > 1e11: %d   steps with a loss > 1e11: %d   median max loss %.2e" %
          (mode, len(v), np.median(v[:, 0]), int((v[:, 2] > 0).sum()), int(v[:, 2].sum()), np.median(v[:, 1])))
