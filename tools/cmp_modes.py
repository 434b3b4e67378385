"""Per-step loss of the C3 training run under two draw modes (same seed): where do they part?  (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = {}
for mode in sys.argv[2:] or ["bf16x3", "f16x2"]:
    eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                     hyper_sampler=bench.bench_hyper, group=16, mfma=mode)
    st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
    losses = []
    for i in range(steps):
        eng.run([(1, i)], st, train=True)
        losses.append(float(eng.loss.item()))
    out[mode] = np.array(losses)
    print(mode, "info_fail", int(eng.info_fail.item()))
modes = list(out)
a, b = out[modes[0]], out[modes[1]]
r = np.abs(a - b) / np.abs(a)
for i in range(steps):
    if r[i] > 1e-3 or i % 50 == 0:
        print(i, a[i], b[i], r[i])
for m in modes:
    l = np.log10(out[m])
    print(m, "mean log10 loss per 100 steps:", " ".join("%.3f" % l[i:i + 100].mean() for i in range(0, steps, 100)), " last", out[m][-1])
