"""first 16 per-step losses, warm-up 0 / 5, bf16x3 against f16x2 (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
np.set_printoptions(linewidth=200, precision=5)
for warm in (0, 5):
    res = {}
    for mode in ("bf16x3", "f16x2"):
        eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                         hyper_sampler=bench.bench_hyper, group=16, mfma=mode)
        st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
        lw = None
        if warm:
            eng.run([(0, i) for i in range(warm)], st, train=True)
            lw = eng.loss_steps.cpu().numpy().astype(np.float64)
        eng.run([(1, i) for i in range(16)], st, train=True)
        res[mode] = (lw, eng.loss_steps.cpu().numpy().astype(np.float64))
    if warm:
        print("warm-up losses ratio h2/b3:", res["f16x2"][0] / res["bf16x3"][0])
    print("warm", warm, "ratio h2/b3 per step:", res["f16x2"][1] / res["bf16x3"][1])
