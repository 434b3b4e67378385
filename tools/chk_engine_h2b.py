"""f16x2 engine, 5 warm-up + 64 steps: which steps' losses are off, with and without stream overlap? (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
res = {}
for mode, overlap, warm in (("bf16x3", True, 5), ("f16x2", True, 5), ("f16x2", False, 5), ("f16x2", True, 16), ("f16x2", True, 1)):
    eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                     hyper_sampler=bench.bench_hyper, group=16, mfma=mode, overlap=overlap)
    st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(warm)], st, train=True)
    eng.run([(1, i) for i in range(64)], st, train=True)
    l = eng.loss_steps.cpu().numpy().astype(np.float64)
    res[(mode, overlap, warm)] = l
    ref = res[("bf16x3", True, 5)]
    print(mode, overlap, warm, "steps off by > 30 %:", [int(i) for i in np.nonzero(np.abs(np.log(l / ref)) > 0.3)[0]])
