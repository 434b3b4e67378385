import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gp_vae_amd import ops, parallel
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1, hyper_sampler=bench.bench_hyper, group=8)
state = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
for ep in range(2):
    for g0 in range(0, 48, 8):
        keys = [(ep, i) for i in range(g0, g0 + 8)]
        hy = [bench.bench_hyper(parallel.step_rng(1, 0, k)) for k in keys]
        l = eng.run(keys, state, train=True).item()
        torch.cuda.synchronize()
        print(ep, g0, "loss_sum %.4g" % l, "info", eng.info[0].cpu().tolist(), "ls", [round(h[0], 3) for h in hy],
              "Ynan", bool(torch.isnan(eng.Y).any()), "pnan", bool(torch.isnan(state.params).any()), "pmax", state.params.abs().max().item(), flush=True)
