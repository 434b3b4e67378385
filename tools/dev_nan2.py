import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gp_vae_amd import ops, parallel
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
G = int(sys.argv[1]); W = int(sys.argv[2]); K = int(sys.argv[3])
eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1, hyper_sampler=bench.bench_hyper, group=G)
state = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
losses = []
orig = eng._consume
def hooked(key, slot, i, last, st, train):
    orig(key, slot, i, last, st, train)
    losses.append((key, eng.loss.clone(), torch.isnan(eng.Y).any(), eng.info[slot].clone(), torch.isnan(eng.Kbuf[slot][i]).any()))
eng._consume = hooked
eng.run([(0, i) for i in range(W)], state, train=True)
eng.run([(1, i) for i in range(K)], state, train=True)
torch.cuda.synchronize()
for key, l, yn, info, kn in losses:
    print(key, "%.4g" % l.item(), "Ynan", bool(yn), "info", info.cpu().tolist(), "Lnan", bool(kn), bench.bench_hyper(parallel.step_rng(1, 0, key)))
