"""C2 (1D n = 1024, B = 4096, VAE 1024-35-30-10, f32) training step alone, as bench.py's `c2` extra times it (dev tool)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x1 = torch.arange(1024, dtype=torch.float32, device=dev) / 1024
eng = StepEngine(x1, 1e-4, (35, 30, 10), 4096, bench.LR, dtype=torch.float32, seed=1, group=16)
st = SVIState(init_params(1024, 35, 30, 10, 1, torch.float32, dev))
eng.run([(0, i) for i in range(32)], st, train=True)
torch.cuda.synchronize()
time.sleep(2.0)
for rep in range(1, 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run([(rep, i) for i in range(256)], st, train=True)
    torch.cuda.synchronize()
    print("C2 ms/step %.4f" % ((time.perf_counter() - t0) / 256 * 1e3))
