"""C3 step rate under three hyper-samplers: the host NumPy one (a Generator per step key), a constant host lambda,
and the device sampler — is the host's sampling time on the step's critical path?  (dev tool)"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
for name, hs in (("host numpy", bench.bench_hyper), ("host constant", lambda rng: (0.2, 1.0, 0.1)), ("device", "device"),
                 ("host numpy", bench.bench_hyper), ("host constant", lambda rng: (0.2, 1.0, 0.1)), ("device", "device")):
    eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1, hyper_sampler=hs)
    st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(32)], st, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run([(1, i) for i in range(2000)], st, train=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-14s %.4f ms/step   host done enqueueing after %.4f ms/step" % (name, (t2 - t0) / 2000 * 1e3, (t1 - t0) / 2000 * 1e3), flush=True)
