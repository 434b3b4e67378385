import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from gp_vae_amd import ops
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
def c2(tag):
    x1 = torch.arange(1024, dtype=torch.float32, device=dev) / 1024
    eng = StepEngine(x1, 1e-4, (35, 30, 10), 4096, 1e-3, dtype=torch.float32, seed=1, group=16)
    st = SVIState(init_params(1024, 35, 30, 10, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(32)], st, train=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run([(1, i) for i in range(256)], st, train=True)
    torch.cuda.synchronize()
    print(tag, (time.perf_counter() - t0) / 256 * 1e3, "ms/step", flush=True)
c2("fresh")
x = torch.tensor(bench.grid_points(), device=dev)
eng = StepEngine(x, 1e-3, (50, 30, 15), 4096, 1e-3, dtype=torch.float32, seed=1, hyper_sampler=bench.bench_hyper, group=16)
state = SVIState(init_params(4096, 50, 30, 15, 1, torch.float32, dev))
eng.run([(0, i) for i in range(100)], state, train=True); torch.cuda.synchronize()
c2("after C3 run")
bench.potrf_microbench(ops, x, batch=32)
c2("after potrf microbench")
time.sleep(2)
c2("after potrf microbench + 2 s sleep")
bench.potrf_microbench(ops, x, batch=32)
torch.cuda.empty_cache()
c2("after potrf microbench + empty_cache")
bench.potrf_microbench(ops, x, batch=32)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
c2("again after microbench")
