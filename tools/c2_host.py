"""C1 / C2: is the step bound by the host's launch rate?  Host time to enqueue vs time to finish."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
out = {}
for tag, n, B, dims, steps in (("c1", 300, 100, (35, 30, 10), 512), ("c2", 1024, 4096, (35, 30, 10), 512)):
    x = torch.arange(n, dtype=torch.float32, device=dev) / n
    eng = StepEngine(x, 1e-4, dims, B, 1e-3, dtype=torch.float32, seed=1, group=16)
    st = SVIState(init_params(n, *dims, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(64)], st, train=True)
    torch.cuda.synchronize(); time.sleep(2.0)
    best = None
    for rep in (1, 2, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run([(rep, i) for i in range(steps)], st, train=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3)
        best = r if best is None or r[1] < best[1] else best
    out[tag] = {"host_enqueue_ms_per_step": best[0], "ms_per_step": best[1]}
print(json.dumps(out))
