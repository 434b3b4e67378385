"""(mode+fresh: empty the caching allocator before building that engine.)  Long C3 training run (default 20 000 steps, both draw modes): failed factorisations, non-finite losses,
steady-state rate.  (dev tool)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for mode in ([a for a in sys.argv[2:] if not a.startswith("--")] or ["bf16x3", "f16x2"]):
    if mode.endswith("+fresh"):  # drop the previous engine's cached blocks first: the new buffers come from hipMalloc
        mode = mode[:-6]
        eng = st = None
        torch.cuda.empty_cache()
    eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=7,
                     hyper_sampler=bench.bench_hyper, mfma=mode)
    st = SVIState(init_params(bench.N, *bench.DIMS, 7, torch.float32, dev))
    eng.run([(0, i) for i in range(32)], st, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = 0
    for ep in range(steps // 1000):
        eng.run([(1 + ep, i) for i in range(1000)], st, train=True)
        l = eng.loss_steps
        bad += int((~torch.isfinite(l)).sum().item())
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%s: %d steps in %.1f s = %.3f ms/step (%.2f M draws/s), failed factors %d, non-finite losses %d, params finite %s, last-epoch median log10 loss %.3f" %
          (mode, steps, t, t / steps * 1e3, bench.B * steps / t / 1e6, eng.failed_factors(), bad,
           bool(torch.isfinite(st.params).all()), float(torch.log10(l.clamp_min(1.0)).median().item())), flush=True)
