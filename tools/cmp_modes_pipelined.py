"""The bench's run (5 warm-up steps, then `steps` pipelined steps in groups of 16) under several draw modes:
per-step losses side by side.  (dev tool)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
modes = sys.argv[2:] or ["bf16x3", "f16x2"]
out = {}
for mode in modes:
    eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                     hyper_sampler=bench.bench_hyper, group=16, mfma=mode)
    st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(5)], st, train=True)
    eng.run([(1, i) for i in range(steps)], st, train=True)
    out[mode] = eng.loss_steps.cpu().numpy().astype(np.float64)
    print(mode, "info_fail", int(eng.info_fail.item()), "finite", bool(np.isfinite(out[mode]).all()))
for m in modes:
    l = np.log10(out[m])
    print(m, "mean log10 loss per 100 steps:", " ".join("%.3f" % l[i:i + 100].mean() for i in range(0, steps, 100)),
          " max %.3e at %d  last %.4e" % (out[m].max(), out[m].argmax(), out[m][-1]))
a, b = out[modes[0]], out[modes[-1]]
r = b / a
for i in np.argsort(-np.abs(np.log(r)))[:15]:
    print("step %d: %s %.4e  %s %.4e  ratio %.3f  hyper %s" % (i, modes[0], a[i], modes[-1], b[i], r[i],
          bench.bench_hyper(__import__("gp_vae_amd.parallel", fromlist=["x"]).step_rng(1, 0, (1, int(i))))))
