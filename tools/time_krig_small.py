"""f2 timing: gpvae_krig_predict_batched, S problems, both routes, several (n_obs, n_test)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = torch.device("cuda", 0)
g = np.arange(25) / 25.0
u, v = np.meshgrid(g, g)
x = np.stack([u.ravel(), v.ravel()], 1).astype(np.float32)
rng = np.random.default_rng(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for n_obs, n_test in ((100, 525), (100, 32), (100, 128), (64, 525), (32, 525), (10, 615), (160, 465)):
    obs = np.sort(rng.permutation(625)[:n_obs])
    mask = np.ones(625, bool); mask[obs] = False
    y = np.sin(4 * x.sum(1)) + 0.1 * rng.standard_normal(625)
    var = rng.lognormal(0, 0.1, S); ls = rng.uniform(0.05, 0.5, S); sig = np.abs(rng.normal(0, 0.1, S)) + 1e-3
    xo = torch.tensor(x[obs], device=dev); xt = torch.tensor(x[mask][:n_test], device=dev)
    yo = torch.tensor(y[obs], dtype=torch.float32, device=dev)
    xi = ops.randn((S, n_test), torch.float32, dev, seed=9)
    res = {}
    for big in (False, True):
        ops.krig_set_path(big)
        res["big" if big else "lds"] = timed(lambda: ops.krig_predict_batched(xo, xt, yo, var, ls, sig, 1e-5, xi))
    ops.krig_set_path(False)
    print(json.dumps({"n_obs": n_obs, "n_test": n_test, "S": S, "lds_ms": res["lds"] * 1e3, "big_ms": res["big"] * 1e3,
                      "lds_Mproblems_per_s": S / res["lds"] / 1e6, "big_Mproblems_per_s": S / res["big"] / 1e6}), flush=True)
