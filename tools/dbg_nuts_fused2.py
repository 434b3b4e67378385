import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd.inference import VAEInference
from oracle import gpvae_oracle as orc
tdt = torch.float64
dev = lambda a: torch.tensor(np.asarray(a), dtype=tdt, device="cuda")
rng = np.random.default_rng(8)
n, dims = 300, (35, 30, 10)
p = {k: 18 * v for k, v in orc.init_params(np.random.default_rng(2), n, *dims, np.float64).items()}
dec = [(dev(p["V1"]), dev(p["c1"])), (), (dev(p["V2"]), dev(p["c2"])), (), (dev(p["V3"]), dev(p["c3"]))]
x = np.arange(n) / n
obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]
y = np.full(n, np.nan)
y[obs] = orc.decoder(p, rng.standard_normal((1, dims[2])))[0][obs] + 0.05 * rng.standard_normal(len(obs))
for warm, ns in ((0, 6), (3, 3), (12, 3)):
    res = {}
    for driver in ("device", "lockstep"):
        args = {"num_warmup": warm, "num_samples": ns, "num_chains": 13, "thinning": 1, "max_tree_depth": 6, "nuts_driver": driver}
        inf = VAEInference(None, dec, dims[2], 1, x, y, obs, args, seed=3, dtype=tdt)
        pot = inf.regression(y=inf.y, obs_idx=inf.obs_idx)
        smp, info = inf._mcmc(11, pot)
        res[driver] = (smp.reshape(13, ns, -1), info)
    a, b = res["device"], res["lockstep"]
    print("warm", warm, "per-sample max diff", (a[0] - b[0]).abs().amax(dim=(0, 2)).tolist())
    print("   leap", a[1]["leapfrogs_per_iteration"].tolist()[:6], b[1]["leapfrogs_per_iteration"].tolist()[:6])
    print("   eps", a[1]["step_size"].tolist()[:4], b[1]["step_size"].tolist()[:4])
