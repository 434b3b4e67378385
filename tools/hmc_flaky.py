import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import gpvae_oracle as orc
from gp_vae_amd.gp import exp_kernel2
from gp_vae_amd.inference import GPInference, VAEInference
for rep in range(6):
    rng = np.random.default_rng(2)
    n = 80
    x = np.arange(n) / n
    obs = np.sort(rng.permutation(n)[:20])
    truth = np.sin(6 * x)
    y = np.full(n, np.nan); y[obs] = truth[obs] + 0.05 * rng.standard_normal(20)
    args = {"num_warmup": 150, "num_samples": 100, "num_chains": 4, "thinning": 2}
    inf = GPInference(exp_kernel2, 1, x, y, obs, args, seed=5, dtype=torch.float64)
    means, preds = inf.gp_fit(plot=False)
    mask = np.ones(n, bool); mask[obs] = False
    err = np.sqrt(np.mean((means.mean(0).cpu().numpy() - truth[mask]) ** 2))
    nz, h1, h0 = 6, 9, 12
    p = orc.init_params(rng, n, h0, h1, nz, np.float64)
    dec = [(50 * p["V1"], p["c1"]), (), (50 * p["V2"], p["c2"]), (), (50 * p["V3"], p["c3"])]
    vinf = VAEInference(None, dec, nz, 1, x, y, obs, args, seed=6, dtype=torch.float64)
    prior, post = vinf.fit(plot=False, num_prior=64)
    yo = torch.tensor(y[obs], device=post.device)
    e_post = (post[:, obs].mean(0) - yo).abs().mean().item()
    e_prior = (prior[:, obs].mean(0) - yo).abs().mean().item()
    print(rep, "gp div", inf.mcmc_info["divergent"], "err", round(float(err), 4), "| vae div", vinf.mcmc_info["divergent"],
          "finite", bool(torch.isfinite(post).all()), "e_post", round(e_post, 4), "e_prior", round(e_prior, 4), flush=True)
