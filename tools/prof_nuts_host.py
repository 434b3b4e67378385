import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, cProfile, pstats
from gp_vae_amd.vae import init_params, decoder_list_from_flat
from gp_vae_amd.inference import VAEInference
dev = torch.device("cuda", 0)
n, dims = 300, (35, 30, 10)
dec = decoder_list_from_flat(init_params(n, *dims, 1, torch.float32, dev) * 3, n, *dims)
x = np.arange(0, 1, 1 / n)
obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]
y = np.full(n, np.nan); y[obs] = np.sin(6 * x[obs])
for rep in range(3):
    args = {"num_warmup": 100, "num_samples": 100, "num_chains": 1024, "thinning": 1, "max_tree_depth": 10}
    inf = VAEInference(None, dec, dims[2], 1, x, y, obs, args, seed=1, dtype=torch.float32)
    torch.cuda.synchronize(); t0 = time.time()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    smp = inf.run_mcmc(3)
    torch.cuda.synchronize(); t = time.time() - t0
    if rep == 2:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
    print("rep", rep, "wall_s %.3f" % t, "leapfrogs/iter %.2f" % inf.mcmc_info["leapfrogs_per_iteration"].mean().item())
