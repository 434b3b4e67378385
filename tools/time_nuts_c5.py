"""C5 end to end: VAEInference.regression under the batched NUTS, 1024 chains, decoder 10-30-35-300 (synthetic
weights), 10 observations.  Wall time of warm-up + sampling and leapfrogs per second."""
import sys, os, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd.vae import init_params, decoder_list_from_flat
from gp_vae_amd.inference import VAEInference
dev = torch.device("cuda", 0)
n, dims = 300, (35, 30, 10)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dec = decoder_list_from_flat(init_params(n, *dims, 1, torch.float32, dev) * 3, n, *dims)
x = np.arange(0, 1, 1 / n)
obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]
y = np.full(n, np.nan); y[obs] = np.sin(6 * x[obs])
args = {"num_warmup": iters, "num_samples": iters, "num_chains": 1024, "thinning": 1, "max_tree_depth": 10}
inf = VAEInference(None, dec, dims[2], 1, x, y, obs, args, seed=1, dtype=torch.float32)
torch.cuda.synchronize(); t0 = time.time()
smp = inf.run_mcmc(3)
torch.cuda.synchronize(); t = time.time() - t0
info = inf.mcmc_info
lf = info["leapfrogs_per_iteration"].mean().item()
print(json.dumps({"config": "C5: VAEInference.run_mcmc, batched NUTS, 1024 chains, %d + %d iterations" % (iters, iters),
                  "wall_s": t, "accept_prob": info["accept_prob"].mean().item(), "divergent": info["divergent"],
                  "leapfrogs_per_iteration_sampling": lf, "step_size": info["step_size"].mean().item(),
                  "samples": list(smp["z"].shape)}))
