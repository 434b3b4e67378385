"""1-D GP-prior VAE, end to end, with the reference's own call sequence
(/root/reference/src/test/toy_examples.py:30-200, 596-667; BASELINE.json configs[0]):

    gp = GP(kernel=exp_kernel2, jitter)                       # toy_examples.py:233
    vae = VAE(gp, hidden_dims, z_dim, n, batch_size, lr, num_epochs, num_train, num_test, x, seed)
    decoder_params = vae.fit()                                # toy_examples.py:150-160
    inference = VAEInference(decoder, decoder_params, z_dim, d, x, y, obs_idx, mcmc_args, seed)
    prior_pred, post_pred = inference.fit(plot=False)         # toy_examples.py:180-200

Everything numerical runs in libgpvae_hip (MI355X); NUTS is the batched restatement in gp_vae_amd/nuts.py.
usage: python examples/toy_1d.py [--num-epochs 5 --num-train 200 --num-test 50 --noise 0.001]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_vae_amd import ops  # noqa: E402
from gp_vae_amd.gp import GP, Predictive, exp_kernel2  # noqa: E402
from gp_vae_amd.inference import GPInference, VAEInference  # noqa: E402
from gp_vae_amd.vae import VAE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)                      # toy_examples.py:615
    ap.add_argument("--hidden-dims", type=int, nargs=2, default=[35, 30])
    ap.add_argument("--z-dim", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--learning-rate", type=float, default=1e-3)
    ap.add_argument("--num-epochs", type=int, default=5)               # reference default 70
    ap.add_argument("--num-train", type=int, default=200)              # reference default 1000
    ap.add_argument("--num-test", type=int, default=50)
    ap.add_argument("--jitter", type=float, default=1e-4)
    ap.add_argument("--noise", type=float, default=0.001)              # dat_noise of vae_mcmc_context
    ap.add_argument("--ls", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()

    x = np.arange(0, 1, 1 / a.n)
    gp = GP(kernel=exp_kernel2, jitter=a.jitter)
    t0 = time.time()
    vae = VAE(gp, a.hidden_dims, a.z_dim, a.n, a.batch_size, a.learning_rate, a.num_epochs, a.num_train,
              a.num_test, x, seed=a.seed)
    decoder_params = vae.fit()
    torch.cuda.synchronize()
    t_fit = time.time() - t0
    steps = a.num_epochs * (a.num_train + a.num_test)
    print("VAE.fit: %d epochs, test loss per sample by epoch: %s" % (a.num_epochs, ["%.1f" % v for v in vae.test_loss_list]))
    print("         %.2f s, %.0f draws/s incl. evaluation" % (t_fit, steps * a.batch_size / t_fit))
    # the reference pickles the decoder here (toy_examples.py:163-165) and reloads it for inference
    import os
    import tempfile
    from gp_vae_amd.vae import load_decoder_params, save_decoder_params
    with tempfile.TemporaryDirectory() as tmp:
        save_decoder_params(os.path.join(tmp, "decoder.npz"), decoder_params)
        decoder_params = load_decoder_params(os.path.join(tmp, "decoder.npz"))

    # ground truth: one GP draw with ls = 0.05, var = 1, observed with noise at 10 points
    draw = Predictive(gp.sample, num_samples=1)(a.seed + 7, x, ls=a.ls, var=1.0, sigma=a.noise)
    y_true = draw["y"][0].cpu().numpy()
    obs_idx = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]          # toy_examples.py:666
    y = np.full(a.n, np.nan)
    y[obs_idx] = y_true[obs_idx]
    mcmc_args = {"num_samples": 400, "num_warmup": 400, "num_chains": 4, "thinning": 2}
    _, decoder = vae.vae_decoder()
    t0 = time.time()
    inf = VAEInference(decoder, decoder_params, a.z_dim, 1, x, y, obs_idx, mcmc_args, seed=a.seed)
    prior_pred, post_pred = inf.fit(plot=False)
    torch.cuda.synchronize()
    t_mcmc = time.time() - t0
    post_mean = post_pred.mean(0).cpu().numpy()
    rmse_obs = float(np.sqrt(np.mean((post_mean[obs_idx] - y_true[obs_idx]) ** 2)))
    rmse_all = float(np.sqrt(np.mean((post_mean - y_true) ** 2)))
    print("VAEInference.fit (NUTS, 4 chains): %.2f s, accept %.2f, divergent %d" %
          (t_mcmc, inf.mcmc_info["accept_prob"].mean().item(), inf.mcmc_info["divergent"]))
    print("         posterior-mean RMSE at the 10 observed points %.3f, over all %d points %.3f (prior draws: %.3f)" %
          (rmse_obs, a.n, rmse_all, float(np.sqrt(np.mean((prior_pred.mean(0).cpu().numpy() - y_true) ** 2)))))

    # the GP itself as the yardstick (gp_krig of the reference): kriging with the true hyper-parameters
    ginf = GPInference(exp_kernel2, 1, x, y, obs_idx, mcmc_args, seed=a.seed)
    mean, _ = ginf.gp_prediction(a.seed, ginf.x, ginf.y, 1.0, a.ls, max(a.noise, 1e-6))
    mask = np.ones(a.n, bool); mask[obs_idx] = False
    print("GP kriging with the true hyper-parameters: RMSE at the %d unobserved points %.3f" %
          (int(mask.sum()), float(np.sqrt(np.mean((mean.cpu().numpy() - y_true[mask]) ** 2)))))


if __name__ == "__main__":
    main()
