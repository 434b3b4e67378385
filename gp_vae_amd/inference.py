"""Inference drivers — mirror /root/reference/src/model/inference.py on top of libgpvae_hip.

What is built natively here is the arithmetic NUTS and the kriging post-processing call on every
evaluation:
  * GPInference.gp_prediction            -> gpvae_krig_predict (one problem, any size)
  * GPInference.gp_fit_from_samples      -> the vmap over posterior samples (inference.py:511-521),
                                            sharded over ranks with no collective
  * VAEInference.potential_and_grad      -> gpvae_decoder_fwd / gpvae_decoder_vjp batched over chains
                                            (what NUTS asks of VAEInference.regression per leapfrog)
The NUTS transition itself is NumPyro's (third party, not in the reference tree): `run_mcmc` raises
unless numpyro is importable.  Plotting (inference.py:149-204) is out of scope.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .hmc import hmc
from ._lib import GpvaeError
from .gp import _as_device, _seed_of, exp_kernel2
from .parallel import shard
from .vae import decoder_flat_from_list


class Inference():
    """Constructor and attributes of the reference base class (inference.py:35-59)."""

    def __init__(self, d, x, y, obs_idx, mcmc_args, seed=0, ground_truth=None, grid=True,
                 dtype=torch.float32):
        self.d = d
        self.x = _as_device(x, dtype)
        self.y = _as_device(y, dtype)
        self.obs_idx = torch.as_tensor(np.asarray(obs_idx), dtype=torch.long, device=self.x.device)
        self.mcmc_args = mcmc_args
        self.seed = seed
        self.ground_truth = ground_truth
        self.grid = grid
        self.dtype = dtype
        if grid:
            self.n = int(round(self.x.shape[0] ** (1 / d)))  # inference.py:57
        else:
            self.n = self.x.shape[0]

    def regression(self, y=None, obs_idx=None):
        raise NotImplementedError  # inference.py:63-66

    def _mcmc(self, rng_key, potential, q0):
        """mcmc_args of the reference (num_warmup, num_samples, num_chains, thinning;
        inference.py:86-93) driving the batched HMC of gp_vae_amd.hmc (NUTS is third-party code
        that is absent here — same target, different trajectory rule).  q0: [dim] start, jittered
        per chain.  Returns samples [num_samples // thinning * num_chains, dim] (chains
        concatenated, as mcmc.get_samples() does) and the sampler's info dict."""
        a = self.mcmc_args
        chains = int(a.get("num_chains", 1))
        gen = torch.Generator(device=q0.device)
        gen.manual_seed(_seed_of(rng_key))
        start = q0[None, :] + 0.01 * torch.randn((chains, q0.numel()), dtype=q0.dtype, device=q0.device,
                                                  generator=gen)
        smp, info = hmc(potential, start, int(a["num_warmup"]), int(a["num_samples"]),
                        thinning=int(a.get("thinning", 1)), num_leapfrog=int(a.get("num_leapfrog", 16)),
                        seed=_seed_of(rng_key) + 1)
        self.mcmc_info = info
        return smp.permute(1, 0, 2).reshape(-1, q0.numel()), info

    def run_mcmc(self, rng_key):
        raise NotImplementedError  # the subclasses know their model (inference.py:69-99)


class GPInference(Inference):
    """Kriging with a GP prior (inference.py:338-549)."""

    def __init__(self, kernel, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if kernel is not exp_kernel2:
            raise NotImplementedError("GPInference is built for exp_kernel2")
        self.kernel = kernel

    def _split(self):
        n = self.x.shape[0]
        mask = torch.ones(n, dtype=torch.bool, device=self.x.device)
        mask[self.obs_idx] = False
        test_idx = torch.nonzero(mask).reshape(-1)  # jnp.delete(x, obs_idx, axis=0)  inference.py:442
        return self.x[self.obs_idx].contiguous(), self.x[test_idx].contiguous()

    def gp_prediction(self, rng_key, x, y, var, ls, sigma, return_var=False):
        """Posterior mean and one posterior sample at the unobserved locations (inference.py:427-460).
        `rng_key` seeds the standard normals of the sample (Philox stream)."""
        if x is not self.x:
            self.x = _as_device(x, self.dtype)
        y = _as_device(y, self.dtype)
        x_obs, x_test = self._split()
        xi = ops.randn((x_test.shape[0],), self.dtype, self.x.device, seed=_seed_of(rng_key),
                       stream_id=0xC41)
        mean, sample, var_out, info = ops.krig_predict(x_obs, x_test, y[self.obs_idx], var, ls,
                                                       sigma, 1.0e-5, xi)
        self.last_info = info
        return (mean, sample, var_out) if return_var else (mean, sample)

    def potential_and_grad(self, f, log_var, log_ls, log_sigma):
        """Potential energy of gp_regression (inference.py:400-424) in NumPyro's unconstrained space
        and its gradient, for a batch of chains — what NUTS evaluates per leapfrog step.

        sites: kernel_var ~ LogNormal(0,0.1), kernel_sigma ~ HalfNormal(0.1),
               kernel_length ~ InverseGamma(4,1), f ~ MVN(0, kernel(x_obs, x_obs, var, ls)),
               y_pred ~ Normal(f, sigma) observed at x_obs.
        f: [chains, n_obs]; log_var, log_ls, log_sigma: [chains].
        Returns (U [chains], dU/df [chains, n_obs], dU/dlog_var, dU/dlog_ls, dU/dlog_sigma).
        The MVN term and its gradient are one gpvae_mvn_logp call; the priors and the Gaussian
        likelihood are O(chains * n_obs) glue."""
        x_obs = self.x[self.obs_idx].contiguous()
        y_obs = self.y[self.obs_idx]
        n = x_obs.shape[0]
        f = _as_device(f, self.dtype).reshape(-1, n).contiguous()
        uv, ul, us = (_as_device(t, self.dtype).reshape(-1) for t in (log_var, log_ls, log_sigma))
        var, ls, sig = torch.exp(uv), torch.exp(ul), torch.exp(us)
        lp, gf, gv, gl, info = ops.mvn_logp(x_obs, f, var, ls, 1.0e-5)
        self.last_info = info
        s0, hl2p = 0.1, 0.5 * math.log(2 * math.pi)
        r = (y_obs[None, :] - f) / sig[:, None]
        sq = (r * r).sum(1)
        logp = (-math.log(s0) - hl2p - uv ** 2 / (2 * s0 ** 2)) \
            + (math.log(2.0) - hl2p - math.log(s0) - 0.5 * (sig / s0) ** 2 + us) \
            + (-math.lgamma(4.0) - 4.0 * ul - 1.0 / ls) \
            + lp + (-0.5 * sq - n * us - n * hl2p)
        d_f = gf + r / sig[:, None]
        d_uv = -uv / s0 ** 2 + gv * var
        d_ul = -4.0 + 1.0 / ls + gl * ls
        d_us = 1.0 - (sig / s0) ** 2 + sq - n
        return -logp, -d_f, -d_uv, -d_ul, -d_us

    def gp_run_mcmc(self, rng_key):
        """Posterior samples of gp_regression at the observed points (inference.py:463-490):
        dict with kernel_var, kernel_sigma, kernel_length [S] and f [S, n_obs]."""
        n = int(self.obs_idx.numel())
        # init_to_median of the priors: var 1, ls median of InverseGamma(4,1), sigma median of HalfNormal(.1)
        q0 = torch.zeros(n + 3, dtype=self.dtype, device=self.x.device)
        q0[n + 1] = math.log(0.2723)
        q0[n + 2] = math.log(0.0674)

        def potential(q):
            U, dF, dV, dL, dS = self.potential_and_grad(q[:, :n], q[:, n], q[:, n + 1], q[:, n + 2])
            return U, torch.cat([dF, dV[:, None], dL[:, None], dS[:, None]], 1)

        smp, _ = self._mcmc(rng_key, potential, q0)
        return {"f": smp[:, :n], "kernel_var": torch.exp(smp[:, n]), "kernel_length": torch.exp(smp[:, n + 1]),
                "kernel_sigma": torch.exp(smp[:, n + 2])}

    def gp_fit(self, plot=False, rng_key=None):
        """MCMC over the GP hyper-parameters, then kriging per posterior draw (inference.py:493-549).
        Returns (means, predictions), each [S, n_test].  Plotting is out of scope."""
        key = _seed_of(self.seed if rng_key is None else rng_key)
        post = self.gp_run_mcmc(key + 2)
        cpu = {k: v.detach().cpu().numpy() for k, v in post.items() if k != "f"}
        return self.gp_fit_from_samples(key + 3, cpu)

    def gp_fit_from_samples(self, rng_key, post_samples):
        """The posterior-predictive sweep of gp_fit (inference.py:511-521): one gp_prediction per
        posterior draw of (kernel_var, kernel_length, kernel_sigma).  With torch.distributed
        initialised the draws are sharded round-robin over ranks; no data-path collective — each
        rank returns its own rows (the host concatenates if it wants all of them)."""
        var = np.asarray(post_samples["kernel_var"], dtype=np.float64)
        ls = np.asarray(post_samples["kernel_length"], dtype=np.float64)
        sig = np.asarray(post_samples["kernel_sigma"], dtype=np.float64)
        mine = np.asarray(list(shard(var.shape[0])), dtype=np.int64)
        base = _seed_of(rng_key)
        x_obs, x_test = self._split()
        n_obs, n_test = x_obs.shape[0], x_test.shape[0]
        y_obs = self.y[self.obs_idx]
        # batches sized to ~4 GiB of workspace: one batched Gram + factorisation + finish each
        per = max(1, min(len(mine), 16384, int(2 ** 32 // max(1, (n_obs + n_test + 2) * n_obs * 8))))
        means, preds = [], []
        self.last_info = []
        for lo in range(0, len(mine), per):
            idx = mine[lo:lo + per]
            # draw s keeps the normals it had one problem at a time: seed base + s, stream 0xC41
            xi = torch.stack([ops.randn((n_test,), self.dtype, self.x.device, seed=base + int(s),
                                        stream_id=0xC41) for s in idx]) if n_test else None
            m, p, _, info = ops.krig_predict_batched(x_obs, x_test, y_obs, var[idx], ls[idx], sig[idx],
                                                     1.0e-5, xi)
            means.append(m)
            preds.append(p)
            self.last_info.append(info)
        if not means:
            e = torch.empty((0, n_test), dtype=self.dtype, device=self.x.device)
            return e, e.clone()
        return torch.cat(means), torch.cat(preds)


class VAEInference(Inference):
    """MCMC with the trained decoder as prior (inference.py:208-254)."""

    def __init__(self, decoder, decoder_params, z_dim, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.decoder = decoder
        self.decoder_params = decoder_params
        self.z_dim = z_dim
        (V1, _), _, (V2, _), _, (V3, _) = decoder_params
        self.h1, self.h0, self.n_out = V1.shape[1], V2.shape[1], V3.shape[1]
        self.dec_flat = decoder_flat_from_list(decoder_params, self.dtype)

    def run_mcmc(self, rng_key):
        """Posterior samples of VAEInference.regression (inference.py:69-99, 243-254): dict with
        noise [S], z [S, z_dim] and the deterministic f [S, n] = decoder(z)."""
        zd = self.z_dim
        q0 = torch.zeros(zd + 1, dtype=self.dtype, device=self.x.device)
        q0[zd] = math.log(0.0674)  # median of HalfNormal(0.1)

        def potential(q):
            U, dz, du = self.potential_and_grad(q[:, :zd], q[:, zd])
            return U, torch.cat([dz, du[:, None]], 1)

        smp, _ = self._mcmc(rng_key, potential, q0)
        z = smp[:, :zd].contiguous()
        return {"z": z, "noise": torch.exp(smp[:, zd]), "f": self.predict(z)}

    def fit(self, plot=False, rng_key=None, num_prior=1000):
        """Prior and posterior predictive draws y_pred ~ Normal(decoder(z), noise)
        (inference.py:101-146).  Returns (prior_predictions [num_prior, n], predictions [S, n])."""
        key = _seed_of(self.seed if rng_key is None else rng_key)
        dev, dt = self.x.device, self.dtype
        zp = ops.randn((num_prior, self.z_dim), dt, dev, seed=key, stream_id=0xF17)
        sp = 0.1 * ops.randn((num_prior,), dt, dev, seed=key, stream_id=0xF18).abs()  # HalfNormal(0.1)
        fp = self.predict(zp)
        prior = fp + sp[:, None] * ops.randn(tuple(fp.shape), dt, dev, seed=key, stream_id=0xF19)
        post = self.run_mcmc(key + 2)
        f = post["f"]
        pred = f + post["noise"][:, None] * ops.randn(tuple(f.shape), dt, dev, seed=key + 3, stream_id=0xF1A)
        return prior, pred

    def predict(self, z):
        """f = decoder(decoder_params, z) for a batch of latent vectors [chains, z_dim]."""
        return ops.decoder_fwd(self.dec_flat, _as_device(z, self.dtype).reshape(-1, self.z_dim),
                               self.n_out, self.h0, self.h1)

    def potential_and_grad(self, z, log_sigma):
        """Potential energy of VAEInference.regression (inference.py:243-254) in NumPyro's
        unconstrained space, and its gradient, for a batch of chains.

        sites: noise ~ HalfNormal(0.1) with sigma = exp(log_sigma); z ~ N(0, I);
               y[obs] ~ Normal(decoder(z)[obs], sigma).
        Returns (U [chains], dU/dz [chains, z_dim], dU/dlog_sigma [chains]).  The decoder forward and
        its vector-Jacobian product are the HIP kernels; the O(chains * n_obs) tail is glue."""
        z = _as_device(z, self.dtype).reshape(-1, self.z_dim).contiguous()
        u = _as_device(log_sigma, self.dtype).reshape(-1)
        sigma = torch.exp(u)
        f = ops.decoder_fwd(self.dec_flat, z, self.n_out, self.h0, self.h1)
        y_obs = self.y[self.obs_idx]
        r = f[:, self.obs_idx] - y_obs[None, :]
        n_obs = self.obs_idx.numel()
        s0 = 0.1
        half_log_2pi = 0.5 * math.log(2 * math.pi)
        sq = (r * r).sum(1)
        log_lik = -0.5 * sq / sigma ** 2 - n_obs * (u + half_log_2pi)
        log_pz = (-0.5 * z * z).sum(1) - self.z_dim * half_log_2pi
        log_ps = math.log(2.0) - half_log_2pi - math.log(s0) - 0.5 * (sigma / s0) ** 2 + u
        U = -(log_lik + log_pz + log_ps)
        dF = torch.zeros_like(f)
        dF[:, self.obs_idx] = r / (sigma ** 2)[:, None]
        dz = z + ops.decoder_vjp(self.dec_flat, z, dF, self.h0, self.h1)
        du = (sigma / s0) ** 2 - 1.0 + n_obs - sq / sigma ** 2
        return U, dz, du
