"""Inference drivers — mirror /root/reference/src/model/inference.py on top of libgpvae_hip.

What is built natively here is the arithmetic NUTS and the kriging post-processing call on every
evaluation:
  * GPInference.gp_prediction            -> gpvae_krig_predict (one problem, any size)
  * GPInference.gp_fit_from_samples      -> the vmap over posterior samples (inference.py:511-521),
                                            sharded over ranks with no collective
  * VAEInference.potential_and_grad      -> gpvae_decoder_fwd / gpvae_decoder_vjp batched over chains
                                            (what NUTS asks of VAEInference.regression per leapfrog)
The NUTS transition itself is NumPyro's (third party, not in the reference tree): `run_mcmc` drives the
batched restatement in gp_vae_amd/nuts.py (or the plain HMC of hmc.py) over the same potentials.  Plotting (inference.py:149-204) is out of scope.
"""
import math

import ctypes as C
import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .hmc import hmc
from .nuts import nuts, nuts_device, nuts_device_run, uniforms_per_iteration
from ._lib import GpvaeError
from .gp import _as_device, _key_parts, _seed_of, exp_kernel2
from .parallel import dist_info, shard
from .vae import decoder_flat_from_list


def _index(idx, device, n=None, what="obs_idx"):
    """Index array (list / numpy / tensor on any device) as a long tensor on `device`; with n, every entry must
    lie in [0, n) (checked once, here, where the index set enters — the kernels that gather and scatter through
    it do not re-check per call).  Repeats are legal, as in the reference's f[obs_idx]."""
    if isinstance(idx, torch.Tensor):
        t = idx.to(device=device, dtype=torch.long)
    else:
        t = torch.as_tensor(np.asarray(idx), dtype=torch.long, device=device)
    t = t.reshape(-1).contiguous()
    if n is not None:
        ops.check_index(t, n, what)
    return t


class Potential():
    """What NUTS/HMC needs of a NumPyro model conditioned on data: the potential energy in the
    unconstrained space and its gradient for a batch of chains (`__call__(q [C, dim]) -> (U [C],
    dU/dq [C, dim])`), a start point (`init`, the init_to_median of inference.py:84,477) and the map
    from unconstrained rows back to the model's sites (`unpack`).  Returned by the `regression` /
    `gp_regression` methods when they are called with data, i.e. where the reference hands the model
    function to NUTS."""

    def __init__(self, fn, init, unpack, eval_into=None):
        self.fn, self.init, self.unpack = fn, init, unpack
        # optional fast path for samplers that call the potential thousands of times per second:
        # eval_into(q, U_out, g_out, stream_handle) writes into the caller's buffers (q, g_out with unit column
        # stride), no allocation and no stream look-up per call
        self.eval_into = eval_into

    def __call__(self, q):
        return self.fn(q)


class Inference():
    """Constructor and attributes of the reference base class (inference.py:35-59)."""

    def __init__(self, d, x, y, obs_idx, mcmc_args, seed=0, ground_truth=None, grid=True,
                 dtype=torch.float32):
        self.d = d
        self.x = _as_device(x, dtype)
        self.y = _as_device(y, dtype)
        self.obs_idx = _index(obs_idx, self.x.device, self.x.shape[0])
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

    def _mcmc(self, rng_key, potential, q0=None):
        """mcmc_args of the reference (num_warmup, num_samples, num_chains, thinning;
        inference.py:86-93) driving the batched NUTS of gp_vae_amd.nuts (numpyro.infer.NUTS is third-party
        code that is absent here; the module restates the published algorithm).  Extra keys:
        "kernel": "nuts" (default) | "hmc" (gp_vae_amd.hmc, fixed-length trajectories), "max_tree_depth"
        (10, NumPyro's default), "num_leapfrog" (hmc).  q0: [dim] start, jittered per chain.  With
        torch.distributed initialised the chains are sharded round-robin over the ranks (SURVEY.md §8e:
        no collective; chain c keeps its start point and its own random stream whatever the world size)
        and each rank returns its own chains' samples [num_samples // thinning * local_chains, dim],
        chains concatenated as mcmc.get_samples() does; the host concatenates ranks if it wants all."""
        a = self.mcmc_args
        chains = int(a.get("num_chains", 1))
        kernel = a.get("kernel", "nuts")
        if kernel not in ("nuts", "hmc"):
            raise ValueError("mcmc_args['kernel'] must be 'nuts' or 'hmc'")
        if q0 is None:
            q0 = potential.init
        mine = list(shard(chains))
        self.chain_ids = mine
        seed = _seed_of(rng_key)
        if not mine:
            self.mcmc_info = {}
            return q0.new_empty((0, q0.numel())), {}
        rows = torch.as_tensor(mine, dtype=torch.long, device=q0.device)
        start = q0[None, :] + 0.01 * ops.randn((chains, q0.numel()), q0.dtype, q0.device, seed=seed,
                                               stream_id=0x1417 << 32)[rows]
        D = q0.numel()
        depth = int(a.get("max_tree_depth", 10))
        nu = uniforms_per_iteration(depth) if kernel == "nuts" else 1

        def noise_fn(it):
            # Philox rows of ALL chains, this rank keeps its own: element (chain c, j) of iteration `it`
            # is the same number on any world size.  Uniforms = Phi(normal).
            g = ops.randn((chains, D + nu), q0.dtype, q0.device, seed=seed + 1, stream_id=(0x1418 << 32) | it)[rows]
            unif = 0.5 * (1.0 + torch.erf(g[:, D:] * (1.0 / math.sqrt(2.0))))
            return g[:, :D], (unif if kernel == "nuts" else unif[:, 0])

        if kernel == "nuts":
            # the transition runs in libgpvae_hip (gpvae_nuts_op); "nuts_driver": "torch" keeps the torch restatement
            # "nuts_driver": "device" (default: the whole run as ONE launch where the potential offers it —
            # gpvae_nuts_vae_run — else as "iteration"), "iteration" (one fused launch per iteration where offered:
            # gpvae_nuts_vae_iteration, adaptation as torch operations), "lockstep" (gpvae_nuts_op around the
            # potential, all chains in lock step), "torch" (the torch restatement).  Same noise in all four.
            driver = a.get("nuts_driver", "device")
            if driver not in ("device", "iteration", "lockstep", "torch"):
                raise ValueError("mcmc_args['nuts_driver'] must be 'device', 'iteration', 'lockstep' or 'torch'")
            if driver == "device" and hasattr(potential, "nuts_run"):
                smp, info = nuts_device_run(potential, start, int(a["num_warmup"]), int(a["num_samples"]),
                                            thinning=int(a.get("thinning", 1)), max_tree_depth=depth,
                                            noise_seed=seed + 1, noise_stream_base=0x1418 << 32, chain_ids=rows)
            else:
                kw = {} if driver == "torch" else {"fused": driver in ("device", "iteration")}
                smp, info = (nuts if driver == "torch" else nuts_device)(
                    potential, start, int(a["num_warmup"]), int(a["num_samples"]), thinning=int(a.get("thinning", 1)),
                    max_tree_depth=depth, seed=seed + 1, noise_fn=noise_fn, **kw)
        else:
            smp, info = hmc(potential, start, int(a["num_warmup"]), int(a["num_samples"]),
                            thinning=int(a.get("thinning", 1)), num_leapfrog=int(a.get("num_leapfrog", 16)),
                            seed=seed + 1, noise_fn=noise_fn)
        self.mcmc_info = info
        return smp.permute(1, 0, 2).reshape(-1, q0.numel()), info

    def run_mcmc(self, rng_key):
        """Posterior samples of self.regression given (self.y, self.obs_idx) (inference.py:69-99):
        the dict mcmc.get_samples() returns, one entry per sample site."""
        pot = self.regression(y=self.y, obs_idx=self.obs_idx)
        smp, _ = self._mcmc(rng_key, pot)
        return pot.unpack(smp)


class GPInference(Inference):
    """Kriging with a GP prior (inference.py:338-549)."""

    def __init__(self, kernel, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if kernel is not exp_kernel2:
            raise NotImplementedError("GPInference is built for exp_kernel2")
        self.kernel = kernel

    def _split(self, x=None):
        x = self.x if x is None else x
        n = x.shape[0]
        mask = torch.ones(n, dtype=torch.bool, device=x.device)
        mask[self.obs_idx] = False
        test_idx = torch.nonzero(mask).reshape(-1)  # jnp.delete(x, obs_idx, axis=0)  inference.py:442
        return x[self.obs_idx].contiguous(), x[test_idx].contiguous()

    def gp_prediction(self, rng_key, x, y, var, ls, sigma, return_var=False):
        """Posterior mean and one posterior sample at the unobserved locations (inference.py:427-460).
        `rng_key` seeds the standard normals of the sample (Philox stream)."""
        x = self.x if x is self.x else _as_device(x, self.dtype)  # a local: self.x stays what it was
        y = _as_device(y, self.dtype)
        x_obs, x_test = self._split(x)
        xi = ops.randn((x_test.shape[0],), self.dtype, x.device, seed=_seed_of(rng_key),
                       stream_id=0xC41)
        mean, sample, var_out, info = ops.krig_predict(x_obs, x_test, y[self.obs_idx], var, ls,
                                                       sigma, 1.0e-5, xi)
        self.last_info = info
        return (mean, sample, var_out) if return_var else (mean, sample)

    def _gp_potential(self, x_pts, y_lik, lik_idx, f, log_var, log_ls, log_sigma):
        """Potential of the latent-f GP model on the points x_pts with a Gaussian likelihood on
        f[lik_idx] (None: every point), in NumPyro's unconstrained space, and its gradient.
        sites: kernel_var ~ LogNormal(0,0.1), kernel_sigma ~ HalfNormal(0.1),
               kernel_length ~ InverseGamma(4,1), f ~ MVN(0, kernel(x, x, var, ls)),
               y ~ Normal(f[lik_idx], sigma) observed.       (inference.py:368-396, 407-424)
        The MVN term and its gradient are one gpvae_mvn_logp call; the priors and the Gaussian
        likelihood are O(chains * n) glue."""
        n = x_pts.shape[0]
        f = _as_device(f, self.dtype).reshape(-1, n).contiguous()
        uv, ul, us = (_as_device(t, self.dtype).reshape(-1) for t in (log_var, log_ls, log_sigma))
        var, ls, sig = torch.exp(uv), torch.exp(ul), torch.exp(us)
        lp, gf, gv, gl, info = ops.mvn_logp(x_pts, f, var, ls, 1.0e-5)
        self.last_info = info
        s0, hl2p = 0.1, 0.5 * math.log(2 * math.pi)
        f_lik = f if lik_idx is None else f[:, lik_idx]
        m = f_lik.shape[1]
        r = (y_lik[None, :] - f_lik) / sig[:, None]
        sq = (r * r).sum(1)
        logp = (-math.log(s0) - hl2p - uv ** 2 / (2 * s0 ** 2)) \
            + (math.log(2.0) - hl2p - math.log(s0) - 0.5 * (sig / s0) ** 2 + us) \
            + (-math.lgamma(4.0) - 4.0 * ul - 1.0 / ls) \
            + lp + (-0.5 * sq - m * us - m * hl2p)
        if lik_idx is None:
            d_f = gf + r / sig[:, None]
        else:
            d_f = gf.clone()
            d_f.index_add_(1, lik_idx, r / sig[:, None])  # repeats in lik_idx accumulate
        d_uv = -uv / s0 ** 2 + gv * var
        d_ul = -4.0 + 1.0 / ls + gl * ls
        d_us = 1.0 - (sig / s0) ** 2 + sq - m
        return -logp, -d_f, -d_uv, -d_ul, -d_us

    def potential_and_grad(self, f, log_var, log_ls, log_sigma):
        """Potential energy of gp_regression (inference.py:400-424) at the observed points and its
        gradient, for a batch of chains — what NUTS evaluates per leapfrog step.
        f: [chains, n_obs]; log_var, log_ls, log_sigma: [chains].
        Returns (U [chains], dU/df [chains, n_obs], dU/dlog_var, dU/dlog_ls, dU/dlog_sigma)."""
        x_obs = self.x[self.obs_idx].contiguous()
        n = x_obs.shape[0]
        q = torch.cat([_as_device(f, self.dtype).reshape(-1, n)] +
                      [_as_device(t, self.dtype).reshape(-1, 1) for t in (log_var, log_ls, log_sigma)], 1)
        U, dU, info = ops.gp_potential(x_obs, q, self.y[self.obs_idx].contiguous(), None, 1.0e-5)
        self.last_info = info
        return U, dU[:, :n], dU[:, n], dU[:, n + 1], dU[:, n + 2]

    def _latent_gp_model(self, x_pts, y_lik, lik_idx):
        """The Potential of the latent-f model: unconstrained vector q = (f [n], log var, log ls,
        log sigma)."""
        n = x_pts.shape[0]
        # init_to_median of the priors: var 1, ls median of InverseGamma(4,1), sigma median of HalfNormal(.1)
        q0 = torch.zeros(n + 3, dtype=self.dtype, device=self.x.device)
        q0[n + 1] = math.log(0.2723)
        q0[n + 2] = math.log(0.0674)

        y_lik = y_lik.contiguous()

        def fn(q):  # one library call per evaluation (gpvae_gp_potential)
            U, dU, info = ops.gp_potential(x_pts, q if q.stride(1) == 1 else q.contiguous(), y_lik, lik_idx, 1.0e-5,
                                           checked=True)
            self.last_info = info
            return U, dU

        def unpack(smp):
            return {"f": smp[:, :n], "kernel_var": torch.exp(smp[:, n]),
                    "kernel_length": torch.exp(smp[:, n + 1]), "kernel_sigma": torch.exp(smp[:, n + 2])}

        return Potential(fn, q0, unpack)

    def _predictive_draws(self, x_pts, rng_key, num_samples, posterior_samples):
        """y_pred ~ Normal(f, sigma) with f ~ MVN(0, kernel(x, x, var, ls)) per draw (the y=None branch,
        inference.py:386-389), hyper-parameters from the priors or from posterior samples."""
        seed, base = _key_parts(rng_key)
        dev, dt = self.x.device, self.dtype
        if posterior_samples is not None:
            var = torch.as_tensor(posterior_samples["kernel_var"], dtype=dt, device=dev).reshape(-1)
            ls = torch.as_tensor(posterior_samples["kernel_length"], dtype=dt, device=dev).reshape(-1)
            sig = torch.as_tensor(posterior_samples["kernel_sigma"], dtype=dt, device=dev).reshape(-1)
        else:
            rng = np.random.default_rng(seed if base == 0 else [seed, base])
            S = int(num_samples)
            var = torch.as_tensor(np.exp(0.1 * rng.standard_normal(S)), dtype=dt, device=dev)   # LogNormal(0,.1)
            sig = torch.as_tensor(np.abs(0.1 * rng.standard_normal(S)), dtype=dt, device=dev)   # HalfNormal(.1)
            ls = torch.as_tensor(1.0 / rng.gamma(4.0, size=S), dtype=dt, device=dev)            # InverseGamma(4,1)
        S, n = var.numel(), x_pts.shape[0]
        E = ops.randn((S, n), dt, dev, seed=seed, stream_id=base | 1)
        E2 = ops.randn((S, n), dt, dev, seed=seed, stream_id=base | 2)
        F = torch.empty((S, n), dtype=dt, device=dev)
        Y = torch.empty((S, n), dtype=dt, device=dev)
        per = max(1, min(S, int(2 ** 31 // max(1, n * n * x_pts.element_size()))))
        for lo in range(0, S, per):
            hi = min(S, lo + per)
            Kb = ops.gram_se(x_pts, x_pts, var[lo:hi], ls[lo:hi], 1.0e-5).reshape(hi - lo, n, n)
            ops.potrf_(Kb)
            F[lo:hi], Y[lo:hi] = ops.draw_batched(Kb, E[lo:hi], E2[lo:hi], sig[lo:hi], want_f=True)
        return {"kernel_var": var, "kernel_sigma": sig, "kernel_length": ls, "f": F, "y_pred": Y}

    def regression(self, y=None, obs_idx=None, rng_key=None, num_samples=None, posterior_samples=None):
        """GP regression model on the whole grid (inference.py:359-397).
        With data (`y`, `obs_idx`; NaN allowed at unobserved entries of y): the :class:`Potential` NUTS
        would be given.  Without (`y=None`, the prediction branch): a dict of draws with 'y_pred' —
        what Predictive(self.regression, num_samples | posterior_samples)(rng_key) returns."""
        if y is None:
            return self._predictive_draws(self.x, rng_key, num_samples or 1, posterior_samples)
        idx = self.obs_idx if obs_idx is None else _index(obs_idx, self.x.device, self.x.shape[0])
        yy = _as_device(y, self.dtype)
        return self._latent_gp_model(self.x, yy[idx], idx)

    def gp_regression(self, x, y=None, rng_key=None, num_samples=None, posterior_samples=None):
        """Kriging / exact GP regression model at the observed locations x (inference.py:400-424):
        with y the :class:`Potential` of (f, kernel_var, kernel_length, kernel_sigma); with y=None
        prior / posterior predictive draws at x."""
        xx = x if (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == self.dtype) else _as_device(x, self.dtype)
        xx = ops._points(xx)
        if y is None:
            return self._predictive_draws(xx, rng_key, num_samples or 1, posterior_samples)
        return self._latent_gp_model(xx, _as_device(y, self.dtype).reshape(-1), None)

    def gp_run_mcmc(self, rng_key, x=None, y=None, model=None):
        """Posterior samples of `model` (default self.gp_regression) given the observed (x, y)
        (inference.py:463-490): dict with kernel_var, kernel_sigma, kernel_length [S] and f [S, n_obs]."""
        x = self.x[self.obs_idx] if x is None else x
        y = self.y[self.obs_idx] if y is None else y
        pot = (self.gp_regression if model is None else model)(x, y)
        if not isinstance(pot, Potential):
            raise NotImplementedError("gp_run_mcmc: `model` must be GPInference.gp_regression (or return a "
                                      "Potential); arbitrary NumPyro models need NumPyro")
        smp, _ = self._mcmc(rng_key, pot)
        return pot.unpack(smp)

    def gp_fit(self, plot=False, rng_key=None):
        """MCMC over the GP hyper-parameters, then kriging per posterior draw (inference.py:493-549).
        Returns (means, predictions), each [S, n_test].  Plotting is out of scope."""
        key = _seed_of(self.seed if rng_key is None else rng_key)
        post = self.gp_run_mcmc(key + 2, self.x[self.obs_idx], self.y[self.obs_idx], model=self.gp_regression)
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
        ops.check_index(self.obs_idx, self.n_out, "obs_idx (decoder outputs)")

    def regression(self, y=None, obs_idx=None, rng_key=None, num_samples=None, posterior_samples=None):
        """Regression model with the decoder as prior (inference.py:235-254).
        With data: the :class:`Potential` over q = (z [z_dim], log noise) NUTS would be given, whose
        `unpack` adds the deterministic site f = decoder(z).  With y=None (prediction branch,
        inference.py:248-249): dict of draws with 'y_pred' ~ Normal(decoder(z), noise), z and noise
        from the priors (num_samples) or from posterior_samples."""
        zd = self.z_dim
        dev, dt = self.x.device, self.dtype
        if y is None:
            seed, base = _key_parts(rng_key)
            if posterior_samples is not None:
                z = _as_device(posterior_samples["z"], dt).reshape(-1, zd)
                noise = _as_device(posterior_samples["noise"], dt).reshape(-1)
            else:
                S = int(num_samples or 1)
                z = ops.randn((S, zd), dt, dev, seed=seed, stream_id=base | 5)
                noise = 0.1 * ops.randn((S,), dt, dev, seed=seed, stream_id=base | 6).abs()  # HalfNormal(0.1)
            f = self.predict(z)
            y_pred = f + noise[:, None] * ops.randn(tuple(f.shape), dt, dev, seed=seed, stream_id=base | 7)
            return {"noise": noise, "z": z, "f": f, "y_pred": y_pred}
        idx = self.obs_idx if obs_idx is None else _index(obs_idx, dev, self.n_out)
        y_obs = _as_device(y, dt)[idx]
        q0 = torch.zeros(zd + 1, dtype=dt, device=dev)
        q0[zd] = math.log(0.0674)  # median of HalfNormal(0.1)

        y_obs = y_obs.contiguous()
        idx = idx.contiguous()
        fused = ops.vae_potential_supported(dt, self.h0, self.h1, zd, idx.numel())

        def fn(q):
            if fused:  # one launch per evaluation (gpvae_vae_potential)
                return ops.vae_potential(self.dec_flat, q if q.stride(1) == 1 else q.contiguous(), y_obs, idx,
                                         self.n_out, self.h0, self.h1, checked=True)
            U, dz, du = self._potential(q[:, :zd], q[:, zd], y_obs, idx)
            return U, torch.cat([dz, du[:, None]], 1)

        def unpack(smp):
            z = smp[:, :zd].contiguous()
            return {"z": z, "noise": torch.exp(smp[:, zd]), "f": self.predict(z)}

        eval_into = None
        if fused:
            from . import _lib
            lib, dtc, vp = _lib.load(), ops._dt(y_obs), C.c_void_p
            p_dec, p_y, p_idx, n_obs = vp(self.dec_flat.data_ptr()), vp(y_obs.data_ptr()), vp(idx.data_ptr()), idx.numel()
            n_out, h0, h1 = self.n_out, self.h0, self.h1

            def eval_into(q, U_out, g_out, stream_handle):
                rc = lib.gpvae_vae_potential(stream_handle, dtc, q.shape[0], n_out, h0, h1, zd, p_dec, vp(q.data_ptr()),
                                             q.stride(0), p_y, p_idx, n_obs, 0.1, vp(U_out.data_ptr()),
                                             vp(g_out.data_ptr()), g_out.stride(0))
                if rc:
                    _lib.check(rc, "gpvae_vae_potential")

        pot = Potential(fn, q0, unpack, eval_into)
        if fused and lib.gpvae_nuts_vae_supported(dtc, self.h0, self.h1, zd, n_obs):
            def nuts_iteration(stream_handle, dtc_, chains, max_depth, buffers):
                rc = lib.gpvae_nuts_vae_iteration(stream_handle, dtc_, chains, max_depth, buffers, n_out, h0, h1, zd,
                                                  p_dec, p_y, p_idx, n_obs, 0.1)
                if rc:
                    _lib.check(rc, "gpvae_nuts_vae_iteration")
            pot.nuts_iteration = nuts_iteration  # one launch per NUTS iteration (gp_vae_amd.nuts.nuts_device)

            def nuts_run(stream_handle, dtc_, chains, run_desc):
                rc = lib.gpvae_nuts_vae_run(stream_handle, dtc_, chains, run_desc, n_out, h0, h1, zd, p_dec, p_y, p_idx,
                                            n_obs, 0.1)
                if rc:
                    _lib.check(rc, "gpvae_nuts_vae_run")
            pot.nuts_run = nuts_run  # the whole run in one launch (gp_vae_amd.nuts.nuts_device_run)
        return pot

    def fit(self, plot=False, rng_key=None, num_prior=1000):
        """Prior and posterior predictive draws y_pred ~ Normal(decoder(z), noise)
        (inference.py:101-146).  Returns (prior_predictions [num_prior, n], predictions [S, n])."""
        key = _seed_of(self.seed if rng_key is None else rng_key)
        prior = self.regression(rng_key=key, num_samples=num_prior)["y_pred"]          # inference.py:113-114
        post = self.run_mcmc(key + 2)                                                   # inference.py:116
        pred = self.regression(rng_key=key + 3, posterior_samples=post)["y_pred"]       # inference.py:119-120
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
        zz = _as_device(z, self.dtype).reshape(-1, self.z_dim)
        if ops.vae_potential_supported(self.dtype, self.h0, self.h1, self.z_dim, self.obs_idx.numel()):
            q = torch.cat([zz, _as_device(log_sigma, self.dtype).reshape(-1, 1)], 1)
            U, dU = ops.vae_potential(self.dec_flat, q, self.y[self.obs_idx].contiguous(), self.obs_idx.contiguous(),
                                      self.n_out, self.h0, self.h1, checked=True)  # obs_idx was checked in __init__
            return U, dU[:, :self.z_dim], dU[:, self.z_dim]
        return self._potential(z, log_sigma, self.y[self.obs_idx], self.obs_idx)

    def _potential(self, z, log_sigma, y_obs, obs_idx):
        z = _as_device(z, self.dtype).reshape(-1, self.z_dim).contiguous()
        u = _as_device(log_sigma, self.dtype).reshape(-1)
        sigma = torch.exp(u)
        f = ops.decoder_fwd(self.dec_flat, z, self.n_out, self.h0, self.h1)
        r = f[:, obs_idx] - y_obs[None, :]
        n_obs = obs_idx.numel()
        s0 = 0.1
        half_log_2pi = 0.5 * math.log(2 * math.pi)
        sq = (r * r).sum(1)
        log_lik = -0.5 * sq / sigma ** 2 - n_obs * (u + half_log_2pi)
        log_pz = (-0.5 * z * z).sum(1) - self.z_dim * half_log_2pi
        log_ps = math.log(2.0) - half_log_2pi - math.log(s0) - 0.5 * (sigma / s0) ** 2 + u
        U = -(log_lik + log_pz + log_ps)
        dF = torch.zeros_like(f)
        dF.index_add_(1, obs_idx, r / (sigma ** 2)[:, None])  # repeats in obs_idx accumulate
        dz = z + ops.decoder_vjp(self.dec_flat, z, dF, self.h0, self.h1)
        du = (sigma / s0) ** 2 - 1.0 + n_obs - sq / sigma ** 2
        return U, dz, du
