"""GP prior — mirrors /root/reference/src/model/gp.py on top of libgpvae_hip.

Same names, argument order and defaults as the reference (`exp_kernel2`, `agg_kernel_grid`, `GP`),
with torch device tensors where the reference has jax arrays.  NumPyro is not a dependency: where
the reference relies on `Predictive(gp.sample, num_samples=B)` to vectorise a NumPyro model,
:class:`Predictive` below gives the same call shape and return dict.
"""
import numpy as np
import torch

from . import ops
from ._lib import GpvaeError


def _as_device(x, dtype=None):
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.as_tensor(np.asarray(x))
    if dtype is not None:
        t = t.to(dtype)
    elif not t.is_floating_point():
        t = t.to(torch.float32)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise GpvaeError("gp_vae_amd needs a HIP device (there is no CPU path)")
        t = t.cuda()
    return t


def exp_kernel2(x, z, var, ls, jitter=1.0e-5):
    """Squared-exponential Gram matrix (gp.py:44-81): var*exp(-0.5*(|x-z|/ls)^2), +jitter*I iff square."""
    x = _as_device(x)
    z = _as_device(z, x.dtype)
    return ops.gram_se(x, z, var, ls, jitter)


def _seed_of(rng_key):
    """An int, a numpy Generator/SeedSequence-like or a 2-word key all reduce to one 64-bit seed."""
    if rng_key is None:
        return 0
    if isinstance(rng_key, (int, np.integer)):
        return int(rng_key) & (2 ** 63 - 1)
    a = np.asarray(rng_key).astype(np.uint64).reshape(-1)
    s = 0
    for w in a:
        s = (s * 0x9E3779B97F4A7C15 + int(w)) & (2 ** 63 - 1)
    return s


class PhiloxKey():
    """A counter-based key: (seed, stream base).  A consumer derives its Philox stream ids as
    base | purpose (purpose 1: eps, 2: eps', 3: eps_z; parallel.philox_stream_id packs rank, epoch key
    and step into the base), so the literal Predictive -> SVI.update loop and the pipelined
    StepEngine draw the same normals for the same step."""

    def __init__(self, seed, base=0):
        self.seed = int(seed) & (2 ** 63 - 1)
        self.base = int(base) & ~0xF & (2 ** 64 - 1)

    def stream(self, purpose):
        return self.base | (purpose & 0xF)


def _key_parts(rng_key):
    """(seed, stream base) of any accepted key."""
    if isinstance(rng_key, PhiloxKey):
        return rng_key.seed, rng_key.base
    return _seed_of(rng_key), 0


def agg_kernel_grid(rng_key, d, m, kernel, var, ls, grid1, grid2=None, jitter=1.0e-6):
    """Cell-averaged kernel over a [0,1]^d grid (gp.py:85-127).

    The Uniform(0,1) locations come from torch's generator seeded by `rng_key` (the reference's
    threefry stream is not reproducible without JAX); `rng_key` may also be a ready-made tensor of
    uniforms [n^d, m, d].  `jitter` keeps the literal behaviour: + jitter/m on every entry."""
    if kernel is not exp_kernel2:
        raise NotImplementedError("agg_kernel_grid is built for exp_kernel2 only")
    if d not in (1, 2):
        raise Warning("Function is only implemented for d=1,2")  # gp.py:117
    grid1 = _as_device(grid1)  # n = grid1.shape[0]: gp.py:99 with the `shape()` typo fixed
    if isinstance(rng_key, torch.Tensor) and rng_key.dim() == 3:
        pts = _agg_points(None, d, m, grid1, grid2, u=_as_device(rng_key, grid1.dtype))
    else:
        pts = _agg_points(rng_key, d, m, grid1, grid2)
    return ops.gram_agg(pts, var, ls, jitter)


def draw_hyper(rng, size=None):
    """(ls, var, sigma) ~ InverseGamma(1,0.1), LogNormal(0,0.1), HalfNormal(0.1)  (gp.py:171-176)."""
    ls = 0.1 / rng.gamma(1.0, size=size)
    var = np.exp(0.1 * rng.standard_normal(size=size))
    sigma = np.abs(0.1 * rng.standard_normal(size=size))
    return ls, var, sigma


class GP():
    """GP prior with the reference's constructor (gp.py:141-156)."""

    def __init__(self, kernel=exp_kernel2, jitter=1.0e-5, d=1):
        self.kernel = kernel
        self.jitter = jitter
        self.d = d

    def sample(self, x, y=None, ls=None, var=None, sigma=None, rng_key=None, num_samples=1):
        """Draw from the GP (gp.py:159-194): k = kernel(x,x,var,ls,jitter); f ~ MVN(0,k);
        y ~ Normal(f, sigma).  Hyper-parameters left None are drawn from their priors, one set PER
        SAMPLE (what Predictive does to the NumPyro model).  Returns {'f','y'[, 'length','var',
        'noise']} with a leading [num_samples] axis."""
        x = _as_device(x)
        n = x.shape[0]
        seed, base = _key_parts(rng_key)
        B = int(num_samples)
        E = ops.randn((B, n), x.dtype, x.device, seed=seed, stream_id=base | 1)
        E2 = ops.randn((B, n), x.dtype, x.device, seed=seed, stream_id=base | 2)
        out = {}
        if ls is not None and var is not None and sigma is not None:
            K = self.kernel(x, x, var, ls, self.jitter)
            self.last_info = ops.potrf_(K)
            F, Y = ops.draw(K, E, E2, sigma, want_f=True)
        else:
            rng = np.random.default_rng(seed if base == 0 else [seed, base])
            ls_s, var_s, sig_s = draw_hyper(rng, B)
            if ls is not None:
                ls_s = np.full(B, float(ls))
            else:
                out["length"] = torch.as_tensor(ls_s, dtype=x.dtype, device=x.device)
            if var is not None:
                var_s = np.full(B, float(var))
            else:
                out["var"] = torch.as_tensor(var_s, dtype=x.dtype, device=x.device)
            if sigma is not None:
                sig_s = np.full(B, float(sigma))
            else:
                out["noise"] = torch.as_tensor(sig_s, dtype=x.dtype, device=x.device)
            if self.kernel is not exp_kernel2:
                raise NotImplementedError("per-sample hyper-parameters need the built-in kernel")
            var_t = torch.as_tensor(var_s, dtype=x.dtype, device=x.device)
            ls_t = torch.as_tensor(ls_s, dtype=x.dtype, device=x.device)
            sig_t = torch.as_tensor(sig_s, dtype=x.dtype, device=x.device)
            F = torch.empty((B, n), dtype=x.dtype, device=x.device)
            Y = torch.empty((B, n), dtype=x.dtype, device=x.device)
            # B distinct K/L (gp.py:171-176 under Predictive): batched Gram, ONE batched factor sweep
            # and one batched-factor draw per chunk of samples; chunks bound the factors to ~2 GiB
            per = max(1, min(B, int(2 ** 31 // max(1, n * n * x.element_size()))))
            infos = []
            for lo in range(0, B, per):
                hi = min(B, lo + per)
                Kb = ops.gram_se(x, x, var_t[lo:hi], ls_t[lo:hi], self.jitter)
                Kb = Kb.reshape(hi - lo, n, n)
                infos.append(ops.potrf_(Kb))
                f_c, y_c = ops.draw_batched(Kb, E[lo:hi], E2[lo:hi], sig_t[lo:hi], want_f=True)
                F[lo:hi], Y[lo:hi] = f_c, y_c
            self.last_info = torch.cat(infos)
        out["f"] = F
        out["y"] = Y if y is None else _as_device(y, x.dtype).expand(B, n)
        return out


def _agg_points(rng_key, d, m, grid1, grid2=None, u=None):
    """The cells' Monte-Carlo sample points [n^d, m, d] of agg_kernel_grid (gp.py:98-113); `u`: ready-made
    uniforms [n^d, m, d] instead of torch's generator seeded by rng_key."""
    grid1 = _as_device(grid1)
    n = grid1.shape[0]
    if u is None:
        gen = torch.Generator(device=grid1.device)
        gen.manual_seed(_seed_of(rng_key))
        u = torch.rand((n ** d, m, d), generator=gen, device=grid1.device, dtype=grid1.dtype)
    if d == 1:
        return u / n + grid1[:, None, None]  # gp.py:107
    g2 = grid1 if grid2 is None else _as_device(grid2, grid1.dtype)
    uu, vv = torch.meshgrid(grid1, g2, indexing="xy")  # jnp.meshgrid default     gp.py:112
    return u / n + torch.stack([uu.flatten(), vv.flatten()], dim=1)[:, None, :]  # gp.py:113


class PoiGP(GP):
    """Log-Gaussian Cox process prior on a [0,1]^d grid (gp.py:197-250) with the reference's argument list:
    k = agg_kernel_grid(PRNGKey(seed), d, m, kernel, var, ls, x1, x2, jitter); f ~ MVN(0, k); rate = exp(f);
    y ~ Poisson(rate) — no observation noise (`sigma` is accepted and ignored, gp.py:228-229,248).

    The reference's method does not run as written (`grid1.shape()`, a loc of n instead of n^d
    entries in 2-D); this is the evident intent: n = x1.shape[0] cells per axis, f over all n^d cells.  Under
    Predictive the hyper-parameters left None are drawn per sample (sites `length`, `var`), the cells' sample
    points come from `seed` and are therefore the same for every sample, as in the reference."""

    def sample(self, m, x1, x2=None, y=None, ls=None, var=None, sigma=None, seed=0, rng_key=None, num_samples=1):
        if self.kernel is not exp_kernel2:
            raise NotImplementedError("PoiGP is built for exp_kernel2 only")
        if self.d not in (1, 2):
            raise Warning("Function is only implemented for d=1,2")  # gp.py:117
        x1 = _as_device(x1)
        pts = _agg_points(seed, self.d, int(m), x1, x2)
        cells, B = pts.shape[0], int(num_samples)
        kseed, base = _key_parts(rng_key)
        E = ops.randn((B, cells), x1.dtype, x1.device, seed=kseed, stream_id=base | 1)
        out = {}
        if ls is not None and var is not None:
            K = ops.gram_agg(pts, var, ls, self.jitter)
            self.last_info = ops.potrf_(K)
            F = ops.draw(K, E)
        else:
            rng = np.random.default_rng(kseed if base == 0 else [kseed, base])
            ls_s, var_s, _ = draw_hyper(rng, B)
            if ls is not None:
                ls_s = np.full(B, float(ls))
            else:
                out["length"] = torch.as_tensor(ls_s, dtype=x1.dtype, device=x1.device)
            if var is not None:
                var_s = np.full(B, float(var))
            else:
                out["var"] = torch.as_tensor(var_s, dtype=x1.dtype, device=x1.device)
            # B distinct aggregated kernels: one gpvae_gram_agg each, ONE batched factor sweep and one
            # batched-factor draw per chunk of samples (chunks bound the factors to ~2 GiB)
            F = torch.empty((B, cells), dtype=x1.dtype, device=x1.device)
            per = max(1, min(B, int(2 ** 31 // max(1, cells * cells * x1.element_size()))))
            infos = []
            for lo in range(0, B, per):
                hi = min(B, lo + per)
                Kb = torch.stack([ops.gram_agg(pts, float(var_s[i]), float(ls_s[i]), self.jitter) for i in range(lo, hi)])
                infos.append(ops.potrf_(Kb))
                F[lo:hi] = ops.draw_batched(Kb, E[lo:hi])
            self.last_info = torch.cat(infos)
        rate, counts = ops.poisson_exp(F, seed=kseed, stream_id=base | 4)
        out["f"], out["rate"] = F, rate
        out["y"] = counts if y is None else _as_device(y).expand(B, cells)
        return out


class Predictive():
    """Call-compatible stand-in for numpyro.infer.Predictive at the reference's call sites:
      Predictive(gp.sample, num_samples=B)(rng_key, x=x, ls=..., var=..., sigma=...) -> {'y': [B,n], 'f', ...}
                                                          (vae.py:223-226,160-162; toy_examples.py:88)
      Predictive(inference.regression, num_samples=1000)(rng_key)["y_pred"]      (inference.py:113-114,502-503)
      Predictive(inference.regression, post_samples)(rng_key)["y_pred"]          (inference.py:119-120)
    The model is a method of this package that vectorises itself over draws: it is called once with
    rng_key= and num_samples= (or posterior_samples=) added to the caller's arguments."""

    def __init__(self, model, posterior_samples=None, num_samples=None):
        if posterior_samples is None and num_samples is None:
            raise ValueError("Predictive: give num_samples or posterior_samples")
        self.model = model
        self.posterior_samples = posterior_samples
        self.num_samples = num_samples

    def __call__(self, rng_key, *args, **kwargs):
        if self.posterior_samples is not None:
            return self.model(*args, rng_key=rng_key, posterior_samples=self.posterior_samples, **kwargs)
        return self.model(*args, rng_key=rng_key, num_samples=self.num_samples, **kwargs)
