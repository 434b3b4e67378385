"""JAX side of the FFI adapter: registers the handlers of libgpvae_ffi.so (gpvae_ffi.cc) and wraps them
with the reference's signatures, so that src/model/{gp,vae,inference}.py keep driving the path from
NumPyro.

NOT RUN IN THIS REPOSITORY'S IMAGE (no jax / jaxlib here, no network); it documents the binding a
maintainer adds on the reference side and is written against jax >= 0.4.31 (jax.ffi).  The ctypes path
(gp_vae_amd/_lib.py + ops.py) is the one the tests exercise.

    import gpvae_jax as gj
    gj.register("/path/to/libgpvae_ffi.so")
    # src/model/gp.py:44            exp_kernel2 = gj.exp_kernel2
    # gp.py:190-193                 f = gj.mvn_sample(key_normals, k)   instead of dist.MultivariateNormal(...).sample
    # vae.py:98-137 under SVI       loss, grads = gj.elbo_and_grads(flat_params, batch, eps_z, dims)
    # inference.py:246              f = gj.decoder(dec_flat, z, dims, n)          (custom_vjp: NUTS differentiates it)
    # inference.py:400-424 to NUTS  NUTS(potential_fn=lambda q: gj.gp_potential(x, q, y)[0]) via custom_vjp below
"""
import ctypes
import functools

import jax
import jax.numpy as jnp
import numpy as np

_TARGETS = ["gram_se", "gram_agg", "potrf", "draw", "draw_batched", "randn", "hyper_sample", "krig_predict",
            "mvn_logp", "gp_potential", "vae_elbo", "encoder_fwd", "adam", "decoder_fwd", "decoder_vjp",
            "vae_potential", "poisson_exp"]


def register(path):
    lib = ctypes.CDLL(path)
    for name in _TARGETS:
        jax.ffi.register_ffi_target("gpvae_" + name, jax.ffi.pycapsule(getattr(lib, "gpvae_%s_ffi" % name)),
                                    platform="ROCM")


def _sds(shape, dtype):
    return jax.ShapeDtypeStruct(tuple(shape), dtype)


def exp_kernel2(x, z, var, ls, jitter=1.0e-5):
    """Drop-in for src/model/gp.py:44-81 (1-D inputs are columns, jitter iff square)."""
    x = x.reshape(-1, 1) if x.ndim == 1 else x
    z = z.reshape(-1, 1) if z.ndim == 1 else z
    var, ls = jnp.reshape(var, (1,)).astype(x.dtype), jnp.reshape(ls, (1,)).astype(x.dtype)
    k = jax.ffi.ffi_call("gpvae_gram_se", _sds((1, x.shape[0], z.shape[0]), x.dtype))(x, z, var, ls, jitter=np.float64(jitter))
    return k[0]


def cholesky(k, nan_on_fail=True):
    """Lower Cholesky factor of k [..., n, n] (what MultivariateNormal(covariance_matrix=k) computes)."""
    a = k.reshape((-1,) + k.shape[-2:])
    empty = jnp.zeros((a.shape[0], 0, a.shape[-1]), k.dtype)
    L, _, info = jax.ffi.ffi_call(
        "gpvae_potrf", (_sds(a.shape, k.dtype), _sds(empty.shape, k.dtype), _sds((a.shape[0],), jnp.int32)),
        input_output_aliases={0: 0, 1: 1})(a, empty, flags=np.int32(1 | (2 if nan_on_fail else 0)))
    return L.reshape(k.shape), info


def mvn_sample(eps, k, eps2=None, sigma=0.0):
    """f = L eps (and y = f + sigma eps2) for eps [nb, n] — gp.py:190-194 under Predictive."""
    L, _ = cholesky(k)
    eps2 = jnp.zeros_like(eps) if eps2 is None else eps2
    sig = jnp.reshape(sigma, (1,)).astype(k.dtype)
    f, y = jax.ffi.ffi_call("gpvae_draw", (_sds(eps.shape, k.dtype), _sds(eps.shape, k.dtype)))(L, eps, eps2, sig)
    return f, y


def _elbo_call(flat, batch, eps_z, dims, obs_sigma, with_grads):
    h0, h1, _ = dims
    return jax.ffi.ffi_call("gpvae_vae_elbo", (_sds((1,), batch.dtype), _sds(flat.shape, batch.dtype)))(
        flat, batch, eps_z, h0=np.int32(h0), h1=np.int32(h1), obs_sigma=np.float64(obs_sigma), with_grads=bool(with_grads))


def elbo_and_grads(flat, batch, eps_z, dims, obs_sigma=0.1):
    """-ELBO of vae_model / vae_guide under Trace_ELBO (vae.py:98-137, 213-218) and its gradient wrt the flat
    parameter vector (layout: include/gpvae.h)."""
    loss, grads = _elbo_call(flat, batch, eps_z, dims, obs_sigma, True)
    return loss[0], grads


@functools.partial(jax.custom_vjp, nondiff_argnums=(3, 4))
def neg_elbo(flat, batch, eps_z, dims, obs_sigma=0.1):
    """Scalar loss that jax.grad / numpyro.optim can differentiate: the backward pass is the library's."""
    return _elbo_call(flat, batch, eps_z, dims, obs_sigma, False)[0][0]


def _neg_elbo_fwd(flat, batch, eps_z, dims, obs_sigma):
    loss, grads = _elbo_call(flat, batch, eps_z, dims, obs_sigma, True)
    return loss[0], (grads, batch, eps_z)


def _neg_elbo_bwd(dims, obs_sigma, res, ct):
    grads, batch, eps_z = res
    return (ct * grads, jnp.zeros_like(batch), jnp.zeros_like(eps_z))  # data and noise are not differentiated


neg_elbo.defvjp(_neg_elbo_fwd, _neg_elbo_bwd)


def adam_update(flat, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """numpyro.optim.Adam's step (vae.py:165, 212) in place on (params, m, v)."""
    return jax.ffi.ffi_call("gpvae_adam", (_sds(flat.shape, flat.dtype),) * 3, input_output_aliases={0: 0, 2: 1, 3: 2})(
        flat, grads, m, v, t=np.int64(t), lr=np.float64(lr), b1=np.float64(b1), b2=np.float64(b2), eps=np.float64(eps))


def _dec_fwd(dec_flat, z, dims, n):
    h0, h1, _ = dims
    return jax.ffi.ffi_call("gpvae_decoder_fwd", _sds((z.shape[0], n), z.dtype))(dec_flat, z, h0=np.int32(h0), h1=np.int32(h1))


@functools.partial(jax.custom_vjp, nondiff_argnums=(2, 3))
def decoder(dec_flat, z, dims, n):
    """decoder(decoder_params, z) of VAEInference.regression (inference.py:246), batched over chains."""
    return _dec_fwd(dec_flat, z, dims, n)


def _decoder_fwd(dec_flat, z, dims, n):
    return _dec_fwd(dec_flat, z, dims, n), (dec_flat, z)


def _decoder_bwd(dims, n, res, dF):
    dec_flat, z = res
    h0, h1, _ = dims
    dz = jax.ffi.ffi_call("gpvae_decoder_vjp", _sds(z.shape, z.dtype))(dec_flat, z, dF, h0=np.int32(h0), h1=np.int32(h1))
    return (jnp.zeros_like(dec_flat), dz)  # the decoder's weights are constants of the MCMC model


decoder.defvjp(_decoder_fwd, _decoder_bwd)


def gp_potential(x, q, y_lik, lik_idx=None, jitter=1.0e-5):
    """(U [chains], dU/dq [chains, n + 3]) of GPInference.gp_regression / regression in NumPyro's unconstrained
    space (inference.py:359-424): hand `lambda q: U` with this gradient (custom_vjp) to NUTS as potential_fn."""
    x = x.reshape(-1, 1) if x.ndim == 1 else x
    idx = jnp.zeros((0,), jnp.int64) if lik_idx is None else jnp.asarray(lik_idx, jnp.int64)
    U, dU, _ = jax.ffi.ffi_call("gpvae_gp_potential", (_sds((q.shape[0],), q.dtype), _sds(q.shape, q.dtype),
                                                        _sds((q.shape[0],), jnp.int32)))(x, q, y_lik, idx, jitter=np.float64(jitter))
    return U, dU


def vae_potential(dec_flat, q, y_obs, obs_idx, n, dims, noise_scale=0.1):
    """(U [chains], dU/dq [chains, z + 1]) of VAEInference.regression (inference.py:243-254), one launch."""
    h0, h1, _ = dims
    return jax.ffi.ffi_call("gpvae_vae_potential", (_sds((q.shape[0],), q.dtype), _sds(q.shape, q.dtype)))(
        dec_flat, q, y_obs, jnp.asarray(obs_idx, jnp.int64), n=np.int64(n), h0=np.int32(h0), h1=np.int32(h1),
        noise_scale=np.float64(noise_scale))


def lgcp_observation(f, seed=0, stream_id=0, first_index=0):
    """(rate, y): rate = exp(f), y ~ Poisson(rate) int32 — the last two lines of PoiGP.sample (gp.py:249-250)."""
    return jax.ffi.ffi_call("gpvae_poisson_exp", (_sds(f.shape, f.dtype), _sds(f.shape, jnp.int32)))(
        f, seed=np.int64(seed), stream_id=np.int64(stream_id), first_index=np.int64(first_index))


def potential_fn(call):
    """Wrap a (q -> (U, dU)) pair for a single chain as a differentiable scalar function for numpyro's NUTS."""
    @jax.custom_vjp
    def u(q):
        return call(q[None, :])[0][0]

    def fwd(q):
        U, dU = call(q[None, :])
        return U[0], dU[0]

    u.defvjp(fwd, lambda dU, ct: (ct * dU,))
    return u
