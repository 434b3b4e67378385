"""CPU oracle for the gp-vae hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (gp_vae_amd/) never does and has no CPU fallback.

What it is: a NumPy/SciPy restatement, operation by operation, of the arithmetic the reference
(/root/reference, JAX + NumPyro) performs on the path SURVEY.md §8 names.  Each function cites the
reference file:line it follows.  The arithmetic the reference delegates to un-vendored third-party
code (jax.numpy / XLA linalg, jax.experimental.stax, numpyro.distributions, numpyro.infer.SVI,
Trace_ELBO, optim.Adam — numpyro~0.6.0 / jax~0.2.10 by the hint at archive/func_vae.py:23; versions
are not pinned by the repo) is restated from those libraries' published definitions.

PARITY UNPINNED: the reference ships no golden vectors, no value-asserting tests and no run logs
for this path, and JAX/NumPyro are not installable here (no network), so the
oracle cannot be checked against reference outputs.  It is pinned instead by closed-form
identities and SciPy cross-checks in tests/test_oracle.py (symmetry, L L^T = K, finite-difference
gradients, scipy.stats logpdf, Adam hand trace, Philox known-answer vectors from the Random123
distribution's kat_vectors).

dtype: every function computes in the dtype of its array inputs (float32 mirrors the reference's
JAX default; float64 is the tight-tolerance variant).
"""
import numpy as np
import scipy.linalg as sla

# --------------------------------------------------------------------------------------------------
# a1  exp_kernel2  (src/model/gp.py:30-41 check_d, 44-81)
# --------------------------------------------------------------------------------------------------


def _col(x):
    x = np.asarray(x)
    return x.reshape(x.shape[0], 1) if x.ndim == 1 else x  # gp.py:36-39


def exp_kernel2(x, z, var, ls, jitter=1.0e-5):
    """k = var*exp(-0.5*(||x-z||_2/ls)^2); += jitter*I iff k is square (gp.py:74-80)."""
    x, z = np.asarray(x), np.asarray(z)
    assert x.ndim == z.ndim and x.ndim in (1, 2)  # gp.py:34,67-68
    x, z = _col(x), _col(z)
    dt = x.dtype.type
    diff = x[:, None, :] - z[None, :, :]
    delta = np.sqrt(np.sum(diff * diff, axis=2, dtype=x.dtype))  # jnp.linalg.norm(ord=2, axis=2)
    q = delta / dt(ls)
    k = dt(var) * np.exp(dt(-0.5) * (q * q))  # jnp.power(deltaX/ls, 2.0)
    if k.shape[0] == k.shape[1]:  # gp.py:79
        k = k + dt(jitter) * np.eye(x.shape[0], dtype=x.dtype)
    return k.astype(x.dtype)


# --------------------------------------------------------------------------------------------------
# a2  agg_kernel_grid  (src/model/gp.py:85-127), with the Uniform draws an input
# --------------------------------------------------------------------------------------------------


def agg_points(u, grid1, grid2=None, d=1):
    """Sample locations _x = u/n + cell origin (gp.py:103-113).  u: [n^d, m, d] Uniform(0,1) draws."""
    grid1 = np.asarray(grid1)
    n = grid1.shape[0]  # gp.py:99 (the reference's `shape()` typo fixed)
    if d == 1:
        return u / n + grid1[:, None, None]  # gp.py:107
    if grid2 is None:
        grid2 = grid1
    uu, vv = np.meshgrid(grid1, np.asarray(grid2))  # gp.py:112
    org = np.array([[uu.flatten()], [vv.flatten()]]).transpose((2, 1, 0))  # gp.py:113 -> [n^2,1,2]
    return u / n + org


def agg_kernel_from_points(pts, var, ls, jitter=1.0e-6):
    """K*[i,j] = sum(kernel(pts_i, pts_j)) / m^2, the kernel adding jitter*I_m to EVERY (square)
    m x m block (gp.py:119-127) — i.e. + jitter/m on every entry."""
    pts = np.asarray(pts)
    nc, m, _ = pts.shape
    out = np.empty((nc, nc), dtype=pts.dtype)
    for i in range(nc):
        for j in range(nc):
            out[i, j] = np.sum(exp_kernel2(pts[i], pts[j], var, ls, jitter)) / pts.dtype.type(m ** 2)
    return out


# --------------------------------------------------------------------------------------------------
# a3/a4  MultivariateNormal(covariance_matrix=k) draw + Normal(f, sigma)  (gp.py:188-194)
# --------------------------------------------------------------------------------------------------


def cholesky_lower(k):
    """scale_tril = cholesky(k): clean lower factor; all-NaN when k is not PD (jnp.linalg.cholesky).
    LAPACK potrf IN THE DTYPE OF k (spotrf for float32, as jaxlib's CPU kernel calls it) — numpy.linalg.cholesky
    is not used: it computes float32 input in float64 and rounds the result, which is not what a float32
    program gets (at n = 4096 its backward error is 15x below spotrf's, and it factors the nearly rank-one
    Gram matrices of the length-scale prior's tail on which spotrf reports info > 0)."""
    k = np.asarray(k)
    if k.ndim > 2:
        return np.stack([cholesky_lower(a) for a in k])
    if k.dtype == np.float32:
        c, info = sla.lapack.spotrf(k, lower=1, clean=1)
    else:
        c, info = sla.lapack.dpotrf(np.asarray(k, np.float64), lower=1, clean=1)
    if info != 0:
        return np.full_like(k, np.nan)
    return c.astype(k.dtype)


def inv(a):
    """jnp.linalg.inv in the dtype of a (LAPACK getrf + getri; numpy.linalg.inv would compute float32 in float64)."""
    a = np.asarray(a)
    return sla.inv(a, check_finite=False).astype(a.dtype)


def potrf_info(k):
    """LAPACK-style info of a lower Cholesky: 0, or the order of the first non-PD leading minor."""
    k = np.asarray(k, dtype=np.float64)
    _, info = sla.lapack.dpotrf(k, lower=1)
    return int(info)


def gp_draw(L, eps, sigma=None, eps2=None):
    """f = L eps (per draw: F = E L^T); y = f + sigma*eps'  (gp.py:190-194).  Returns (F, Y)."""
    F = (eps @ L.T).astype(L.dtype)
    if eps2 is None:
        return F, F
    return F, (F + L.dtype.type(sigma) * eps2).astype(L.dtype)


def right_solve_lt(L, X):
    """X L^-T (rows of X solved against L^T) — what gpvae_potrf leaves in its trailing rows."""
    return sla.solve_triangular(L, X.T, lower=True).T.astype(L.dtype)


# --------------------------------------------------------------------------------------------------
# f3  Philox4x32-10 + Box-Muller (stand-in for jax.random.normal; same construction as draw.hip)
# --------------------------------------------------------------------------------------------------

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: [N,4] uint32, key: (k0,k1).  Random123 Philox4x32 with 10 rounds."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c = [n0, n1, n2, n3]
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def randn(count, seed=0, stream_id=0, dtype=np.float32):
    nq = (count + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    ctr = np.stack([q & _MASK, q >> np.uint64(32),
                    np.full(nq, stream_id & 0xFFFFFFFF, dtype=np.uint64),
                    np.full(nq, (stream_id >> 32) & 0xFFFFFFFF, dtype=np.uint64)], axis=1)
    w = philox4x32_10(ctr.astype(np.uint32), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    dt = np.dtype(dtype).type
    u = (w.astype(dtype) + dt(0.5)) * dt(2.0 ** -32)
    out = np.empty((nq, 4), dtype=dtype)
    for h in range(2):
        rad = np.sqrt(dt(-2.0) * np.log(u[:, 2 * h]))
        ang = dt(2.0) * u[:, 2 * h + 1]
        out[:, 2 * h] = rad * np.cos(np.pi * ang.astype(np.float64)).astype(dtype)
        out[:, 2 * h + 1] = rad * np.sin(np.pi * ang.astype(np.float64)).astype(dtype)
    return out.reshape(-1)[:count]


# --------------------------------------------------------------------------------------------------
# a9  GPInference.gp_prediction  (src/model/inference.py:427-460), literal
# --------------------------------------------------------------------------------------------------


def gp_prediction(x, y, obs_idx, var, ls, sigma, xi=None, jitter=1.0e-5, kernel=exp_kernel2):
    """Returns (mean, mean + noise, diagK).  xi stands in for jax.random.normal(rng_key, (n_test,))."""
    x, y = np.asarray(x), np.asarray(y)
    obs_idx = np.asarray(obs_idx)
    dt = x.dtype.type
    x_test = np.delete(x, obs_idx, axis=0)  # inference.py:442
    x_obs = x[obs_idx]  # inference.py:443
    k_pp = kernel(x_test, x_test, var, ls, jitter)  # inference.py:447
    k_px = kernel(x_test, x_obs, var, ls, jitter)  # inference.py:448
    k_xx = kernel(x_obs, x_obs, var, ls, jitter)  # inference.py:449
    k_xx_inv = inv(k_xx + np.eye(len(obs_idx), dtype=x.dtype) * dt(sigma))  # :450
    K = k_pp - k_px @ (k_xx_inv @ k_px.T)  # inference.py:451
    dK = np.clip(np.diag(K), 0.0, None)  # inference.py:452
    mean = k_px @ (k_xx_inv @ y[obs_idx])  # inference.py:455
    noise = np.sqrt(dK) * (xi if xi is not None else 0)
    return mean.astype(x.dtype), (mean + noise).astype(x.dtype), np.diag(K).astype(x.dtype)


# --------------------------------------------------------------------------------------------------
# a10 / f1  log N(f; 0, k) of gp_regression (src/model/inference.py:414-422) and its gradient
# --------------------------------------------------------------------------------------------------
def mvn_logp(x, f, var, ls, jitter=1.0e-5, kernel=exp_kernel2):
    """log MultivariateNormal(0, kernel(x, x, var, ls)).log_prob(f)  (inference.py:414,417-422)."""
    K = kernel(x, x, var, ls, jitter)
    L = cholesky_lower(K)
    w = sla.solve_triangular(L, np.asarray(f, K.dtype), lower=True)
    n = K.shape[0]
    return float(-0.5 * w @ w - np.log(np.diag(L)).sum() - 0.5 * n * np.log(2 * np.pi))


def mvn_logp_grads(x, f, var, ls, jitter=1.0e-5):
    """(logp, dlogp/df [n], dlogp/dvar, dlogp/dls): closed form of what reverse-mode through
    jnp.linalg.cholesky gives NUTS:  G = 1/2 (alpha alpha^T - K^-1),  d/dtheta = sum G * dK/dtheta."""
    x2 = _col(x)
    K = exp_kernel2(x, x, var, ls, jitter)
    n = K.shape[0]
    K0 = K - jitter * np.eye(n, dtype=K.dtype)
    Kinv = inv(K)
    alpha = Kinv @ np.asarray(f, K.dtype)
    G = 0.5 * (np.outer(alpha, alpha) - Kinv)
    D2 = ((x2[:, None, :] - x2[None, :, :]) ** 2).sum(-1)
    return mvn_logp(x, f, var, ls, jitter), -alpha, float((G * K0).sum() / var), float((G * K0 * D2).sum() / ls ** 3)


def gp_regression_potential(x, y, f, u_var, u_ls, u_sig, jitter=1.0e-5):
    """Potential energy (-log joint in NumPyro's unconstrained space, exp transforms) of
    GPInference.gp_regression (inference.py:407-424): kernel_var ~ LogNormal(0, 0.1),
    kernel_sigma ~ HalfNormal(0.1), kernel_length ~ InverseGamma(4, 1), f ~ MVN(0, k),
    y_pred ~ Normal(f, sigma) observed."""
    import math
    var, ls, sig = np.exp(u_var), np.exp(u_ls), np.exp(u_sig)
    n = len(f)
    lp = -math.log(0.1 * math.sqrt(2 * math.pi)) - u_var ** 2 / (2 * 0.1 ** 2)            # LogNormal + Jacobian
    lp += math.log(2.0) - 0.5 * math.log(2 * math.pi) - math.log(0.1) - 0.5 * (sig / 0.1) ** 2 + u_sig  # HalfNormal
    lp += 4 * math.log(1.0) - math.lgamma(4.0) - 4 * u_ls - 1.0 / ls                         # InverseGamma(4,1)
    lp += mvn_logp(x, f, var, ls, jitter)
    r = (np.asarray(y) - np.asarray(f)) / sig
    lp += float(-0.5 * r @ r - n * u_sig - 0.5 * n * math.log(2 * math.pi))
    return -lp


# --------------------------------------------------------------------------------------------------
# a5/a6  stax encoder/decoder + Trace_ELBO loss  (src/model/vae.py:72-137, 213-218)
# --------------------------------------------------------------------------------------------------

PARAM_NAMES = ("W1", "b1", "W2", "b2", "Wmu", "bmu", "Wsd", "bsd", "V1", "c1", "V2", "c2", "V3", "c3")


def param_shapes(n, h0, h1, z):
    return {"W1": (n, h0), "b1": (h0,), "W2": (h0, h1), "b2": (h1,), "Wmu": (h1, z), "bmu": (z,),
            "Wsd": (h1, z), "bsd": (z,), "V1": (z, h1), "c1": (h1,), "V2": (h1, h0), "c2": (h0,),
            "V3": (h0, n), "c3": (n,)}


def unflatten(flat, n, h0, h1, z):
    out, o = {}, 0
    for k in PARAM_NAMES:
        shp = param_shapes(n, h0, h1, z)[k]
        sz = int(np.prod(shp))
        out[k] = flat[o:o + sz].reshape(shp)
        o += sz
    assert o == flat.size
    return out


def flatten(p):
    return np.concatenate([np.asarray(p[k]).reshape(-1) for k in PARAM_NAMES])


def init_params(rng, n, h0, h1, z, dtype=np.float32):
    """stax.randn() W_init and the default bias init are both 1e-2 * N(0,1) (vae.py:76-95)."""
    return {k: (1e-2 * rng.standard_normal(s)).astype(dtype) for k, s in param_shapes(n, h0, h1, z).items()}


def encoder(p, y):
    """stax.serial(Dense, Relu, Dense, Relu, FanOut(2), parallel(Dense, serial(Dense, Exp))) vae.py:75-85"""
    h1_ = np.maximum(y @ p["W1"] + p["b1"], 0)
    h2_ = np.maximum(h1_ @ p["W2"] + p["b2"], 0)
    return h2_ @ p["Wmu"] + p["bmu"], np.exp(h2_ @ p["Wsd"] + p["bsd"])


def decoder(p, zz):
    """stax.serial(Dense(h1), Relu, Dense(h0), Relu, Dense(out_dim))  vae.py:90-96"""
    d1 = np.maximum(zz @ p["V1"] + p["c1"], 0)
    d2 = np.maximum(d1 @ p["V2"] + p["c2"], 0)
    return d2 @ p["V3"] + p["c3"]


def decoder_from_list(params_list, zz):
    """decoder(decoder_params, z) with the stax list layout [(W,b),(),(W,b),(),(W,b)] (fixtures §4)."""
    (V1, c1), _, (V2, c2), _, (V3, c3) = params_list
    return decoder({"V1": V1, "c1": c1, "V2": V2, "c2": c2, "V3": V3, "c3": c3}, zz)


_HALF_LOG_2PI = 0.9189385332046727


def _normal_logpdf(v, loc, scale):
    return -0.5 * ((v - loc) / scale) ** 2 - np.log(scale) - _HALF_LOG_2PI


def elbo_loss(p, y, eps_z, obs_sigma=0.1):
    """-ELBO, one particle, summed over batch and dims (Trace_ELBO with vae_guide then vae_model
    replayed on the guide's z; vae.py:98-137).  eps_z stands in for the guide's Normal draw."""
    dt = y.dtype.type
    z_loc, z_std = encoder(p, y)
    zz = z_loc + z_std * eps_z  # z ~ Normal(z_loc, z_std)          vae.py:135
    log_q = np.sum(_normal_logpdf(zz, z_loc, z_std))
    log_pz = np.sum(_normal_logpdf(zz, dt(0), dt(1)))  # vae.py:112-114
    f = decoder(p, zz)  # vae.py:115
    log_py = np.sum(_normal_logpdf(y, f, dt(obs_sigma)))  # vae.py:117
    return -(log_py + log_pz - log_q)


def elbo_loss_and_grads(p, y, eps_z, obs_sigma=0.1):
    """Loss and hand-derived reverse-mode gradients (what value_and_grad gives svi.update)."""
    dt = y.dtype.type
    A1 = y @ p["W1"] + p["b1"]; H1 = np.maximum(A1, 0)
    A2 = H1 @ p["W2"] + p["b2"]; H2 = np.maximum(A2, 0)
    MU = H2 @ p["Wmu"] + p["bmu"]; LS = H2 @ p["Wsd"] + p["bsd"]; SD = np.exp(LS)
    Z = MU + SD * eps_z
    G1 = Z @ p["V1"] + p["c1"]; D1 = np.maximum(G1, 0)
    G2 = D1 @ p["V2"] + p["c2"]; D2 = np.maximum(G2, 0)
    F = D2 @ p["V3"] + p["c3"]
    s2 = dt(obs_sigma) ** 2
    loss = (np.sum(0.5 * (y - F) ** 2 / s2) + y.size * (np.log(obs_sigma) + _HALF_LOG_2PI)
            + np.sum(0.5 * Z * Z - 0.5 * eps_z * eps_z - LS))
    dF = (F - y) / s2
    g = {}
    g["V3"] = D2.T @ dF; g["c3"] = dF.sum(0); dD2 = dF @ p["V3"].T
    dG2 = dD2 * (G2 > 0); g["V2"] = D1.T @ dG2; g["c2"] = dG2.sum(0); dD1 = dG2 @ p["V2"].T
    dG1 = dD1 * (G1 > 0); g["V1"] = Z.T @ dG1; g["c1"] = dG1.sum(0); dZ = dG1 @ p["V1"].T + Z
    dMU = dZ; dLS = dZ * eps_z * SD - 1
    g["Wmu"] = H2.T @ dMU; g["bmu"] = dMU.sum(0); g["Wsd"] = H2.T @ dLS; g["bsd"] = dLS.sum(0)
    dH2 = dMU @ p["Wmu"].T + dLS @ p["Wsd"].T
    dA2 = dH2 * (A2 > 0); g["W2"] = H1.T @ dA2; g["b2"] = dA2.sum(0); dH1 = dA2 @ p["W2"].T
    dA1 = dH1 * (A1 > 0); g["W1"] = y.T @ dA1; g["b1"] = dA1.sum(0)
    return loss, {k: v.astype(y.dtype) for k, v in g.items()}


def decoder_vjp(p, zz, dF):
    """dZ = (d decoder / d z)^T dF, per row."""
    G1 = zz @ p["V1"] + p["c1"]; D1 = np.maximum(G1, 0)
    G2 = D1 @ p["V2"] + p["c2"]
    dD2 = dF @ p["V3"].T
    dD1 = (dD2 * (G2 > 0)) @ p["V2"].T
    return (dD1 * (G1 > 0)) @ p["V1"].T


# --------------------------------------------------------------------------------------------------
# a7  Adam as numpyro.optim.Adam -> jax.experimental.optimizers.adam  (vae.py:212)
# --------------------------------------------------------------------------------------------------


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """One update, step t counts from 1 (jax: i = t-1; mhat = m/(1-b1^(i+1)))."""
    dt = p.dtype.type
    m = (1 - dt(b1)) * g + dt(b1) * m
    v = (1 - dt(b2)) * g * g + dt(b2) * v
    mhat = m / dt(1 - b1 ** t)
    vhat = v / dt(1 - b2 ** t)
    p = p - dt(lr) * mhat / (np.sqrt(vhat) + dt(eps))
    return p.astype(m.dtype), m, v


# --------------------------------------------------------------------------------------------------
# hyper-priors of GP.sample / epoch_train  (gp.py:171-176, vae.py:157-159), host-side draws
# --------------------------------------------------------------------------------------------------


def draw_hyper(rng):
    """(ls, var, sigma): InverseGamma(1, 0.1), LogNormal(0, 0.1), HalfNormal(0.1)."""
    ls = 0.1 / rng.gamma(1.0)  # InverseGamma(a, b) = b / Gamma(a, 1)
    var = float(np.exp(0.1 * rng.standard_normal()))
    sigma = abs(0.1 * rng.standard_normal())
    return float(ls), var, float(sigma)


def _hyper_words(idx, purpose, attempt, seed, stream_id):
    ctr = np.array([[idx & 0xFFFFFFFF, ((purpose << 24) | attempt) & 0xFFFFFFFF, stream_id & 0xFFFFFFFF,
                     (stream_id >> 32) & 0xFFFFFFFF]], dtype=np.uint32)
    return [int(w) for w in philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[0]]


def _unif(w):
    return (float(w) + 0.5) / 4294967296.0


def _philox_gamma(a, idx, purpose, seed, stream_id):
    """Marsaglia & Tsang (2000) on Philox words, as gpvae_hyper_sample's device sampler does it."""
    import math
    a1 = a + 1.0 if a < 1.0 else a
    d = a1 - 1.0 / 3.0
    cc = 1.0 / math.sqrt(9.0 * d)
    g = d
    for k in range(64):
        c = _hyper_words(idx, purpose, k, seed, stream_id)
        x = math.sqrt(-2.0 * math.log(_unif(c[0]))) * math.cos(2.0 * math.pi * _unif(c[1]))
        t = 1.0 + cc * x
        v = t * t * t
        if v > 0.0 and math.log(_unif(c[2])) < 0.5 * x * x + d - d * v + d * math.log(v):
            g = d * v
            break
    if a < 1.0:
        g *= _unif(_hyper_words(idx, purpose + 1, 0, seed, stream_id)[0]) ** (1.0 / a)
    return g


def hyper_sample(count, seed=0, stream_id=0, first_index=0, ls_prior="invgamma", ls_a=1.0, ls_b=0.1,
                 var_scale=0.1, sigma_scale=0.1):
    """(ls, var, sigma) [count] (float64): the hyper-priors of gp.py:171-176 / vae.py:157-159 drawn from
    counter-based Philox words — var ~ LogNormal(0, var_scale), sigma ~ HalfNormal(sigma_scale),
    ls ~ InverseGamma(a, b) = b / Gamma(a) or Beta(a, b) = Ga / (Ga + Gb) (README.md:17)."""
    import math
    ls, var, sig = np.empty(count), np.empty(count), np.empty(count)
    for i in range(count):
        idx = first_index + i
        c = _hyper_words(idx, 0, 0, seed, stream_id)
        rad = math.sqrt(-2.0 * math.log(_unif(c[0])))
        var[i] = math.exp(var_scale * rad * math.cos(2.0 * math.pi * _unif(c[1])))
        sig[i] = abs(sigma_scale * rad * math.sin(2.0 * math.pi * _unif(c[1])))
        if ls_prior == "invgamma":
            ls[i] = ls_b / _philox_gamma(ls_a, idx, 1, seed, stream_id)
        else:
            ga, gb = _philox_gamma(ls_a, idx, 1, seed, stream_id), _philox_gamma(ls_b, idx, 3, seed, stream_id)
            ls[i] = ga / (ga + gb)
    return ls, var, sig


def _poisson_one(lam, idx, seed, stream_id):
    """One Poisson(lam) draw from the Philox words of counter (idx, 7 << 24 | attempt): inversion by sequential
    search below 10, Hoermann's PTRS (1993; NumPy's and JAX's method) from 10 up — gpvae_poisson_exp's sampler."""
    import math
    if not lam >= 0.0:
        return -1
    if lam == 0.0:
        return 0
    if lam > 2147483647.0:
        return 2147483647
    if lam < 10.0:
        u = _unif(_hyper_words(idx, 7, 0, seed, stream_id)[0])
        p = math.exp(-lam)
        s, k = p, 0
        while u > s and k < 200:
            k += 1
            p *= lam / k
            s += p
        return k
    slam, loglam = math.sqrt(lam), math.log(lam)
    b = 0.931 + 2.53 * slam
    a = -0.059 + 0.02483 * b
    inv_alpha = 1.1239 + 1.1328 / (b - 3.4)
    vr = 0.9277 - 3.6224 / (b - 2.0)
    for att in range(64):
        c = _hyper_words(idx, 7, att, seed, stream_id)
        U, V = _unif(c[0]) - 0.5, _unif(c[1])
        us = 0.5 - abs(U)
        kf = math.floor((2.0 * a / us + b) * U + lam + 0.43)
        if us >= 0.07 and V <= vr:
            return int(min(kf, 2147483647))
        if kf < 0 or (us < 0.013 and V > us):
            continue
        if math.log(V) + math.log(inv_alpha) - math.log(a / (us * us) + b) <= -lam + kf * loglam - math.lgamma(kf + 1.0):
            return int(min(kf, 2147483647))
    return int(min(math.floor(lam), 2147483647))


def poisson_exp(f, seed=0, stream_id=0, first_index=0, rate=None):
    """rate = exp(f) (in f's dtype) and y ~ Poisson(rate), int32: the observation of PoiGP.sample
    (gp.py:249-250: `rate = exp(f)`, `y ~ Poisson(rate)`), element i from Philox counter first_index + i.
    `rate` overrides exp(f) (tests hand over the device's rates so that the draws compare bit for bit)."""
    f = np.asarray(f)
    with np.errstate(over="ignore"):
        rt = np.exp(f.astype(np.float64)).astype(f.dtype) if rate is None else np.asarray(rate)
    y = np.array([_poisson_one(float(r), first_index + i, seed, stream_id) for i, r in enumerate(rt.reshape(-1))],
                 dtype=np.int32).reshape(rt.shape)
    return rt, y


def poigp_sample(pts, var, ls, eps, jitter, seed=0, stream_id=0):
    """PoiGP.sample with the hyper-parameters given (gp.py:234-250): k = aggregated kernel of the cells'
    sample points, f = chol(k) eps (MultivariateNormal's draw), rate = exp(f), y ~ Poisson(rate).
    pts [cells, m, d], eps [B, cells].  Returns f, rate, y."""
    k = agg_kernel_from_points(pts, var, ls, jitter)
    f = gp_draw(cholesky_lower(k), eps)[0]
    rate, y = poisson_exp(f, seed, stream_id)
    return f, rate, y


def train_step(p, m, v, t, x, hyper, eps, eps2, eps_z, lr, jitter, obs_sigma=0.1):
    """One body_fn of VAE.epoch_train (vae.py:150-167) with every random input explicit."""
    ls, var, sigma = hyper
    K = exp_kernel2(x, x, var, ls, jitter)  # gp.py:188
    L = cholesky_lower(K)
    _, Y = gp_draw(L, eps, sigma, eps2)  # gp.py:190-194 under Predictive
    loss, g = elbo_loss_and_grads(p, Y, eps_z, obs_sigma)
    out_p, out_m, out_v = {}, {}, {}
    for k in PARAM_NAMES:
        out_p[k], out_m[k], out_v[k] = adam_step(p[k], g[k], m[k], v[k], t, lr)
    return loss, out_p, out_m, out_v


# --------------------------------------------------------------------------------------------------
# a11  VAEInference.regression potential (src/model/inference.py:243-254), unconstrained space
# --------------------------------------------------------------------------------------------------


def vae_potential(p, zz, log_sigma, y, obs_idx, s0=0.1):
    """-log joint of (noise ~ HalfNormal(s0), z ~ N(0,I), y[obs] ~ Normal(decoder(z)[obs], noise))
    with noise = exp(log_sigma) and the log-Jacobian of that transform (NumPyro's potential)."""
    sigma = np.exp(log_sigma)
    f = decoder(p, zz)
    r = f[..., obs_idx] - y[obs_idx]
    log_lik = np.sum(_normal_logpdf(y[obs_idx], f[..., obs_idx], sigma[..., None]), axis=-1)
    log_pz = np.sum(_normal_logpdf(zz, 0.0, 1.0), axis=-1)
    log_ps = np.log(2.0) - _HALF_LOG_2PI - np.log(s0) - 0.5 * (sigma / s0) ** 2 + log_sigma
    del r
    return -(log_lik + log_pz + log_ps)
