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
    """scale_tril = cholesky(k): clean lower factor; all-NaN when k is not PD (jnp.linalg.cholesky)."""
    k = np.asarray(k)
    try:
        return np.linalg.cholesky(k).astype(k.dtype)
    except np.linalg.LinAlgError:
        return np.full_like(k, np.nan)


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
