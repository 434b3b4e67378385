"""CPU tests of the oracle itself: golden vectors + closed-form identities (SURVEY.md §8c list).

The reference has no value-asserting tests, so this file is what pins the restatement."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.stats as sst

from oracle import gpvae_oracle as orc


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ---- golden vectors ----------------------------------------------------------------------------------
def test_golden_gram(golden):
    g = golden
    assert rel(orc.exp_kernel2(g["gram_x16"], g["gram_x16"], 1.3, 0.15, 1e-4), g["gram_k16"]) < 1e-14
    assert rel(orc.exp_kernel2(g["gram_x44"], g["gram_x44"], 0.9, 0.4, 1e-3), g["gram_k44"]) < 1e-14
    assert rel(orc.exp_kernel2(g["gram_xa"], g["gram_xb"], 1.1, 0.3, 1e-2), g["gram_rect"]) < 1e-14
    assert rel(orc.exp_kernel2(g["gram_xc"], g["gram_xd"], 1.0, 0.5, 1e-2), g["gram_sqdiff"]) < 1e-14


def test_golden_chain(golden):
    g = golden
    for n in (8, 64):
        var, ls, jit = g["chol_n%d_hyp" % n]
        x = g["chol_n%d_x" % n]
        assert rel(orc.cholesky_lower(orc.exp_kernel2(x, x, var, ls, jit)), g["chol_n%d_L" % n]) < 1e-12
    F, Y = orc.gp_draw(g["chol_n64_L"], g["draw_E"], 0.07, g["draw_E2"])
    assert rel(F, g["draw_F"]) < 1e-14 and rel(Y, g["draw_Y"]) < 1e-14
    m, s, v = orc.gp_prediction(g["krig_x"], g["krig_y"], g["krig_obs"], *g["krig_hyp"], g["krig_xi"])
    assert rel(m, g["krig_mean"]) < 1e-10 and rel(s, g["krig_sample"]) < 1e-10 and rel(v, g["krig_var"]) < 1e-9
    n, h0, h1, z, B = g["elbo_dims"]
    p = orc.unflatten(g["elbo_params"], n, h0, h1, z)
    loss, gr = orc.elbo_loss_and_grads(p, g["elbo_Y"], g["elbo_eps"], 0.1)
    assert abs(loss - g["elbo_loss"]) < 1e-9 * abs(loss) and rel(orc.flatten(gr), g["elbo_grads"]) < 1e-13
    assert rel(orc.randn(37, 123456789012345, 9, np.float64), g["randn_f64"]) < 1e-15


# ---- a1 identities ---------------------------------------------------------------------------------------
def test_gram_identities():
    x = np.arange(0, 1, 1 / 16)
    k = orc.exp_kernel2(x, x, 1.3, 0.2, 1e-4)
    assert np.allclose(k, k.T) and np.allclose(np.diag(k), 1.3 + 1e-4)
    assert np.allclose(orc.exp_kernel2(x, x, 1.3, 1e9, 1e-4), 1.3 + 1e-4 * np.eye(16))  # ls -> inf
    a, b = np.random.default_rng(0).random(5), np.random.default_rng(1).random(3)
    kr = orc.exp_kernel2(a, b, 1.0, 0.3, 0.5)  # rectangular: jitter must NOT appear
    assert np.allclose(kr, np.exp(-0.5 * ((a[:, None] - b[None]) / 0.3) ** 2))
    c, d = np.random.default_rng(2).random((4, 2)), np.random.default_rng(3).random((4, 2))
    ks = orc.exp_kernel2(c, d, 1.0, 0.3, 0.5)  # square with different points: jitter IS added (gp.py:79)
    base = np.exp(-0.5 * (np.linalg.norm(c[:, None] - d[None], axis=2) / 0.3) ** 2)
    assert np.allclose(ks, base + 0.5 * np.eye(4))
    assert orc.exp_kernel2(x.astype(np.float32), x.astype(np.float32), 1, 0.2).dtype == np.float32
    with pytest.raises(AssertionError):
        orc.exp_kernel2(np.zeros(3), np.zeros((3, 1)), 1, 1)  # gp.py:34


def test_agg_kernel_is_block_mean_plus_jitter_over_m():
    rng = np.random.default_rng(5)
    pts = orc.agg_points(rng.random((4, 3, 1)), np.arange(0, 1, 1 / 4), d=1)
    k = orc.agg_kernel_from_points(pts, 1.0, 0.3, 1e-2)
    want = np.array([[np.mean(np.exp(-0.5 * ((pts[i, :, None, 0] - pts[j, None, :, 0]) / 0.3) ** 2))
                      for j in range(4)] for i in range(4)]) + 1e-2 / 3
    assert np.allclose(k, want)
    p2 = orc.agg_points(rng.random((9, 2, 2)), np.arange(0, 1, 1 / 3), d=2)
    # cell order = meshgrid(grid1, grid2) flattened, point = (u, v)   (gp.py:112-113)
    assert np.all(p2[1, :, 0] >= 1 / 3) and np.all(p2[1, :, 1] < 1 / 3)


# ---- a3 -----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [8, 64, 300])
def test_cholesky_properties(n):
    x = np.arange(0, 1, 1 / n)[:n]
    k = orc.exp_kernel2(x, x, 1.0, 0.05, 1e-4)
    L = orc.cholesky_lower(k)
    assert rel(L @ L.T, k) < 1e-12 and np.all(np.triu(L, 1) == 0)
    assert rel(L, sla.cholesky(k, lower=True)) < 1e-13
    assert orc.potrf_info(k) == 0


def test_cholesky_non_pd_gives_nan_and_info():
    k = np.eye(6)
    k[3, 3] = -1.0
    assert np.all(np.isnan(orc.cholesky_lower(k)))
    assert orc.potrf_info(k) == 4


def test_draw_sample_covariance_matches_kernel():
    x = np.arange(0, 1, 1 / 12)
    k = orc.exp_kernel2(x, x, 1.0, 0.3, 1e-3)
    L = orc.cholesky_lower(k)
    E = np.random.default_rng(0).standard_normal((200000, 12))
    F, _ = orc.gp_draw(L, E)
    assert np.abs(F.T @ F / E.shape[0] - k).max() < 0.02
    Xs = orc.right_solve_lt(L, E[:5])
    assert rel(Xs @ L.T, E[:5]) < 1e-9


# ---- a9 ---------------------------------------------------------------------------------------------------------
def test_kriging_inverse_vs_cholesky_and_limits():
    rng = np.random.default_rng(1)
    x = np.arange(0, 1, 1 / 60)
    obs = np.sort(rng.permutation(60)[:15])
    y = np.sin(7 * x)
    m, s, v = orc.gp_prediction(x, y, obs, 1.0, 0.15, 0.02, None)
    xo, xt = x[obs], np.delete(x, obs)
    A = orc.exp_kernel2(xo, xo, 1.0, 0.15) + 0.02 * np.eye(15)  # +sigma, not sigma^2 (inference.py:450)
    L = np.linalg.cholesky(A)
    X = orc.right_solve_lt(L, orc.exp_kernel2(xt, xo, 1.0, 0.15))
    w = sla.solve_triangular(L, y[obs], lower=True)
    assert rel(X @ w, m) < 1e-9 and rel((1.0 + 1e-5) - np.sum(X * X, 1), v) < 1e-7
    assert np.all(v > -1e-9) and np.array_equal(m, s)  # no xi => sample == mean
    # interpolation: tiny noise reproduces the data at an observed point treated as a test point
    m2, _, v2 = orc.gp_prediction(np.r_[x[obs], x[obs][3] + 1e-9], np.r_[y[obs], 0.0], np.arange(15), 1.0, 0.15, 1e-8, None)
    assert abs(m2[0] - y[obs][3]) < 1e-3 and v2[0] < 1e-3


def test_kriging_square_kpx_gets_jitter(golden):
    g = golden
    x, y, obs = g["krigsq_x"], g["krigsq_y"], g["krigsq_obs"]
    m, _, v = orc.gp_prediction(x, y, obs, 1.0, 0.2, 0.05, None)
    assert rel(m, g["krigsq_mean"]) < 1e-10
    xo, xt = x[obs], np.delete(x, obs)
    kpx_plain = np.exp(-0.5 * ((xt[:, None] - xo[None]) / 0.2) ** 2)
    A = orc.exp_kernel2(xo, xo, 1.0, 0.2) + 0.05 * np.eye(10)
    m_plain = kpx_plain @ np.linalg.solve(A, y[obs])
    assert rel(m, m_plain) > 1e-7  # the literal rule (jitter on the square k_px) is observable


# ---- a5/a6/a7/a11 ---------------------------------------------------------------------------------------
def _small_problem(seed=0, n=12, h0=5, h1=4, z=3, B=4):
    rng = np.random.default_rng(seed)
    p = orc.init_params(rng, n, h0, h1, z, np.float64)
    for k in p:
        p[k] = p[k] * 30
    return p, rng.standard_normal((B, n)), rng.standard_normal((B, z)), (n, h0, h1, z)


def test_elbo_matches_logpdf_form_and_scipy():
    p, Y, ez, _ = _small_problem()
    l1 = orc.elbo_loss(p, Y, ez, 0.1)
    l2, _ = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    assert abs(l1 - l2) < 1e-9 * abs(l1)
    z_loc, z_std = orc.encoder(p, Y)
    zz = z_loc + z_std * ez
    want = -(sst.norm.logpdf(Y, orc.decoder(p, zz), 0.1).sum() + sst.norm.logpdf(zz).sum()
             - sst.norm.logpdf(zz, z_loc, z_std).sum())
    assert abs(l1 - want) < 1e-9 * abs(want)


def test_elbo_gradients_match_finite_differences():
    p, Y, ez, dims = _small_problem(3)
    _, g = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    flat, gflat = orc.flatten(p), orc.flatten(g)
    rng = np.random.default_rng(0)
    for i in rng.permutation(flat.size)[:60]:
        h = 1e-6 * max(1.0, abs(flat[i]))
        fp, fm = flat.copy(), flat.copy()
        fp[i] += h
        fm[i] -= h
        fd = (orc.elbo_loss(orc.unflatten(fp, *dims), Y, ez) - orc.elbo_loss(orc.unflatten(fm, *dims), Y, ez)) / (2 * h)
        assert abs(fd - gflat[i]) <= 2e-5 * max(1.0, abs(gflat[i]), abs(fd)), (i, fd, gflat[i])


def test_elbo_closed_form_with_zero_weights():
    n, h0, h1, z, B = 6, 3, 2, 2, 5
    p = {k: np.zeros(s) for k, s in orc.param_shapes(n, h0, h1, z).items()}
    rng = np.random.default_rng(1)
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    # mu = 0, sd = 1 => z = eps, log q = log p(z); f = 0 => loss = -sum log N(y; 0, 0.1)
    want = np.sum(0.5 * (Y / 0.1) ** 2) + Y.size * (np.log(0.1) + 0.5 * np.log(2 * np.pi))
    assert abs(orc.elbo_loss(p, Y, ez) - want) < 1e-9 * want


def test_adam_first_step_and_trace():
    p = np.array([1.0, -2.0, 0.5])
    g = np.array([0.3, -4.0, 1e-3])
    p1, m1, v1 = orc.adam_step(p, g, np.zeros(3), np.zeros(3), 1, 1e-3)
    assert np.allclose(p1, p - 1e-3 * g / (np.abs(g) + 1e-8))  # first step: -lr * g / (|g| + eps)
    m, v, q = np.zeros(3), np.zeros(3), p.copy()
    for t in range(1, 4):
        q, m, v = orc.adam_step(q, g, m, v, t, 1e-3)
    assert np.allclose(m, g * (1 - 0.9 ** 3)) and np.allclose(v, g * g * (1 - 0.999 ** 3))
    assert np.allclose(q, p - 3e-3 * g / (np.abs(g) + 1e-8))


def test_decoder_and_potential_gradients():
    p, _, _, (n, h0, h1, z) = _small_problem(4)
    rng = np.random.default_rng(2)
    Z, dF = rng.standard_normal((3, z)), rng.standard_normal((3, n))
    lst = [(p["V1"], p["c1"]), (), (p["V2"], p["c2"]), (), (p["V3"], p["c3"])]
    assert np.allclose(orc.decoder_from_list(lst, Z), orc.decoder(p, Z))
    dZ = orc.decoder_vjp(p, Z, dF)
    for i in range(z):
        h = 1e-6
        zp, zm = Z.copy(), Z.copy()
        zp[:, i] += h
        zm[:, i] -= h
        fd = np.sum((orc.decoder(p, zp) - orc.decoder(p, zm)) * dF, 1) / (2 * h)
        assert np.allclose(fd, dZ[:, i], rtol=1e-5, atol=1e-6)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    c = np.array([[0, 0, 0, 0], [0xFFFFFFFF] * 4, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32)
    assert list(orc.philox4x32_10(c[:1], (0, 0))[0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(orc.philox4x32_10(c[1:2], (0xFFFFFFFF, 0xFFFFFFFF))[0]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(orc.philox4x32_10(c[2:3], (0xA4093822, 0x299F31D0))[0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    zz = orc.randn(200000, 7, 3, np.float64)
    assert abs(zz.mean()) < 0.01 and abs(zz.std() - 1) < 0.01 and abs(sst.kurtosis(zz)) < 0.05


def test_hyper_priors_have_the_right_moments():
    rng = np.random.default_rng(0)
    d = np.array([orc.draw_hyper(rng) for _ in range(40000)])
    assert abs(np.median(d[:, 0]) - 0.1 / np.log(2)) < 0.01  # InverseGamma(1, .1) median
    assert abs(np.mean(np.log(d[:, 1]))) < 0.005 and abs(np.std(np.log(d[:, 1])) - 0.1) < 0.005
    assert abs(np.mean(d[:, 2]) - 0.1 * np.sqrt(2 / np.pi)) < 0.002  # HalfNormal(0.1) mean


def test_mvn_logp_matches_scipy_and_finite_differences():
    """f1 oracle pins: scipy.stats.multivariate_normal.logpdf and central
    finite differences of the closed-form gradient."""
    import scipy.stats
    rng = np.random.default_rng(3)
    for x in (np.arange(40) / 40.0, rng.uniform(0, 1, (30, 2))):
        n = x.shape[0]
        var, ls = 1.3, 0.21
        jit = 1e-3  # keeps cond(K) ~1e4 so that central differences are meaningful
        K = orc.exp_kernel2(x, x, var, ls, jit)
        f = np.linalg.cholesky(K) @ rng.standard_normal(n)
        lp, gf, gv, gl = orc.mvn_logp_grads(x, f, var, ls, jit)
        want = scipy.stats.multivariate_normal(np.zeros(n), K, allow_singular=False).logpdf(f)
        assert abs(lp - want) < 1e-8 * abs(want)
        h = 1e-5
        fd_v = (orc.mvn_logp(x, f, var + h, ls, jit) - orc.mvn_logp(x, f, var - h, ls, jit)) / (2 * h)
        fd_l = (orc.mvn_logp(x, f, var, ls + h, jit) - orc.mvn_logp(x, f, var, ls - h, jit)) / (2 * h)
        assert abs(gv - fd_v) < 1e-4 * max(1, abs(fd_v)) and abs(gl - fd_l) < 1e-4 * max(1, abs(fd_l))
        e = np.zeros(n); e[3] = h
        fd_f = (orc.mvn_logp(x, f + e, var, ls, jit) - orc.mvn_logp(x, f - e, var, ls, jit)) / (2 * h)
        assert abs(gf[3] - fd_f) < 1e-4 * max(1, abs(fd_f))


def test_philox_hyper_prior_samplers():
    """The counter-based hyper-prior samplers (restated for gpvae_hyper_sample): moments of
    InverseGamma / LogNormal / HalfNormal / Beta, batching independence, determinism."""
    ls, var, sig = orc.hyper_sample(6000, seed=3, stream_id=9)
    assert abs(np.median(ls) - 0.1 / np.log(2)) < 0.01                      # InverseGamma(1, .1)
    assert abs(np.std(np.log(var)) - 0.1) < 0.005 and abs(np.mean(np.log(var))) < 0.005
    assert abs(np.mean(sig) - 0.1 * np.sqrt(2 / np.pi)) < 0.003 and (sig >= 0).all()
    lb, _, _ = orc.hyper_sample(6000, seed=3, stream_id=9, ls_prior="beta", ls_a=0.1, ls_b=2.0)
    assert abs(lb.mean() - 0.1 / 2.1) < 0.006 and ((lb >= 0) & (lb <= 1)).all()  # Beta(.1, 2): mean a/(a+b)
    l4, _, _ = orc.hyper_sample(6000, seed=4, ls_a=4.0, ls_b=1.0)
    assert abs(l4.mean() - 1 / 3.0) < 0.01                                   # InverseGamma(4, 1): b/(a-1)
    a = orc.hyper_sample(10, seed=3, stream_id=9, first_index=100)
    assert all(np.array_equal(u, v[100:110]) for u, v in zip(a, orc.hyper_sample(110, seed=3, stream_id=9)))


def test_poisson_sampler_against_scipy():
    """The Philox Poisson sampler (inversion below 10, PTRS above; the observation of PoiGP.sample, gp.py:249-250):
    chi-square against scipy's pmf on both branches, moments at large rates, and the edge values."""
    for lam in (0.4, 4.0, 9.99, 10.0, 37.5):
        y = np.array([orc._poisson_one(lam, i, 11, 3) for i in range(12000)])
        hi = int(sst.poisson.ppf(1 - 1e-4, lam)) + 1
        obs = np.bincount(np.minimum(y, hi), minlength=hi + 1).astype(float)
        pmf = sst.poisson.pmf(np.arange(hi + 1), lam)
        pmf[hi] = sst.poisson.sf(hi - 1, lam)
        keep = pmf * len(y) >= 5
        obs_k, exp_k = np.append(obs[keep], obs[~keep].sum()), np.append(pmf[keep], pmf[~keep].sum()) * len(y)
        ok = exp_k > 0
        chi2 = ((obs_k[ok] - exp_k[ok]) ** 2 / exp_k[ok]).sum()
        assert chi2 < sst.chi2.ppf(1 - 1e-4, ok.sum() - 1), (lam, chi2)
    y = np.array([orc._poisson_one(5000.0, i, 1, 1) for i in range(8000)])
    assert abs(y.mean() - 5000.0) < 5 * np.sqrt(5000.0 / 8000) and abs(y.var() / 5000.0 - 1) < 0.08
    assert orc._poisson_one(float("nan"), 0, 0, 0) == -1 and orc._poisson_one(-1.0, 0, 0, 0) == -1
    assert orc._poisson_one(0.0, 0, 0, 0) == 0 and orc._poisson_one(float("inf"), 0, 0, 0) == 2 ** 31 - 1
    # batching: element i depends on (first_index + i) only
    f = np.linspace(-3, 4, 40)
    r, ya = orc.poisson_exp(f, 7, 2)
    _, yb = orc.poisson_exp(f[25:], 7, 2, first_index=25)
    assert np.array_equal(ya[25:], yb) and np.allclose(r, np.exp(f))


def test_poigp_prior_draw_moments():
    """oracle.poigp_sample: f has the aggregated kernel's covariance and E[y | f] = exp(f)."""
    rng = np.random.default_rng(3)
    grid = np.arange(0, 1, 1 / 6)
    pts = orc.agg_points(rng.random((6, 5, 1)), grid, d=1)
    k = orc.agg_kernel_from_points(pts, 0.8, 0.3, 1e-4)
    eps = rng.standard_normal((20000, 6))
    f, rate, y = orc.poigp_sample(pts, 0.8, 0.3, eps, 1e-4, seed=5)
    assert np.abs(np.cov(f.T) - k).max() < 0.04
    assert np.allclose(rate, np.exp(f)) and (y >= 0).all()
    assert abs((y - rate).mean()) < 5 * np.sqrt(rate.mean() / y.size)
