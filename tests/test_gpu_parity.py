"""GPU parity tests: every hot-path entry point, called through the C ABI (gp_vae_amd.ops ->
libgpvae_hip.so), against the CPU oracle on identical inputs.

Tolerances (BASELINE.json north_star): 1e-5 relative in float64 (we assert 1e-9 or better where
the arithmetic allows), 1e-3 relative in float32 with the scripts' default jitters.  For float32
Cholesky FACTORS the attainable forward error is cond(K)*eps, whichever algorithm is used
(SURVEY.md §6), so the assertion is  err <= max(1e-3, 3 * err of LAPACK's own float32 factor, spotrf)
plus a backward-error bound ||L L^T - K|| / ||K|| <= 2e-6.  The same yard-stick holds for every
float32 quantity that goes through a solve (kriging variances, batched kriging, MVN gradients):
err <= max(1e-3, 3 * error of the ORACLE run in float32 on the same inputs) — `f32_bar` below.
"""
import numpy as np
import pytest
import torch

from oracle import gpvae_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [(np.float64, torch.float64, 1e-9), (np.float32, torch.float32, 1e-3)]


def rel(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a, tdt):
    return torch.tensor(np.asarray(a), dtype=tdt, device=DEV)


def f32_bar(err_oracle_f32):
    """The float32 bar: 1e-3 (north_star), or 3x what the oracle's own arithmetic loses in float32 on
    the same inputs when conditioning puts even LAPACK-on-CPU above 1e-3."""
    return max(1e-3, 3.0 * float(err_oracle_f32))


def grid2(g):
    gg = np.arange(0, 1, 1 / g)
    u, v = np.meshgrid(gg, gg)
    return np.stack([u.ravel(), v.ravel()], 1)


@pytest.fixture(scope="module")
def ops():
    from gp_vae_amd import ops as o
    assert torch.cuda.is_available()
    return o


# ---- a1 ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_gram_golden_and_edges(ops, golden, ndt, tdt, tol):
    g = golden
    gt = 1e-12 if ndt == np.float64 else 1e-6
    cases = [("gram_x16", "gram_x16", 1.3, 0.15, 1e-4, "gram_k16"), ("gram_x44", "gram_x44", 0.9, 0.4, 1e-3, "gram_k44"),
             ("gram_xa", "gram_xb", 1.1, 0.3, 1e-2, "gram_rect"), ("gram_xc", "gram_xd", 1.0, 0.5, 1e-2, "gram_sqdiff")]
    for xa, xb, var, ls, jit, want in cases:
        k = ops.gram_se(dev(g[xa], tdt), dev(g[xb], tdt), var, ls, jit)
        assert rel(k, g[want]) < gt, want
    # ragged sizes (not multiples of the 4-wide store / 256-wide block), 1-D and 2-D
    rng = np.random.default_rng(0)
    for nx, nz, d in ((1, 1, 1), (3, 257, 1), (130, 67, 2), (33, 33, 2)):
        x, z = rng.random((nx, d)).astype(ndt), rng.random((nz, d)).astype(ndt)
        k = ops.gram_se(dev(x, tdt), dev(z, tdt), 0.7, 0.25, 1e-3)
        assert k.shape == (nx, nz) and rel(k, orc.exp_kernel2(x.astype(np.float64), z.astype(np.float64), 0.7, 0.25, 1e-3)) < gt
    # batched hyper-parameters
    x = np.arange(0, 1, 1 / 50).astype(ndt)
    var, ls = np.array([1.0, 0.5, 2.0]), np.array([0.05, 0.3, 1.0])
    kb = ops.gram_se(dev(x, tdt), dev(x, tdt), dev(var, tdt), dev(ls, tdt), 1e-4)
    for b in range(3):
        assert rel(kb[b], orc.exp_kernel2(x.astype(np.float64), x.astype(np.float64), var[b], ls[b], 1e-4)) < gt
    # empty input
    assert ops.gram_se(dev(np.zeros((0,)), tdt), dev(x, tdt), 1.0, 1.0).shape == (0, 50)


def test_gram_full_size_properties(ops):
    x = dev(grid2(64), torch.float32)  # C3: 64x64 grid
    k = ops.gram_se(x, x, 1.0, 0.3, 1e-3)
    assert k.shape == (4096, 4096)
    assert torch.equal(k, k.T) and torch.allclose(torch.diagonal(k), torch.full((4096,), 1.001, device=DEV))
    idx = np.random.default_rng(0).integers(0, 4096, 64)
    sub = orc.exp_kernel2(grid2(64)[idx], grid2(64)[idx], 1.0, 0.3, 0.0)
    got = k[idx][:, idx].cpu().numpy() - 1e-3 * (idx[:, None] == idx[None, :])
    assert np.abs(got - sub).max() < 1e-6


# ---- a2 ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_agg_kernel(ops, golden, ndt, tdt, tol):
    gt = 1e-12 if ndt == np.float64 else 1e-6
    assert rel(ops.gram_agg(dev(golden["agg_pts1"], tdt), 1.2, 0.2, 1e-3), golden["agg_k1"]) < gt
    assert rel(ops.gram_agg(dev(golden["agg_pts2"], tdt), 0.8, 0.3, 1e-3), golden["agg_k2"]) < gt
    rng = np.random.default_rng(3)
    pts = orc.agg_points(rng.random((37, 3, 1)), np.arange(0, 1, 1 / 37), d=1)  # ragged vs 16x16 tiles
    assert rel(ops.gram_agg(dev(pts, tdt), 1.0, 0.1, 0.0), orc.agg_kernel_from_points(pts, 1.0, 0.1, 0.0)) < gt


def test_agg_kernel_grid_surface(ops):
    from gp_vae_amd import gp
    grid = np.arange(0, 1, 1 / 6)
    k = gp.agg_kernel_grid(3, 2, 4, gp.exp_kernel2, 1.0, 0.3, grid)
    assert k.shape == (36, 36) and torch.allclose(k, k.T, atol=1e-6)
    with pytest.raises(NotImplementedError):
        gp.agg_kernel_grid(3, 1, 4, lambda *a: None, 1.0, 0.3, grid)


# ---- a3 ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
@pytest.mark.parametrize("n", [8, 64, 300, 625])
def test_potrf_golden(ops, golden, ndt, tdt, tol, n):
    var, ls, jit = golden["chol_n%d_hyp" % n]
    x = golden["chol_n%d_x" % n]
    K64 = orc.exp_kernel2(x, x, var, ls, jit)
    L64 = orc.cholesky_lower(K64)
    if n <= 64:
        assert rel(L64, golden["chol_n%d_L" % n]) < 1e-12
    A = ops.gram_se(dev(x, tdt), dev(x, tdt), var, ls, jit).clone()
    Kd = A.clone()
    info = ops.potrf_(A)
    assert info.item() == 0
    L = A.cpu().numpy().astype(np.float64)
    assert np.all(np.triu(L, 1) == 0)  # clean lower factor, as jnp.linalg.cholesky returns
    Kh = Kd.cpu().numpy().astype(np.float64)
    assert rel(L @ L.T, Kh) < (1e-14 if ndt == np.float64 else 2e-6)
    err = rel(L, L64)
    if ndt == np.float64:
        assert err < tol
    else:
        lap = rel(orc.cholesky_lower(Kh.astype(np.float32)), L64)  # LAPACK spotrf on the same float32 matrix
        assert err < max(1e-3, 3 * lap), (err, lap)


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_potrf_non_pd_batched_and_trailing_rows(ops, ndt, tdt, tol):
    rng = np.random.default_rng(1)
    # batch of 3, the middle one indefinite from column 70 on
    n = 150
    x = np.sort(rng.random(n))
    Ks = np.stack([orc.exp_kernel2(x, x, 1.0, ls, 1e-3) for ls in (0.05, 0.1, 0.2)])
    Ks[1, 69, 69] = -5.0
    A = dev(Ks, tdt)
    info = ops.potrf_(A)
    assert info.cpu().tolist() == [0, 70, 0]
    assert torch.isnan(A[1]).all()  # jnp.linalg.cholesky semantics: failed factor is all NaN
    for b in (0, 2):
        assert rel(A[b], orc.cholesky_lower(Ks[b])) < (1e-9 if ndt == np.float64 else 5e-3)
    A2 = dev(Ks, tdt)
    info2 = ops.potrf_(A2, zero_upper=False, nan_on_fail=False)  # LAPACK-style: info only
    assert info2.cpu().tolist() == [0, 70, 0] and not torch.isnan(A2[0]).any()
    # trailing rows ride along: X <- X L^-T, several workgroups' worth, ragged n and lda != n
    for n, m in ((10, 291), (100, 526), (257, 130)):
        x = np.sort(rng.random(n))
        K = orc.exp_kernel2(x, x, 1.0, 0.2, 1e-5) + 0.05 * np.eye(n)
        X = rng.standard_normal((m, n))
        ld = n + 5
        Abuf = torch.zeros((n, ld), dtype=tdt, device=DEV)
        Xbuf = torch.zeros((m, ld), dtype=tdt, device=DEV)
        Abuf[:, :n] = dev(K, tdt)
        Xbuf[:, :n] = dev(X, tdt)
        assert ops.potrf_(Abuf[:, :n], Xbuf[:, :n]).item() == 0
        L = orc.cholesky_lower(K)
        t = 1e-9 if ndt == np.float64 else 2e-3
        assert rel(Abuf[:, :n], L) < t and rel(Xbuf[:, :n], orc.right_solve_lt(L, X)) < t
        assert torch.all(Abuf[:, n:] == 0) and torch.all(Xbuf[:, n:] == 0)  # padding untouched


def test_potrf_more_workgroups_than_cus(ops):
    """Regression: panel launches with more workgroups than the chip holds at once (320 > 256) —
    late workgroups must still read the unfactored diagonal block."""
    rng = np.random.default_rng(7)
    n, batch = 1100, 64
    x = dev(np.sort(rng.random(n)), torch.float32)
    ls = torch.linspace(0.02, 0.2, batch, device=DEV)
    K = ops.gram_se(x, x, torch.ones(batch, device=DEV), ls, 1e-3)
    A = K.clone()
    info = ops.potrf_(A)
    assert int(info.abs().sum().item()) == 0
    L = A.double()
    err = torch.linalg.norm(L @ L.transpose(1, 2) - K.double(), dim=(1, 2)) / torch.linalg.norm(K.double(), dim=(1, 2))
    assert err.max().item() < 2e-6, err.max().item()


@pytest.mark.parametrize("batch", [1, 5, 8, 9, 13])
def test_potrf_tile_map_batches(ops, batch):
    """The trailing update is one 1-D launch whose workgroup ids are dealt to (XCD, matrix, tile):
    batch sizes on both sides of the 8-matrix switch and not multiples of 8, with extra rows (their
    tiles form the rectangle under the triangle), n = 3 outer blocks."""
    rng = np.random.default_rng(21)
    n, m = 700, 200
    K = np.empty((batch, n, n)); X = rng.standard_normal((batch, m, n))
    for b in range(batch):
        g = rng.standard_normal((n, n + 16))
        K[b] = g @ g.T / n + np.eye(n)
    A, Xd = dev(K, torch.float64), dev(X, torch.float64)
    info = ops.potrf_(A, Xd)
    assert info.cpu().tolist() == [0] * batch
    for b in range(batch):
        L = np.linalg.cholesky(K[b])
        assert rel(A[b], L) < 1e-12 and rel(Xd[b], orc.right_solve_lt(L, X[b])) < 1e-11


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_potrf_delayed_trailing_update_ragged_with_extra_rows(ops, ndt, tdt, tol):
    """From 15 outer blocks up the trailing update is delayed over pairs of outer blocks (potrf.hip mfma_group):
    inside a pair only the next 256 columns are brought up to date (the narrow tile map), after it everything
    right of it with K = 512.  Ragged n (a partial last outer block and 128-tile) with extra rows, whose tiles
    sit in the narrow map too, batch on both sides of the 8-matrix XCD switch."""
    rng = np.random.default_rng(77)
    n, m = 3900, 200
    x = rng.random((n, 2))
    for batch in (2, 9):
        ls = np.linspace(0.08, 0.5, batch)
        K = np.stack([orc.exp_kernel2(x, x, 1.0, l, 1e-3) for l in ls]).astype(ndt)
        X = rng.standard_normal((batch, m, n)).astype(ndt)
        A, Xd = dev(K, tdt), dev(X, tdt)
        info = ops.potrf_(A, Xd)
        assert info.cpu().tolist() == [0] * batch
        for b in (0, batch - 1):
            L64 = np.linalg.cholesky(K[b].astype(np.float64))
            Lg = A[b].cpu().numpy().astype(np.float64)
            Kb = K[b].astype(np.float64)
            bw = np.linalg.norm(Lg @ Lg.T - Kb) / np.linalg.norm(Kb)
            assert bw < (1e-14 if ndt == np.float64 else 2e-6), bw
            want = orc.right_solve_lt(L64, X[b].astype(np.float64))
            if ndt == np.float64:
                t = ts = 1e-9
            else:  # what LAPACK's float32 spotrf + the same solve lose on these inputs
                Lf = orc.cholesky_lower(K[b])
                t, ts = f32_bar(rel(Lf, L64)), f32_bar(rel(orc.right_solve_lt(Lf, X[b]), want))
            assert rel(Lg, L64) < t and rel(Xd[b], want) < ts


def test_potrf_f64_three_chunk_panel_launches(ops):
    """float64 panel launches take three 64-row chunks per workgroup where that saves a round of one-per-CU
    workgroups (potrf.hip PanelCfg<double>): batch 48 x n = 1024 (+ 70 extra rows) has such launches, the same
    matrices factored one at a time have none — same bits either way, and the factor against the oracle."""
    rng = np.random.default_rng(5)
    n, m, batch = 1024, 70, 48
    x = rng.random((n, 2))
    ls = np.linspace(0.05, 0.6, batch)
    K = np.stack([orc.exp_kernel2(x, x, 1.0, l, 1e-3) for l in ls])
    K[7, 600, 600] = -1.0  # not positive definite from column 600 on: info must say so in the batched call too
    X = rng.standard_normal((batch, m, n))
    A, Xd = dev(K, torch.float64), dev(X, torch.float64)
    info = ops.potrf_(A, Xd, zero_upper=True, nan_on_fail=False)
    want = [0] * batch
    want[7] = 601
    assert info.cpu().tolist() == want
    for b in (0, 7, 20, 47):
        A1, X1 = dev(K[b:b + 1], torch.float64), dev(X[b:b + 1], torch.float64)
        i1 = ops.potrf_(A1, X1, zero_upper=True, nan_on_fail=False)
        assert i1.item() == want[b]
        assert torch.equal(torch.nan_to_num(A1[0]), torch.nan_to_num(A[b]))
        assert torch.equal(torch.nan_to_num(X1[0]), torch.nan_to_num(Xd[b]))
    L64 = np.linalg.cholesky(K[20])
    assert rel(A[20], L64) < 1e-10 and rel(Xd[20], orc.right_solve_lt(L64, X[20])) < 1e-9


def test_potrf_overlapped_sub_batches(ops):
    """The sub-batch / side-stream form gives the factors and info values of the plain call."""
    rng = np.random.default_rng(31)
    n, batch = 200, 7
    M = rng.standard_normal((batch, n, 2 * n))
    A = M @ M.transpose(0, 2, 1) / (2 * n) + 0.1 * np.eye(n)
    A[5] = -A[5]  # not positive definite: info > 0 must come back in position 5
    for parts in (1, 2, 3, 9):
        ref, got = dev(A, torch.float32), dev(A, torch.float32)
        i0 = ops.potrf_(ref)
        i1 = ops.potrf_overlapped_(got, parts=parts)
        torch.cuda.synchronize()
        assert torch.equal(i0, i1) and i1[5].item() > 0 and (i1[:5] == 0).all()
        assert torch.equal(torch.nan_to_num(ref), torch.nan_to_num(got))


def test_potrf_near_rank_one(ops):
    """Long lengthscales (the heavy tail of the training prior InverseGamma(1, .1), vae.py:157) make
    K ~ var 11^T + jitter I.  The HIP path must factor these with a backward error within 3x that of the
    CORRECTLY ROUNDED factor (numpy.linalg.cholesky computes float32 input in float64 and rounds: a
    stricter yard-stick than LAPACK's float32 spotrf, which tests/test_gpu_bench_shapes.py compares with
    at n = 4096).  Regression: summing the panel products before subtracting them, and a 1-ulp biased
    1/sqrt(pivot), gave 30x this error and spurious info > 0 at C1."""
    n = 300
    x = (np.arange(n) / n).astype(np.float32)
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(24):
        ls, var = float(10 ** rng.uniform(1, 5)), float(rng.lognormal(0, 0.1))
        K = ops.gram_se(dev(x, torch.float32), dev(x, torch.float32), var, ls, 1e-4)
        Kc = K.cpu().numpy()
        A = K.clone()
        info = ops.potrf_(A)
        Ll = np.linalg.cholesky(Kc)  # float64 arithmetic, result rounded to float32: succeeds on every one of these
        assert int(info.item()) == 0, (ls, var)
        L = A.cpu().numpy().astype(np.float64)
        e_gpu = np.abs(L @ L.T - Kc).max()
        e_lap = np.abs(Ll.astype(np.float64) @ Ll.astype(np.float64).T - Kc).max()
        worst = max(worst, e_gpu / e_lap)
    assert worst < 3.0, worst


def test_potrf_full_size_backward_error(ops):
    x = dev(grid2(64), torch.float32)  # C3 size, script jitter (2d_examples.py:503)
    ls = torch.tensor([0.05, 0.3, 1.0], device=DEV)
    K = ops.gram_se(x, x, torch.ones(3, device=DEV), ls, 1e-3)
    A = K.clone()
    info = ops.potrf_(A)
    assert info.cpu().tolist() == [0, 0, 0]
    for b in range(3):
        L = A[b].double()
        assert (torch.linalg.norm(L @ L.T - K[b].double()) / torch.linalg.norm(K[b].double())).item() < 2e-6
    x1 = dev(np.arange(0, 1, 1 / 1024), torch.float64)  # C2 size in float64 against LAPACK
    K1 = ops.gram_se(x1, x1, 1.0, 0.05, 1e-4)
    A1 = K1.clone()
    ops.potrf_(A1)
    assert rel(A1, np.linalg.cholesky(K1.cpu().numpy())) < 1e-9


# ---- a3/a4 draws -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_draw(ops, golden, ndt, tdt, tol):
    gt = 1e-12 if ndt == np.float64 else 1e-5
    L = golden["chol_n64_L"]
    Ld = dev(L + np.triu(np.full((64, 64), 7.0), 1), tdt)  # garbage above the diagonal must be ignored
    F, Y = ops.draw(Ld, dev(golden["draw_E"], tdt), dev(golden["draw_E2"], tdt), 0.07, want_f=True)
    assert rel(F, golden["draw_F"]) < gt and rel(Y, golden["draw_Y"]) < gt
    rng = np.random.default_rng(2)
    for n, nb in ((1, 1), (130, 5), (300, 257), (625, 100)):
        Lr = np.tril(rng.standard_normal((n, n)))
        E, E2 = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
        Fo, Yo = orc.gp_draw(Lr, E, 0.3, E2)
        Yd = ops.draw(dev(Lr, tdt), dev(E, tdt), dev(E2, tdt), 0.3)
        assert rel(Yd, Yo) < gt
        assert rel(ops.draw(dev(Lr, tdt), dev(E, tdt)), Fo) < gt  # no noise term


def test_draw_full_size_covariance(ops):
    n, B = 4096, 4096
    x = dev(grid2(64), torch.float32)
    K = ops.gram_se(x, x, 1.0, 0.2, 1e-3)
    A = K.clone()
    ops.potrf_(A, zero_upper=False)
    E = ops.randn((B, n), seed=3, stream_id=1)
    Y = ops.draw(A, E)
    # linearity: draw(E1 + E2) == draw(E1) + draw(E2)
    E2 = ops.randn((B, n), seed=3, stream_id=2)
    assert rel(ops.draw(A, E + E2), (Y + ops.draw(A, E2)).cpu().numpy()) < 1e-5
    idx = torch.tensor(np.random.default_rng(0).integers(0, n, 32), device=DEV)
    S = (Y[:, idx].double().T @ Y[:, idx].double() / B).cpu().numpy()
    assert np.abs(S - K[idx][:, idx].cpu().numpy()).max() < 0.12  # ~ 5 sigma of a 4096-sample covariance


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_randn(ops, golden, ndt, tdt, tol):
    want = golden["randn_f64" if ndt == np.float64 else "randn_f32"]
    got = ops.randn((37,), tdt, DEV, seed=123456789012345, stream_id=9).cpu().numpy()
    # f32 normals come from the hardware log2 / sqrt / sin / cos instructions: |error| <= 3e-5 at |z| ~ 5
    assert np.abs(got - want).max() < (1e-13 if ndt == np.float64 else 5e-5)
    if ndt == np.float32:
        big = ops.randn((1 << 20,), tdt, DEV, seed=7, stream_id=3).cpu().numpy().astype(np.float64)
        assert np.abs(big - orc.randn(1 << 20, seed=7, stream_id=3, dtype=np.float64)).max() < 5e-5
    z = ops.randn((1 << 20,), tdt, DEV, seed=1, stream_id=0)
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    assert not torch.equal(z[:1000], ops.randn((1000,), tdt, DEV, seed=1, stream_id=1))


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_trsm_against_existing_factor(ops, ndt, tdt, tol):
    """gpvae_trsm: X L^-T and X L^-1 against scipy.linalg.solve_triangular, ragged n (not a multiple of
    the 64-column block), padded leading dimensions, more rows than one workgroup; both in turn = X A^-1."""
    import scipy.linalg
    rng = np.random.default_rng(31)
    for n, m in ((1, 3), (50, 7), (64, 300), (333, 515), (1000, 260)):
        g = rng.standard_normal((n, n + 8))
        A = g @ g.T / n + np.eye(n)
        L = np.linalg.cholesky(A)
        X = rng.standard_normal((m, n))
        Lb = torch.zeros((n, n + 3), dtype=tdt, device=DEV); Lb[:, :n] = dev(L, tdt)
        Lb[:, :n] += torch.triu(torch.full((n, n), 7.0, dtype=tdt, device=DEV), 1)  # junk above the diagonal
        t = 1e-10 if ndt == np.float64 else 2e-4
        Xb = torch.zeros((m, n + 5), dtype=tdt, device=DEV); Xb[:, :n] = dev(X, tdt)
        ops.trsm_(Lb[:, :n], Xb[:, :n], trans=True)
        want = scipy.linalg.solve_triangular(L, X.T, lower=True).T
        assert rel(Xb[:, :n], want) < t and torch.all(Xb[:, n:] == 0)
        ops.trsm_(Lb[:, :n], Xb[:, :n], trans=False)
        assert rel(Xb[:, :n], np.linalg.solve(A, X.T).T) < 20 * t
        Xc = dev(X, tdt)
        ops.trsm_(Lb[:, :n], Xc, trans=False)
        assert rel(Xc, scipy.linalg.solve_triangular(L, X.T, lower=True, trans="T").T) < t


# ---- a9 kriging ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_kriging(ops, golden, ndt, tdt, tol):
    g = golden
    x, y, obs, xi = g["krig_x"], g["krig_y"], g["krig_obs"], g["krig_xi"]
    var, ls, sig = g["krig_hyp"]
    mask = np.ones(300, bool)
    mask[obs] = False
    m, s, v, info = ops.krig_predict(dev(x[obs], tdt), dev(x[mask], tdt), dev(y[obs], tdt), var, ls, sig, 1e-5, dev(xi, tdt))
    t = 1e-8 if ndt == np.float64 else 1e-3
    assert info.item() == 0
    assert rel(m, g["krig_mean"]) < t and rel(s, g["krig_sample"]) < t and rel(v, g["krig_var"]) < t
    # the n_test == n_obs quirk (jitter on k_px)
    x2, y2, o2 = g["krigsq_x"], g["krigsq_y"], g["krigsq_obs"]
    mk = np.ones(20, bool)
    mk[o2] = False
    m2, _, v2, _ = ops.krig_predict(dev(x2[o2], tdt), dev(x2[mk], tdt), dev(y2[o2], tdt), 1.0, 0.2, 0.05)
    if ndt == np.float64:
        assert rel(m2, g["krigsq_mean"]) < t and rel(v2, g["krigsq_var"]) < 1e-7
    else:  # the variance is a difference of two O(var) numbers: judged against the oracle in float32
        f = np.float32
        m32, _, v32 = orc.gp_prediction(x2.astype(f), y2.astype(f), o2, f(1.0), f(0.2), f(0.05))
        assert rel(m2, g["krigsq_mean"]) < f32_bar(rel(m32, g["krigsq_mean"]))
        assert rel(v2, g["krigsq_var"]) < f32_bar(rel(v32, g["krigsq_var"])), (rel(v2, g["krigsq_var"]), rel(v32, g["krigsq_var"]))
    # 2-D, several panels, against the literal inverse-based oracle
    rng = np.random.default_rng(4)
    xg = grid2(30)
    o = np.sort(rng.permutation(900)[:400])
    yy = np.sin(5 * xg[:, 0]) * np.cos(3 * xg[:, 1]) + 0.05 * rng.standard_normal(900)
    mo, so, vo = orc.gp_prediction(xg, yy, o, 1.1, 0.25, 0.05, None)
    mk = np.ones(900, bool)
    mk[o] = False
    md, _, vd, _ = ops.krig_predict(dev(xg[o], tdt), dev(xg[mk], tdt), dev(yy[o], tdt), 1.1, 0.25, 0.05)
    assert rel(md, mo) < (1e-7 if ndt == np.float64 else 2e-3)
    if ndt == np.float64:
        assert np.abs(vd.cpu().numpy() - vo).max() < 1e-8


def test_kriging_mid_size_f64(ops):
    """C4's arithmetic at a size the oracle's literal inverse-based form finishes in seconds
    (48x48 grid, 2000 observations, f64): every blocked path of the factorisation with trailing rows
    (outer blocks, panel chunks, extra-row tiles of the trailing update) against inference.py:442-460."""
    g = np.arange(48) / 48.0
    u, v = np.meshgrid(g, g)
    x = np.stack([u.ravel(), v.ravel()], 1)
    rng = np.random.default_rng(3)
    obs = np.sort(rng.permutation(x.shape[0])[:2000])
    y = np.full(x.shape[0], np.nan)
    y[obs] = np.sin(5 * x[obs, 0]) * np.cos(3 * x[obs, 1]) + 0.05 * rng.standard_normal(2000)
    mask = np.ones(x.shape[0], bool); mask[obs] = False
    xi = rng.standard_normal(int(mask.sum()))
    m_ref, s_ref, v_ref = orc.gp_prediction(x, y, obs, 1.3, 0.15, 0.05, xi)
    xt = torch.tensor(x, dtype=torch.float64, device=DEV)
    mean, samp, var_out, info = ops.krig_predict(xt[torch.tensor(obs, device=DEV)], xt[torch.tensor(np.nonzero(mask)[0], device=DEV)],
                                                 torch.tensor(y[obs], dtype=torch.float64, device=DEV), 1.3, 0.15, 0.05, 1e-5,
                                                 torch.tensor(xi, dtype=torch.float64, device=DEV))
    assert int(info.item()) == 0
    assert rel(mean, m_ref) < 1e-9 and rel(samp, s_ref) < 1e-9   # north_star: 1e-5 rel in fp64
    assert np.abs(var_out.cpu().numpy() - np.maximum(v_ref, 0)).max() < 1e-9


def test_kriging_c4_full_size_properties(ops):
    """C4 at full size (192x192 grid, 32768 observations, 4096 test points, f64): an 8 GiB factor in place,
    buffer extents beyond 4 GiB and row offsets beyond 2^31 bytes (TileIO::usable / LoaderKC::fast_end
    guards).  The oracle's LU inverse of this size is out of reach, so: info == 0; backward error of the
    factor on sampled rows (L L^T)[rows] vs K[rows]; predictive variance >= 0 and <= prior variance;
    the mean recovers the noise-free surface to the noise level; the sample uses the given normals."""
    g = np.arange(192) / 192.0
    u, v = np.meshgrid(g, g)
    x = np.stack([u.ravel(), v.ravel()], 1)
    rng = np.random.default_rng(0)
    obs = np.sort(rng.permutation(x.shape[0])[:32768])
    mask = np.ones(x.shape[0], bool); mask[obs] = False
    surf = lambda p: np.sin(3 * np.pi * p[:, 0]) + 0.3 * np.cos(9 * np.pi * p[:, 1]) + 0.5 * np.sin(7 * np.pi * (p[:, 0] + p[:, 1]))
    y = surf(x) + 0.05 * rng.standard_normal(x.shape[0])
    xt = torch.tensor(x, dtype=torch.float64, device=DEV)
    x_obs = xt[torch.tensor(obs, device=DEV)].contiguous()
    x_test = xt[torch.tensor(np.nonzero(mask)[0], device=DEV)].contiguous()
    n_obs = x_obs.shape[0]
    # the factor alone: A = k_xx + (jitter + sigma) I, in place
    A = ops.gram_se(x_obs, x_obs, 1.0, 0.2, 1e-5)
    A.diagonal().add_(0.05)
    assert int(ops.potrf_(A).item()) == 0  # upper triangle cleared: A is L
    rows = torch.tensor(np.sort(rng.integers(0, n_obs, 24)), device=DEV)
    Krows = ops.gram_se(x_obs[rows].contiguous(), x_obs, 1.0, 0.2, 0.0)
    Krows[torch.arange(24, device=DEV), rows] += 1e-5 + 0.05
    LLt = A[rows] @ A.T
    assert (torch.linalg.norm(LLt - Krows) / torch.linalg.norm(Krows)).item() < 1e-13
    del A, LLt
    xi = ops.randn((x_test.shape[0],), torch.float64, DEV, seed=3)
    mean, samp, var_out, info = ops.krig_predict(x_obs, x_test, torch.tensor(y[obs], dtype=torch.float64, device=DEV),
                                                 1.0, 0.2, 0.05, 1e-5, xi)
    assert int(info.item()) == 0
    vo = var_out.cpu().numpy()
    assert vo.min() >= 0.0 and vo.max() <= 1.0 + 1e-5 + 1e-12
    # (ls = 0.2 smooths the 7-9 pi features: 0.351, the value round 1 measured; the zero predictor gives 0.93)
    assert np.sqrt(np.mean((mean.cpu().numpy() - surf(x[mask])) ** 2)) < 0.4
    assert rel(samp, (mean + torch.sqrt(var_out) * xi).cpu().numpy()) < 1e-14
    # a closed-form answer at full size: for y = sum_j c_j A[:, j] (columns of A = k_xx + (jitter + sigma) I)
    # A^-1 y = sum_j c_j e_j, so the kriging mean must be sum_j c_j k(x_test, x_obs[j]) exactly
    cols = [123, 17000, 32767]
    coef = [1.0, -2.5, 0.7]
    yc = torch.zeros(n_obs, dtype=torch.float64, device=DEV)
    want = torch.zeros(x_test.shape[0], dtype=torch.float64, device=DEV)
    for j, c in zip(cols, coef):
        col = ops.gram_se(x_obs, x_obs[j:j + 1].contiguous(), 1.0, 0.2, 0.0).reshape(-1)
        col[j] += 1e-5 + 0.05
        yc += c * col
        want += c * ops.gram_se(x_test, x_obs[j:j + 1].contiguous(), 1.0, 0.2, 0.0).reshape(-1)
    mean_c, _, _, info_c = ops.krig_predict(x_obs, x_test, yc, 1.0, 0.2, 0.05, 1e-5)
    assert int(info_c.item()) == 0 and rel(mean_c, want.cpu().numpy()) < 1e-8, rel(mean_c, want.cpu().numpy())


def test_gpinference_surface(ops, golden):
    from gp_vae_amd.gp import exp_kernel2
    from gp_vae_amd.inference import GPInference
    g = golden
    inf = GPInference(exp_kernel2, 1, g["krig_x"], g["krig_y"], g["krig_obs"], {}, dtype=torch.float64)
    var, ls, sig = g["krig_hyp"]
    mean, pred = inf.gp_prediction(7, inf.x, inf.y, var, ls, sig)
    assert rel(mean, g["krig_mean"]) < 1e-8 and mean.shape == pred.shape == (290,)
    post = {"kernel_var": [1.0, 1.1], "kernel_length": [0.07, 0.1], "kernel_sigma": [0.03, 0.05]}
    means, preds = inf.gp_fit_from_samples(1, post)
    assert means.shape == (2, 290)
    assert rel(means[1], orc.gp_prediction(g["krig_x"], g["krig_y"], g["krig_obs"], 1.1, 0.1, 0.05)[0]) < 1e-8
    # the batched sweep gives every draw what gp_prediction gives it alone (same normals too)
    m1, p1 = inf.gp_prediction(1 + 1, inf.x, inf.y, 1.1, 0.1, 0.05)
    assert rel(means[1], m1.cpu().numpy()) < 1e-12 and rel(preds[1], p1.cpu().numpy()) < 1e-12


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_kriging_batched(ops, ndt, tdt, tol):
    """f2: S hyper-parameter draws in one call against the oracle's literal arithmetic per draw,
    1-D and 2-D points, a non-PD problem in the middle of the batch (info > 0, NaN outputs)."""
    rng = np.random.default_rng(21)
    for d, n, n_obs in ((1, 150, 37), (2, 144, 100)):
        x = np.arange(n) / n if d == 1 else np.stack(np.meshgrid(np.arange(12) / 12, np.arange(12) / 12), -1).reshape(-1, 2)
        obs = np.sort(rng.permutation(n)[:n_obs])
        y = np.full(n, np.nan)
        y[obs] = np.sin(5 * (x[obs] if d == 1 else x[obs].sum(1))) + 0.1 * rng.standard_normal(n_obs)
        mask = np.ones(n, bool); mask[obs] = False
        S = 5
        var = rng.uniform(0.5, 1.5, S); ls = rng.uniform(0.05, 0.3, S); sig = rng.uniform(0.01, 0.1, S)
        sig[2] = -10.0  # A = K + sigma I not positive definite
        xi = rng.standard_normal((S, n - n_obs))
        m, p, v, info = ops.krig_predict_batched(dev(x[obs].astype(ndt), tdt), dev(x[mask].astype(ndt), tdt), dev(y[obs], tdt),
                                                 var, ls, sig, 1e-5, dev(xi, tdt))
        info = info.cpu().numpy()
        assert info[2] > 0 and (np.delete(info, 2) == 0).all()
        assert np.isnan(m[2].cpu().numpy()).all()
        for s in (0, 1, 3, 4):
            mo, po, vo = orc.gp_prediction(x, y, obs, var[s], ls[s], sig[s], xi[s])  # float64 truth
            if ndt == np.float64:
                assert rel(m[s], mo) < 1e-7 and rel(p[s], po) < 1e-7, (d, s)
            else:
                f = np.float32
                m32, p32, _ = orc.gp_prediction(x.astype(f), y.astype(f), obs, f(var[s]), f(ls[s]), f(sig[s]), xi[s].astype(f))
                assert rel(m[s], mo) < f32_bar(rel(m32, mo)), (d, s, rel(m[s], mo), rel(m32, mo))
                assert rel(p[s], po) < f32_bar(rel(p32, po)), (d, s, rel(p[s], po), rel(p32, po))


# ---- a10 / f1 ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_mvn_logp_and_gradient(ops, ndt, tdt, tol):
    """Batch of chains against the oracle's closed form (itself pinned to SciPy and finite differences
    in tests/test_oracle.py); 1-D and 2-D points, sizes that are not tile multiples, one chain whose
    K is not positive definite."""
    rng = np.random.default_rng(8)
    for x in (np.arange(150) / 150.0, rng.uniform(0, 1, (300, 2)), np.arange(37) / 37.0):
        n = x.shape[0]
        S = 4
        var = rng.uniform(0.7, 1.4, S); ls = rng.uniform(0.1, 0.3, S)
        jit = 1e-5 if ndt == np.float64 else 1e-3   # fp32: keep cond(K) within reach of the type
        F = np.stack([np.linalg.cholesky(orc.exp_kernel2(x, x, var[s], ls[s], jit)) @ rng.standard_normal(n) for s in range(S)])
        var_bad = var.copy(); var_bad[2] = -1.0
        lp, gf, gv, gl, info = ops.mvn_logp(dev(x.astype(ndt), tdt), dev(F, tdt), var_bad, ls, jit)
        info = info.cpu().numpy()
        assert info[2] > 0 and np.isnan(lp[2].item()) and np.isnan(gv[2].item())
        # fp32: cond(K) ~ 1e5 at these sizes; d/dvar and d/dls are differences of two O(n) terms
        # (alpha^T dK alpha and tr(K^-1 dK)).  The bar is 1e-3 unless the oracle's own arithmetic run
        # in float32 on these inputs loses more (worst chain of the batch on both sides).
        e_gpu, e_o32 = np.zeros(3), np.zeros(3)
        for s in (0, 1, 3):
            lo, gfo, gvo, glo = orc.mvn_logp_grads(x, F[s], var[s], ls[s], jit)
            assert abs(lp[s].item() - lo) < (1e-7 if ndt == np.float64 else 1e-3) * abs(lo), (n, s)
            errs = (rel(gf[s], gfo), abs(gv[s].item() - gvo) / max(1.0, abs(gvo)), abs(gl[s].item() - glo) / max(1.0, abs(glo)))
            if ndt == np.float64:
                assert max(errs) < 1e-7, (n, s, errs)
                continue
            f = np.float32
            _, gf3, gv3, gl3 = orc.mvn_logp_grads(x.astype(f), F[s].astype(f), f(var[s]), f(ls[s]), f(jit))
            o32 = (rel(gf3, gfo), abs(gv3 - gvo) / max(1.0, abs(gvo)), abs(gl3 - glo) / max(1.0, abs(glo)))
            e_gpu, e_o32 = np.maximum(e_gpu, errs), np.maximum(e_o32, o32)
        if ndt == np.float32:
            for q in range(3):
                assert e_gpu[q] < f32_bar(e_o32[q]), (n, q, e_gpu, e_o32)


def test_gp_regression_potential_gradient(ops, golden):
    """GPInference.potential_and_grad (what NUTS differentiates in gp_regression) against the oracle's
    potential and central finite differences of it, fp64."""
    from gp_vae_amd.gp import exp_kernel2
    from gp_vae_amd.inference import GPInference
    rng = np.random.default_rng(12)
    n = 60
    x = np.arange(n) / n
    obs = np.sort(rng.permutation(n)[:25])
    y = np.full(n, np.nan); y[obs] = np.sin(6 * x[obs]) + 0.05 * rng.standard_normal(25)
    inf = GPInference(exp_kernel2, 1, x, y, obs, {}, dtype=torch.float64)
    C = 3
    uv, ul, us = rng.normal(0, 0.1, C), np.log(rng.uniform(0.15, 0.4, C)), np.log(rng.uniform(0.05, 0.2, C))
    # latent values drawn from the prior they are scored under (a random f would make f^T K^-1 f ~ 1e6
    # and the finite differences below meaningless)
    f = np.stack([np.linalg.cholesky(orc.exp_kernel2(x[obs], x[obs], np.exp(uv[c]), np.exp(ul[c]), 1e-5)) @ rng.standard_normal(25) for c in range(C)])
    U, dF, dV, dL, dS = inf.potential_and_grad(f, uv, ul, us)
    h = 1e-5
    for c in range(C):
        P = lambda ff, a, b, g: orc.gp_regression_potential(x[obs], y[obs], ff, a, b, g)
        u0 = P(f[c], uv[c], ul[c], us[c])
        assert abs(U[c].item() - u0) < 1e-9 * abs(u0)
        fd = lambda fn: (fn(h) - fn(-h)) / (2 * h)
        assert abs(dV[c].item() - fd(lambda e: P(f[c], uv[c] + e, ul[c], us[c]))) < 1e-4 * max(1, abs(dV[c].item()))
        assert abs(dL[c].item() - fd(lambda e: P(f[c], uv[c], ul[c] + e, us[c]))) < 1e-4 * max(1, abs(dL[c].item()))
        assert abs(dS[c].item() - fd(lambda e: P(f[c], uv[c], ul[c], us[c] + e))) < 1e-4 * max(1, abs(dS[c].item()))
        e7 = np.zeros(25); e7[7] = 1.0
        assert abs(dF[c, 7].item() - fd(lambda e: P(f[c] + e * e7, uv[c], ul[c], us[c]))) < 1e-4 * max(1, abs(dF[c, 7].item()))


# ---- a5/a6/a7/a11 --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_vae_elbo_adam_decoder_golden(ops, golden, ndt, tdt, tol):
    g = golden
    n, h0, h1, z, B = (int(v) for v in g["elbo_dims"])
    t = 1e-11 if ndt == np.float64 else 2e-5
    flat = dev(g["elbo_params"], tdt)
    grads = torch.empty_like(flat)
    loss = ops.vae_elbo(flat, dev(g["elbo_Y"], tdt), dev(g["elbo_eps"], tdt), (h0, h1, z), 0.1, grads=grads)
    assert abs(loss.item() - g["elbo_loss"]) < t * abs(g["elbo_loss"])
    go, gd = orc.unflatten(g["elbo_grads"], n, h0, h1, z), orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
    for k in orc.PARAM_NAMES:
        assert rel(gd[k], go[k]) < t, k
    loss2 = ops.vae_elbo(flat, dev(g["elbo_Y"], tdt), dev(g["elbo_eps"], tdt), (h0, h1, z), 0.1)  # svi.evaluate
    assert abs(loss2.item() - g["elbo_loss"]) < t * abs(g["elbo_loss"])
    p, m, v = flat.clone(), dev(g["adam_m0"], tdt), dev(g["adam_v0"], tdt)
    ops.adam_(p, dev(g["elbo_grads"], tdt), m, v, 4, 1e-3)
    ta = 1e-13 if ndt == np.float64 else 5e-5  # (1 - 0.999f) carries a 1.3e-5 rounding
    assert rel(p, g["adam_p1"]) < ta and rel(m, g["adam_m1"]) < ta and rel(v, g["adam_v1"]) < ta
    pp = orc.unflatten(g["elbo_params"], n, h0, h1, z)
    dec = dev(np.concatenate([pp[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]), tdt)
    assert rel(ops.decoder_fwd(dec, dev(g["dec_Z"], tdt), n, h0, h1), g["dec_F"]) < t
    assert rel(ops.decoder_vjp(dec, dev(g["dec_Z"], tdt), dev(g["dec_dF"], tdt), h0, h1), g["dec_dZ"]) < t


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
@pytest.mark.parametrize("dims", [(100, 300, 35, 30, 10), (37, 625, 50, 30, 15), (300, 1024, 64, 32, 16), (5, 19, 3, 2, 1)])
def test_vae_elbo_reference_dims(ops, ndt, tdt, tol, dims):
    B, n, h0, h1, z = dims  # reference 1-D / 2-D layer sizes (toy_examples.py:615-630, 2d_examples.py)
    rng = np.random.default_rng(5)
    p = {k: (10 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    lo, go = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    flat = dev(orc.flatten(p), tdt)
    grads = torch.empty_like(flat)
    loss = ops.vae_elbo(flat, dev(Y, tdt), dev(ez, tdt), (h0, h1, z), 0.1, grads=grads)
    t = 1e-10 if ndt == np.float64 else 1e-4
    assert abs(loss.item() - lo) < t * abs(lo)
    gd = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
    for k in orc.PARAM_NAMES:
        assert rel(gd[k], go[k]) < t, k


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
@pytest.mark.parametrize("dims", [(64, 256, 512, 256, 64), (300, 100, 1024, 700, 33), (9, 40, 8, 1024, 1024), (13, 50, 600, 2, 2)])
def test_vae_wide_hidden_layers(ops, ndt, tdt, tol, dims):
    """SURVEY §8d's scaled MLP (512, 256, 64) and the API's widest layers (1024): every layer runs as a
    tile product (EPI_FUSED) when the mid weights do not fit LDS; (600, 2, 2) keeps the per-sample-block
    kernels but with a smaller block (16 -> 8/4/2 samples) so the activation buffers stay inside 64 KB.
    ELBO + gradients, decoder forward and its VJP against the oracle."""
    B, n, h0, h1, z = dims
    rng = np.random.default_rng(15)
    p = {k: (3 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    lo, go = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    flat = dev(orc.flatten(p), tdt)
    grads = torch.empty_like(flat)
    loss = ops.vae_elbo(flat, dev(Y, tdt), dev(ez, tdt), (h0, h1, z), 0.1, grads=grads)
    t = 1e-10 if ndt == np.float64 else 1e-4
    assert abs(loss.item() - lo) < t * abs(lo)
    gd = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
    for k in orc.PARAM_NAMES:
        assert rel(gd[k], go[k]) < t, k
    decp = dev(np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]), tdt)
    dF = rng.standard_normal((B, n))
    assert rel(ops.decoder_fwd(decp, dev(ez, tdt), n, h0, h1), orc.decoder(p, ez)) < t
    assert rel(ops.decoder_vjp(decp, dev(ez, tdt), dev(dF, tdt), h0, h1), orc.decoder_vjp(p, ez, dF)) < t


@pytest.mark.parametrize("dims", [(256, 192, 63, 5, 3), (128, 64, 1, 2, 1), (384, 1024, 35, 30, 10), (128, 640, 50, 30, 15),
                                  (4096, 1088, 50, 30, 15)])  # last: 2 column tiles per workgroup + a remainder
def test_vae_fused_output_layer(ops, dims):
    """f32 with gradients, B % 128 == 0, n % 64 == 0, h0 + 1 <= 64: the decoder's output layer runs as
    ONE kernel (forward product, data term, dD2 and [dV3; dc3]; vae_out_fused_kernel).  Widest and
    narrowest h0, one and several column tiles per workgroup; loss + 14 gradient arrays vs the oracle,
    and the evaluate-only call (separate kernels) must give the same loss."""
    B, n, h0, h1, z = dims
    rng = np.random.default_rng(17)
    p = {k: (5 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    lo, go = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    flat = dev(orc.flatten(p), torch.float32)
    grads = torch.empty_like(flat)
    loss = ops.vae_elbo(flat, dev(Y, torch.float32), dev(ez, torch.float32), (h0, h1, z), 0.1, grads=grads)
    loss_eval = ops.vae_elbo(flat, dev(Y, torch.float32), dev(ez, torch.float32), (h0, h1, z), 0.1)
    assert abs(loss.item() - lo) < 1e-4 * abs(lo) and abs(loss.item() - loss_eval.item()) < 1e-5 * abs(lo)
    gd = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
    for k in orc.PARAM_NAMES:
        assert rel(gd[k], go[k]) < 1e-4, k


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
@pytest.mark.parametrize("dims", [(256, 256, 50, 30, 15), (100, 300, 35, 30, 10), (128, 256, 512, 256, 64), (512, 1088, 50, 30, 15)])
def test_vae_elbo_is_bitwise_reproducible(ops, tdt, dims):
    """gpvae_vae_elbo sums across workgroups in a fixed order (slabs of partial sums, no floating-point
    atomics): two evaluations of the same inputs give the same bits — loss and all 14 gradient arrays — on
    the fused, unfused and wide-layer paths, and whatever the workspace held before (nothing in it needs
    clearing: it is handed over full of NaN)."""
    B, n, h0, h1, z = dims
    rng = np.random.default_rng(21)
    p = {k: (5 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    flat = dev(orc.flatten(p), tdt)
    Y, ez = dev(rng.standard_normal((B, n)), tdt), dev(rng.standard_normal((B, z)), tdt)
    out = []
    for fill in (float("nan"), 1.0e30):
        work = ops.VaeWorkspace(B, n, h0, h1, z, tdt, DEV)
        work.buf.view(torch.float32)[:] = fill
        grads = torch.full_like(flat, fill)
        loss = ops.vae_elbo(flat, Y, ez, (h0, h1, z), 0.1, grads=grads, work=work)
        ev = ops.vae_elbo(flat, Y, ez, (h0, h1, z), 0.1, work=work)  # evaluate-only on the used workspace
        out.append((loss.clone(), grads, ev.clone()))
    assert torch.isfinite(out[0][1]).all() and torch.isfinite(out[0][0]).all()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    lo, _ = orc.elbo_loss_and_grads(p, Y.double().cpu().numpy(), ez.double().cpu().numpy(), 0.1)
    assert abs(out[0][0].item() - lo) < (1e-10 if tdt == torch.float64 else 1e-4) * abs(lo)


def test_vae_elbo_full_size_c3(ops):
    """The bench shape (B = n = 4096, 50-30-15, fp32) against the oracle: every fast path of the tile
    products (VALU-free k loops, ragged-K through the buffer range check, TileIO epilogues, split-K
    atomics) is on at this size.  fp32 tolerance 1e-3 (north_star)."""
    B, n, h0, h1, z = 4096, 4096, 50, 30, 15
    rng = np.random.default_rng(11)
    p = {k: (10 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    lo, go = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    flat = dev(orc.flatten(p), torch.float32)
    grads = torch.empty_like(flat)
    loss = ops.vae_elbo(flat, dev(Y, torch.float32), dev(ez, torch.float32), (h0, h1, z), 0.1, grads=grads)
    assert abs(loss.item() - lo) < 1e-3 * abs(lo)
    gd = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
    for k in orc.PARAM_NAMES:
        assert rel(gd[k], go[k]) < 1e-3, k


def test_vaeinference_potential_gradient(ops):
    from gp_vae_amd.inference import VAEInference
    rng = np.random.default_rng(6)
    n, h0, h1, z = 300, 35, 30, 10  # shapes of the reference's pickled decoders (synthetic weights:
    # the fixtures are pickles, which this build does not unpickle)
    p = {k: 15 * v for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
    lst = [(p["V1"], p["c1"]), (), (p["V2"], p["c2"]), (), (p["V3"], p["c3"])]
    obs = np.array([20, 23, 100, 110, 117, 130, 133, 140, 170, 190])  # toy_examples.py:199
    y = np.sin(np.arange(300) / 30.0)
    inf = VAEInference(None, lst, z, 1, np.arange(0, 1, 1 / 300), y, obs, {}, dtype=torch.float64)
    Z, u = rng.standard_normal((64, z)), np.log(0.1) + 0.3 * rng.standard_normal(64)
    U, dz, du = inf.potential_and_grad(Z, u)
    assert rel(U, orc.vae_potential(p, Z, u, y, obs)) < 1e-10
    h = 1e-6
    for i in range(z):
        zp, zm = Z.copy(), Z.copy()
        zp[:, i] += h
        zm[:, i] -= h
        fd = (orc.vae_potential(p, zp, u, y, obs) - orc.vae_potential(p, zm, u, y, obs)) / (2 * h)
        assert np.allclose(dz[:, i].cpu().numpy(), fd, rtol=1e-5, atol=1e-5)
    fd = (orc.vae_potential(p, Z, u + h, y, obs) - orc.vae_potential(p, Z, u - h, y, obs)) / (2 * h)
    assert np.allclose(du.cpu().numpy(), fd, rtol=1e-5, atol=1e-5)
    assert rel(inf.predict(Z), orc.decoder(p, Z)) < 1e-12


def test_inference_fit_surface_with_nuts(ops):
    """The reference's end-to-end inference calls (VAEInference.fit, GPInference.gp_fit) run with the
    batched NUTS of gp_vae_amd/nuts.py in NumPyro's place: shapes of what they return, no divergence, and the posterior
    concentrates on the data (posterior-mean error at the observed points well below the prior's)."""
    from gp_vae_amd.gp import exp_kernel2
    from gp_vae_amd.inference import GPInference, VAEInference
    rng = np.random.default_rng(2)
    n = 80
    x = np.arange(n) / n
    obs = np.sort(rng.permutation(n)[:20])
    truth = np.sin(6 * x)
    y = np.full(n, np.nan); y[obs] = truth[obs] + 0.05 * rng.standard_normal(20)
    args = {"num_warmup": 150, "num_samples": 100, "num_chains": 4, "thinning": 2, "max_tree_depth": 6}
    inf = GPInference(exp_kernel2, 1, x, y, obs, args, seed=5, dtype=torch.float64)
    means, preds = inf.gp_fit(plot=False)
    assert inf.mcmc_info["leapfrogs_per_iteration"].mean().item() >= 1.0  # the NUTS driver ran
    # proposals that leave the positive-definite region come back NaN and are rejected: a few per cent
    assert means.shape == preds.shape == (200, n - 20) and inf.mcmc_info["divergent"] < 50
    mask = np.ones(n, bool); mask[obs] = False
    err = np.sqrt(np.mean((means.mean(0).cpu().numpy() - truth[mask]) ** 2))
    # 0.22-0.27 over repeated runs (the split-K atomics of gpvae_mvn_logp make the chains differ run to run);
    # the prior mean's error at these points is ~0.7
    assert err < 0.4, err
    # VAE-prior regression with a random (untrained) decoder: the flow runs and keeps its shapes
    nz, h1, h0 = 6, 9, 12
    p = orc.init_params(rng, n, h0, h1, nz, np.float64)
    dec = [(50 * p["V1"], p["c1"]), (), (50 * p["V2"], p["c2"]), (), (50 * p["V3"], p["c3"])]
    vinf = VAEInference(None, dec, nz, 1, x, y, obs, args, seed=6, dtype=torch.float64)
    prior, post = vinf.fit(plot=False, num_prior=64)
    assert prior.shape == (64, n) and post.shape == (200, n)
    assert torch.isfinite(post).all() and vinf.mcmc_info["divergent"] < 50  # warm-up overshoots, rejected
    yo = torch.tensor(y[obs], device=post.device)
    e_post = (post[:, obs].mean(0) - yo).abs().mean().item()
    e_prior = (prior[:, obs].mean(0) - yo).abs().mean().item()
    assert e_post < e_prior


@pytest.mark.parametrize("ndt,tdt,tol", DTYPES)
def test_kriging_batched_small_problems_in_lds(ops, ndt, tdt, tol):
    """f2 as the reference uses it (vmap of gp_prediction over posterior draws, n_obs = 10..100: inference.py:511-521,
    toy_examples.py:199): batches of small problems take the one-workgroup-per-problem route (Gram matrix, factor
    and its inverse in LDS).  Against the oracle's literal arithmetic per draw and against the batched
    Gram + gpvae_potrf route on the same inputs; every LDS tile size, ragged test tiles, the n_test == n_obs
    jitter quirk, a non-PD problem in the middle of the batch, more problems than the chip holds at once."""
    rng = np.random.default_rng(33)
    f32 = ndt == np.float32
    cases = [(1, 150, 37, 9), (2, 144, 100, 12), (1, 40, 20, 8), (2, 625, 10, 700), (1, 300, 64, 9), (1, 97, 96, 10)]
    if f32:
        cases += [(2, 625, 160, 9), (2, 400, 129, 8)]
    for d, n, n_obs, S in cases:
        g = int(round(n ** 0.5))
        x = np.arange(n) / n if d == 1 else np.stack(np.meshgrid(np.arange(g) / g, np.arange(g) / g), -1).reshape(-1, 2)
        obs = np.sort(rng.permutation(n)[:n_obs])
        y = np.full(n, np.nan)
        y[obs] = np.sin(5 * (x[obs] if d == 1 else x[obs].sum(1))) + 0.1 * rng.standard_normal(n_obs)
        mask = np.ones(n, bool); mask[obs] = False
        var = rng.uniform(0.5, 1.5, S); ls = rng.uniform(0.05, 0.3, S); sig = rng.uniform(0.01, 0.1, S)
        bad = S // 2
        sig[bad] = -10.0  # A = K + sigma I not positive definite
        xi = rng.standard_normal((S, n - n_obs))
        args = (dev(x[obs].astype(ndt), tdt), dev(x[mask].astype(ndt), tdt), dev(y[obs], tdt), var, ls, sig, 1e-5, dev(xi, tdt))
        lib_ws = ops._lib.load().gpvae_krig_batched_workspace_bytes(0 if f32 else 1, n_obs, n - n_obs, S)
        if n_obs <= (160 if f32 else 96):
            assert lib_ws == 256, (n_obs, lib_ws)  # the LDS route takes no workspace
        m, p, v, info = ops.krig_predict_batched(*args)
        ops.krig_set_path(True)
        try:
            mb, pb, vb, infob = ops.krig_predict_batched(*args)
        finally:
            ops.krig_set_path(False)
        info = info.cpu().numpy()
        assert info[bad] > 0 and (np.delete(info, bad) == 0).all() and info[bad] == infob[bad].item()
        assert np.isnan(m[bad].cpu().numpy()).all()
        for s in sorted(set([0, 1, S - 1, min(S - 1, 300)]) - {bad}):
            mo, po, vo = orc.gp_prediction(x, y, obs, var[s], ls[s], sig[s], xi[s])  # float64 truth
            if not f32:
                assert rel(m[s], mo) < 1e-7 and rel(p[s], po) < 1e-7 and rel(v[s], np.clip(vo, 0, None)) < 1e-6, (d, n_obs, s)
            else:
                f = np.float32
                m32, p32, _ = orc.gp_prediction(x.astype(f), y.astype(f), obs, f(var[s]), f(ls[s]), f(sig[s]), xi[s].astype(f))
                # the yard-stick: the worse of the oracle's own float32 run and the batched-kernel route
                bar_m = f32_bar(max(rel(m32, mo), rel(mb[s], mo))); bar_p = f32_bar(max(rel(p32, po), rel(pb[s], po)))
                assert rel(m[s], mo) < bar_m, (d, n_obs, s, rel(m[s], mo), rel(m32, mo), rel(mb[s], mo))
                assert rel(p[s], po) < bar_p, (d, n_obs, s, rel(p[s], po), rel(p32, po), rel(pb[s], po))
    # n_test == n_obs: the reference's kernel adds jitter to the (square) k_px too — inference.py:448 with gp.py:79-80
    x = np.arange(40) / 40.0
    obs = np.arange(0, 40, 2)
    y = np.full(40, np.nan); y[obs] = np.cos(3 * x[obs])
    mask = np.ones(40, bool); mask[obs] = False
    S = 8
    var = rng.uniform(0.5, 1.5, S); ls = rng.uniform(0.1, 0.3, S); sig = rng.uniform(0.05, 0.1, S)
    m, p, v, info = ops.krig_predict_batched(dev(x[obs].astype(ndt), tdt), dev(x[mask].astype(ndt), tdt), dev(y[obs], tdt), var, ls, sig, 1e-2, None)
    for s in (0, 7):
        mo, po, vo = orc.gp_prediction(x, y, obs, var[s], ls[s], sig[s], None, 1e-2)
        assert rel(m[s], mo) < (1e-9 if not f32 else 1e-4) and rel(v[s], np.clip(vo, 0, None)) < (1e-8 if not f32 else 1e-3)
        assert torch.equal(m[s], p[s])  # no xi: the sample is the mean
