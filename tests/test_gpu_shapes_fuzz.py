"""Seeded random-shape sweep of the entry points whose kernels switch paths with the shape (fast
aligned tiles / generic ragged tiles / padded leading dimensions / batch maps), through the C ABI,
against the CPU oracle or NumPy/LAPACK on identical inputs.  Complements the fixed cases of
test_gpu_parity.py: sizes straddle the 32 / 64 / 128 / 256 tile and block edges, leading dimensions
are padded (sub-views of wider buffers), batches straddle the 8-XCD map.

Tolerances as in test_gpu_parity.py: float64 far below the 1e-5 of BASELINE.json, float32 1e-3 or
better (the arithmetic here is well conditioned by construction, so tighter bounds are asserted).
"""
import numpy as np
import pytest
import torch

from oracle import gpvae_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [(np.float64, torch.float64), (np.float32, torch.float32)]
EDGES = [1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 191, 255, 256, 257, 300, 383, 511, 512, 513, 640, 700]


def rel(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def padded(a, tdt, pad):
    """Device copy of the 2-D / 3-D array `a` living inside a buffer whose rows are `pad` elements
    longer (leading dimension > row length), the surplus filled with NaN so that any read of it shows."""
    a = np.asarray(a)
    buf = torch.full(a.shape[:-1] + (a.shape[-1] + pad,), float("nan"), dtype=tdt, device=DEV)
    view = buf[..., :a.shape[-1]]
    view.copy_(torch.tensor(a, dtype=tdt, device=DEV))
    return view


@pytest.fixture(scope="module")
def ops():
    from gp_vae_amd import ops as o
    assert torch.cuda.is_available()
    return o


def spd(rng, batch, n):
    """Well-conditioned SPD matrices (cond about 10) so that float32 forward errors stay at 1e-5."""
    M = rng.standard_normal((batch, n, max(2 * n, 4)))
    A = M @ M.transpose(0, 2, 1) / M.shape[2]
    return A + 0.5 * np.eye(n)


@pytest.mark.parametrize("ndt,tdt", DTYPES)
def test_potrf_random_shapes_with_extra_rows(ops, ndt, tdt):
    rng = np.random.default_rng(11)
    tol = 1e-11 if ndt == np.float64 else 2e-5
    for case in range(40):
        n = int(rng.choice(EDGES))
        batch = int(rng.choice([1, 2, 3, 7, 8, 9, 17]))
        if batch * n * n > 6e6:
            batch = max(1, int(6e6 // (n * n)))
        m = int(rng.choice([0, 1, 2, 63, 64, 65, 130, 300]))
        pad = int(rng.choice([0, 4, 12]))
        A = spd(rng, batch, n)
        X = rng.standard_normal((batch, m, n)) if m else None
        Ad = padded(A.astype(ndt), tdt, pad)
        Xd = padded(X.astype(ndt), tdt, pad) if m else None
        info = ops.potrf_(Ad, Xd)
        assert (info == 0).all(), (case, n, batch, m, pad)
        L = np.linalg.cholesky(A)
        assert rel(Ad, L) < tol, (case, n, batch, m, pad, rel(Ad, L))
        if m:
            want = np.stack([np.linalg.solve(L[b], X[b].T).T for b in range(batch)])  # X L^-T
            assert rel(Xd, want) < 10 * tol, (case, n, batch, m, pad, rel(Xd, want))


@pytest.mark.parametrize("ndt,tdt", DTYPES)
def test_draw_and_trsm_random_shapes(ops, ndt, tdt):
    rng = np.random.default_rng(12)
    tol = 1e-12 if ndt == np.float64 else 1e-5
    for case in range(40):
        n, nb = int(rng.choice(EDGES)), int(rng.choice(EDGES))
        pad = int(rng.choice([0, 4, 8]))
        L = np.tril(rng.standard_normal((n, n))) + 4.0 * np.sqrt(n) * np.eye(n)
        E, E2 = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
        Fo, Yo = orc.gp_draw(L, E, 0.3, E2)
        junk = np.triu(np.full((n, n), 9.0), 1)  # above the diagonal: must never be read as data
        Ld = padded((L + junk).astype(ndt), tdt, pad)
        Ed, E2d = padded(E.astype(ndt), tdt, pad), padded(E2.astype(ndt), tdt, pad)
        F, Y = ops.draw(Ld, Ed, E2d, 0.3, want_f=True)
        assert rel(F, Fo) < tol and rel(Y, Yo) < tol, (case, n, nb, pad, rel(F, Fo), rel(Y, Yo))
        # the stand-alone solve against the same factor: X L^-T then X L^-1 = X (L L^T)^-1
        Xd = padded(E.astype(ndt), tdt, pad)
        ops.trsm_(Ld, Xd, trans=True)
        w1 = np.linalg.solve(L, E.T).T
        assert rel(Xd, w1) < 20 * tol, (case, n, nb, pad, "trans", rel(Xd, w1))
        ops.trsm_(Ld, Xd, trans=False)
        w2 = np.linalg.solve(L.T, w1.T).T
        assert rel(Xd, w2) < 400 * tol, (case, n, nb, pad, "notrans", rel(Xd, w2))


@pytest.mark.parametrize("ndt,tdt", DTYPES)
def test_vae_elbo_random_shapes(ops, ndt, tdt):
    rng = np.random.default_rng(13)
    t = 1e-10 if ndt == np.float64 else 2e-4
    for case in range(24):
        B = int(rng.choice([1, 7, 16, 33, 128, 200, 256, 384]))
        n = int(rng.choice([1, 19, 64, 100, 192, 300, 625, 1024]))
        h0 = int(rng.choice([1, 3, 35, 50, 63, 64, 65, 200]))
        h1 = int(rng.choice([1, 2, 30, 64, 129]))
        z = int(rng.choice([1, 2, 10, 15, 40]))
        p = {k: (6 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
        Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
        lo, go = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
        flat = torch.tensor(orc.flatten(p), dtype=tdt, device=DEV)
        grads = torch.empty_like(flat)
        Yd = torch.tensor(Y, dtype=tdt, device=DEV)
        ezd = torch.tensor(ez, dtype=tdt, device=DEV)
        loss = ops.vae_elbo(flat, Yd, ezd, (h0, h1, z), 0.1, grads=grads)
        tag = (case, B, n, h0, h1, z)
        assert abs(loss.item() - lo) < t * abs(lo), tag
        gd = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
        for k in orc.PARAM_NAMES:
            assert rel(gd[k], go[k]) < t, tag + (k, rel(gd[k], go[k]))
        # evaluate-only call (no gradient buffers): same loss
        l2 = ops.vae_elbo(flat, Yd, ezd, (h0, h1, z), 0.1)
        assert abs(l2.item() - lo) < t * abs(lo), tag


@pytest.mark.parametrize("ndt,tdt", DTYPES)
def test_kriging_random_shapes(ops, ndt, tdt):
    rng = np.random.default_rng(14)
    for case in range(16):
        n_obs = int(rng.choice([1, 5, 63, 64, 100, 257, 400]))
        n_test = int(rng.choice([1, 17, 64, 130, 625]))
        d = int(rng.choice([1, 2]))
        x = rng.random((n_obs + n_test, d))
        obs = np.sort(rng.permutation(n_obs + n_test)[:n_obs])
        y = np.sin(4 * x[:, 0]) + 0.1 * rng.standard_normal(n_obs + n_test)
        var, ls, sigma = 1.0 + rng.random(), 0.1 + 0.4 * rng.random(), 0.05 + 0.1 * rng.random()
        xi = rng.standard_normal(n_test)
        mo, so, vo = orc.gp_prediction(x, y, obs, var, ls, sigma, xi)
        mask = np.ones(n_obs + n_test, bool)
        mask[obs] = False
        dv = lambda a: torch.tensor(np.asarray(a), dtype=tdt, device=DEV)
        md, sd, vd, info = ops.krig_predict(dv(x[obs]), dv(x[mask]), dv(y[obs]), var, ls, sigma, 1e-5, xi=dv(xi))
        tag = (case, n_obs, n_test, d)
        assert info.item() == 0, tag
        if ndt == np.float64:
            assert rel(md, mo) < 1e-7 and np.abs(vd.cpu().numpy() - vo).max() < 1e-8 and rel(sd, so) < 1e-6, tag
        else:  # f32: 1e-3 (north_star), or 3x what the oracle's arithmetic loses in float32 on these inputs
            f = np.float32
            m32, _, v32 = orc.gp_prediction(x.astype(f), y.astype(f), obs, f(var), f(ls), f(sigma), xi.astype(f))
            assert rel(md, mo) < max(1e-3, 3 * rel(m32, mo)), tag + (rel(md, mo), rel(m32, mo))
            ev, ev32 = np.abs(vd.cpu().numpy() - vo).max() / var, np.abs(v32 - vo).max() / var
            assert ev < max(1e-3, 3 * ev32), tag + (ev, ev32)


@pytest.mark.parametrize("ndt,tdt", DTYPES)
def test_decoder_random_shapes(ops, ndt, tdt):
    rng = np.random.default_rng(15)
    t = 1e-11 if ndt == np.float64 else 1e-4
    for case in range(20):
        C_ = int(rng.choice([1, 3, 16, 17, 100, 256, 300]))
        n = int(rng.choice([1, 19, 64, 100, 300, 625, 1024]))
        h0 = int(rng.choice([1, 3, 35, 50, 64, 65, 200, 600]))
        h1 = int(rng.choice([1, 2, 30, 64, 129, 300]))
        z = int(rng.choice([1, 2, 10, 15, 40]))
        p = {k: (6 * v) for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
        Z, dF = rng.standard_normal((C_, z)), rng.standard_normal((C_, n))
        dec = torch.tensor(np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]),
                           dtype=tdt, device=DEV)
        Zd, dFd = torch.tensor(Z, dtype=tdt, device=DEV), torch.tensor(dF, dtype=tdt, device=DEV)
        tag = (case, C_, n, h0, h1, z)
        assert rel(ops.decoder_fwd(dec, Zd, n, h0, h1), orc.decoder(p, Z)) < t, tag
        assert rel(ops.decoder_vjp(dec, Zd, dFd, h0, h1), orc.decoder_vjp(p, Z, dF)) < t, tag


def test_mvn_logp_random_shapes_f64(ops):
    """float64 only: in float32 the gradient bounds depend on cond(K) (test_gpu_parity.py states them
    per size); the path taken through the kernels is the same for both types."""
    rng = np.random.default_rng(16)
    for case in range(10):
        n = int(rng.choice([1, 2, 31, 64, 65, 128, 150, 257, 400]))
        d = int(rng.choice([1, 2]))
        S = int(rng.choice([1, 2, 7, 9]))
        x = rng.random((n, d))
        var = rng.uniform(0.7, 1.4, S); ls = rng.uniform(0.1, 0.3, S)
        jit = 1e-4
        F = np.stack([np.linalg.cholesky(orc.exp_kernel2(x, x, var[s], ls[s], jit)) @ rng.standard_normal(n) for s in range(S)])
        xd = torch.tensor(x if d == 2 else x[:, 0], dtype=torch.float64, device=DEV)
        lp, gf, gv, gl, info = ops.mvn_logp(xd, torch.tensor(F, dtype=torch.float64, device=DEV), var, ls, jit)
        assert (info == 0).all(), (case, n, d, S)
        for s in range(S):
            lo, gfo, gvo, glo = orc.mvn_logp_grads(x if d == 2 else x[:, 0], F[s], var[s], ls[s], jit)
            tag = (case, n, d, S, s)
            assert abs(lp[s].item() - lo) < 1e-8 * max(1.0, abs(lo)), tag
            assert rel(gf[s], gfo) < 1e-6, tag + (rel(gf[s], gfo),)
            assert abs(gv[s].item() - gvo) < 1e-6 * max(1.0, abs(gvo)) and abs(gl[s].item() - glo) < 1e-6 * max(1.0, abs(glo)), tag


def test_kriging_batched_random_shapes_f64(ops):
    rng = np.random.default_rng(17)
    for case in range(8):
        n_obs = int(rng.choice([1, 5, 63, 64, 100, 257]))
        n_test = int(rng.choice([1, 17, 64, 130, 300]))
        d = int(rng.choice([1, 2]))
        S = int(rng.choice([1, 3, 8, 11]))
        x = rng.random((n_obs + n_test, d))
        obs = np.sort(rng.permutation(n_obs + n_test)[:n_obs])
        mask = np.ones(n_obs + n_test, bool)
        mask[obs] = False
        y = np.sin(4 * x[:, 0]) + 0.1 * rng.standard_normal(n_obs + n_test)
        var = rng.uniform(0.5, 1.5, S); ls = rng.uniform(0.1, 0.4, S); sig = rng.uniform(0.05, 0.15, S)
        xi = rng.standard_normal((S, n_test))
        dv = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=DEV)
        m, p, v, info = ops.krig_predict_batched(dv(x[obs]), dv(x[mask]), dv(y[obs]), var, ls, sig, 1e-5, dv(xi))
        assert (info == 0).all(), (case, n_obs, n_test, d, S)
        for s in range(S):
            mo, po, vo = orc.gp_prediction(x, y, obs, var[s], ls[s], sig[s], xi[s])
            tag = (case, n_obs, n_test, d, S, s)
            assert rel(m[s], mo) < 1e-7 and rel(p[s], po) < 1e-6, tag
            assert np.abs(v[s].cpu().numpy() - vo).max() < 1e-8, tag


def test_gp_potential_random_shapes_f64(ops):
    """gpvae_gp_potential over ragged sizes, both point dimensions, padded q / dU rows and ragged chain
    counts: the potential against the oracle's gp_regression potential, its gradient against central
    differences of the same oracle (a few coordinates per case), and chain c of a batch against the same chain
    evaluated alone (fixed-order sums: bit for bit)."""
    rng = np.random.default_rng(23)
    for case in range(8):
        n = int(rng.choice([1, 3, 31, 64, 65, 129, 200]))
        d = int(rng.choice([1, 2]))
        C = int(rng.choice([1, 2, 5, 9]))
        x = rng.random((n, d))
        y = np.sin(5 * x[:, 0]) + 0.1 * rng.standard_normal(n)
        uv, ul, us = 0.1 * rng.standard_normal(C), np.log(rng.uniform(0.15, 0.4, C)), np.log(rng.uniform(0.05, 0.2, C))
        F = np.stack([np.linalg.cholesky(orc.exp_kernel2(x, x, np.exp(uv[c]), np.exp(ul[c]), 1e-5)) @ rng.standard_normal(n) for c in range(C)])
        q = np.concatenate([F, uv[:, None], ul[:, None], us[:, None]], 1)
        xs = x if d == 2 else x[:, 0]
        xd = torch.tensor(xs, dtype=torch.float64, device=DEV)
        qd = padded(q, torch.float64, 5)
        U, dU, info = ops.gp_potential(xd, qd, torch.tensor(y, dtype=torch.float64, device=DEV))
        tag = (case, n, d, C)
        assert (info == 0).all(), tag
        for c in range(C):
            want = orc.gp_regression_potential(xs, y, q[c, :n], uv[c], ul[c], us[c])
            assert abs(U[c].item() - want) < 1e-8 * max(1.0, abs(want)), tag + (c,)
        c = int(rng.integers(C))
        for j in {0, n // 2, n - 1, n, n + 1, n + 2}:
            h = 1e-6
            qp, qm = q[c].copy(), q[c].copy()
            qp[j] += h; qm[j] -= h
            fd = (orc.gp_regression_potential(xs, y, qp[:n], qp[n], qp[n + 1], qp[n + 2])
                  - orc.gp_regression_potential(xs, y, qm[:n], qm[n], qm[n + 1], qm[n + 2])) / (2 * h)
            assert abs(dU[c, j].item() - fd) < 2e-4 * max(1.0, abs(fd)), tag + (c, j, dU[c, j].item(), fd)
        U1, dU1, _ = ops.gp_potential(xd, qd[c:c + 1], torch.tensor(y, dtype=torch.float64, device=DEV))
        assert torch.equal(U1[0], U[c]) and torch.equal(dU1[0], dU[c]), tag


def test_potrf_b3_and_draw_b3_random_shapes(ops):
    """The bf16 x 3 route over its supported shapes (n a multiple of 256 from 512 up; batches that straddle
    the 8-XCD tile map; A with a padded leading dimension; draw batches that are multiples of 128 and of 256):
    factor against LAPACK on well-conditioned SPD matrices, planes == factor, and the draw from the planes
    against a float64 product — all at the float32 bar of the f32-MFMA kernels."""
    rng = np.random.default_rng(29)
    for case in range(6):
        n = int(rng.choice([512, 768, 1024, 1536]))
        batch = int(rng.choice([1, 2, 7, 8, 9, 17]))
        nb = int(rng.choice([128, 256, 384, 640]))
        A = spd(rng, batch, n)
        Ad = padded(A, torch.float32, int(rng.choice([0, 4, 12])))
        info, L3 = ops.potrf_b3_(Ad)
        tag = (case, n, batch, nb)
        assert (info == 0).all(), tag
        for b in {0, batch // 2, batch - 1}:
            Lo = np.linalg.cholesky(A[b])
            Ld = torch.tril(Ad[b])
            assert rel(Ld, Lo) < 2e-5, tag + (b, rel(Ld, Lo))
            assert torch.equal(torch.tril(L3[b].float().sum(0)), Ld), tag  # entries are O(1): no piece underflows
            E3 = ops.randn_b3((nb, n), DEV, seed=case, stream_id=b)
            E2 = ops.randn((nb, n), torch.float32, DEV, seed=case, stream_id=100 + b)
            Y = ops.draw_b3(L3[b], E3, E2, 0.25)
            want = E3.float().sum(0).double() @ Ld.double().T + 0.25 * E2.double()
            assert rel(Y, want.cpu().numpy()) < 1e-6, tag + (b,)


def test_potrf_share_chip_random_shapes(ops):
    """GPVAE_POTRF_SHARE_CHIP over shapes on both sides of its switch (the rows below an outer block go to
    potrf_rows_kernel from 128 workgroups of rows up: batch x rows / 128), padded leading dimensions included: same
    info, factor against LAPACK at the bar of the plain call and within rounding of it, planes == factor."""
    rng = np.random.default_rng(37)
    for case, (n, batch) in enumerate([(2048, 12), (1024, 40), (1536, 17), (768, 64), (512, 70), (2304, 9)]):
        A = spd(rng, batch, n)
        pad = int(rng.choice([0, 4, 12]))
        Ad, Ap = padded(A, torch.float32, pad), padded(A, torch.float32, pad)
        info, L3 = ops.potrf_b3_(Ad, share_chip=True)
        info_p, _ = ops.potrf_b3_(Ap)
        tag = (case, n, batch, pad)
        assert (info == 0).all() and (info_p == 0).all(), tag
        for b in {0, batch // 2, batch - 1}:
            Lo = np.linalg.cholesky(A[b])
            Ld = torch.tril(Ad[b])
            assert rel(Ld, Lo) < 2e-5, tag + (b, rel(Ld, Lo))
            assert (Ld - torch.tril(Ap[b])).abs().max().item() <= 4e-6 * Ld.abs().max().item(), tag
            assert torch.equal(torch.tril(L3[b].float().sum(0)), Ld), tag


def test_potrf_h2_and_draw_h2_random_shapes(ops):
    """The opt-in fp16 x 2 route over the same shapes, with matrix scales from 1e-3 to 1e3 (the planes' power-of-two
    scale is chosen per matrix from its diagonal): factor against LAPACK, planes == factor * 2^e to 2^-21, draw from
    the planes against a float64 product at the float32 bar."""
    rng = np.random.default_rng(31)
    for case in range(6):
        n = int(rng.choice([512, 768, 1024, 1536]))
        batch = int(rng.choice([1, 2, 7, 8, 9, 17]))
        nb = int(rng.choice([128, 256, 384, 640]))
        scale = 10 ** rng.uniform(-3, 3, batch)
        if case == 5:
            scale[0], scale[-1] = 1e-30, 1e30   # the planes' exponent at both ends of float32's range
        A = spd(rng, batch, n) * scale[:, None, None]
        Ad = padded(A, torch.float32, int(rng.choice([0, 4, 12])))
        info, L2, inv = ops.potrf_h2_(Ad)
        tag = (case, n, batch, nb)
        assert (info == 0).all(), tag
        for b in {0, batch // 2, batch - 1}:
            Lo = np.linalg.cholesky(A[b])
            Ld = torch.tril(Ad[b])
            assert rel(Ld, Lo) < 2e-5, tag + (b, rel(Ld, Lo))
            top = float(np.sqrt(A[b].diagonal().max()))
            s = 1.0 / inv[b].item()
            clamped = s in (2.0 ** 62, 2.0 ** -62)   # the exponent is limited so that s^2 stays a float
            assert (clamped or 2.0 ** 13 * (1 - 1e-6) < top * s) and top * s <= 2.0 ** 14 * (1 + 1e-6), tag + (top, s)
            back = torch.tril(L2[b].float().sum(0).double() / s)
            assert ((back - Ld.double()).abs() <= 2.0 ** -21 * Ld.double().abs() + 2.0 ** -36 * top).all(), tag
            E2h = ops.randn_h2((nb, n), DEV, seed=case, stream_id=b)
            E2 = ops.randn((nb, n), torch.float32, DEV, seed=case, stream_id=100 + b)
            Y = ops.draw_h2(L2[b], inv[b:b + 1], E2h, E2, 0.25 * top)
            want = (E2h.float().sum(0).double() / 4096.0) @ Ld.double().T + 0.25 * top * E2.double()
            assert rel(Y, want.cpu().numpy()) < 2e-6, tag + (b, rel(Y, want.cpu().numpy()))
