"""Writes tests/golden/golden.npz: inputs and expected outputs of the hot path, computed by the CPU
oracle (oracle/gpvae_oracle.py) in float64.  The reference ships no golden vectors
and JAX/NumPyro cannot be installed here, so these vectors pin the ORACLE (against silent edits)
and give the GPU tests fixed inputs; they are not reference outputs ("parity unpinned").

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gpvae_oracle as orc  # noqa: E402


def grid2(g):
    gg = np.arange(0, 1, 1 / g)
    u, v = np.meshgrid(gg, gg)  # 2d_examples.py:562-564
    return np.stack([u.ravel(), v.ravel()], 1)


def main():
    out = {}
    rng = np.random.default_rng(20240611)
    # --- Gram ----------------------------------------------------------------------------------------
    x16 = np.arange(0, 1, 1 / 16)
    out["gram_x16"] = x16
    out["gram_k16"] = orc.exp_kernel2(x16, x16, 1.3, 0.15, 1e-4)
    x44 = grid2(4)
    out["gram_x44"] = x44
    out["gram_k44"] = orc.exp_kernel2(x44, x44, 0.9, 0.4, 1e-3)
    xa, xb = rng.random(5), rng.random(3)
    out["gram_xa"], out["gram_xb"] = xa, xb
    out["gram_rect"] = orc.exp_kernel2(xa, xb, 1.1, 0.3, 1e-2)  # rectangular: no jitter
    xc, xd = rng.random((6, 2)), rng.random((6, 2))
    out["gram_xc"], out["gram_xd"] = xc, xd
    out["gram_sqdiff"] = orc.exp_kernel2(xc, xd, 1.0, 0.5, 1e-2)  # square, different points: jitter
    # --- aggregated kernel -----------------------------------------------------------------------------
    u1 = rng.random((8, 5, 1))
    pts1 = orc.agg_points(u1, np.arange(0, 1, 1 / 8), d=1)
    out["agg_pts1"] = pts1
    out["agg_k1"] = orc.agg_kernel_from_points(pts1, 1.2, 0.2, 1e-3)
    u2 = rng.random((9, 4, 2))
    pts2 = orc.agg_points(u2, np.arange(0, 1, 1 / 3), d=2)
    out["agg_pts2"] = pts2
    out["agg_k2"] = orc.agg_kernel_from_points(pts2, 0.8, 0.3, 1e-3)
    # --- Cholesky / draw -----------------------------------------------------------------------------------
    for n, d, jit, ls in ((8, 1, 1e-4, 0.05), (64, 1, 1e-4, 0.05), (300, 1, 1e-4, 0.05), (625, 2, 1e-3, 0.1)):
        x = np.arange(0, 1, 1 / n)[:n] if d == 1 else grid2(int(round(n ** 0.5)))
        K = orc.exp_kernel2(x, x, 1.0, ls, jit)
        L = orc.cholesky_lower(K)
        out["chol_n%d_x" % n] = x
        out["chol_n%d_hyp" % n] = np.array([1.0, ls, jit])
        if n <= 64:
            out["chol_n%d_L" % n] = L
        else:  # keep the file small: pin a checksum-like digest of L
            out["chol_n%d_Ldiag" % n] = np.diag(L)
            out["chol_n%d_Lrow" % n] = L[-1]
    E, E2 = rng.standard_normal((7, 64)), rng.standard_normal((7, 64))
    x = out["chol_n64_x"]
    L = out["chol_n64_L"]
    out["draw_E"], out["draw_E2"] = E, E2
    out["draw_F"], out["draw_Y"] = orc.gp_draw(L, E, 0.07, E2)
    # --- kriging (toy_examples.py:199 observation set; n_test == n_obs quirk case) ------------------------
    x = np.arange(0, 1, 1 / 300)
    obs = np.array([20, 23, 100, 110, 117, 130, 133, 140, 170, 190])
    y = np.sin(9 * x) + 0.05 * rng.standard_normal(300)
    xi = rng.standard_normal(290)
    m, s, v = orc.gp_prediction(x, y, obs, 1.05, 0.07, 0.03, xi)
    out["krig_x"], out["krig_y"], out["krig_obs"], out["krig_xi"] = x, y, obs, xi
    out["krig_hyp"] = np.array([1.05, 0.07, 0.03])
    out["krig_mean"], out["krig_sample"], out["krig_var"] = m, s, v
    x20 = np.arange(0, 1, 1 / 20)
    obs20 = np.arange(0, 20, 2)  # 10 observed, 10 test => k_px is square and gets jitter too
    y20 = np.cos(5 * x20)
    m, s, v = orc.gp_prediction(x20, y20, obs20, 1.0, 0.2, 0.05, None)
    out["krigsq_x"], out["krigsq_y"], out["krigsq_obs"] = x20, y20, obs20
    out["krigsq_mean"], out["krigsq_var"] = m, v
    # --- ELBO / Adam / decoder (C1 dims) -------------------------------------------------------------------------
    n, h0, h1, z, B = 40, 7, 5, 3, 6
    p = orc.init_params(rng, n, h0, h1, z, np.float64)
    for k in p:
        p[k] = p[k] * 20
    Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
    loss, g = orc.elbo_loss_and_grads(p, Y, ez, 0.1)
    out["elbo_dims"] = np.array([n, h0, h1, z, B])
    out["elbo_params"], out["elbo_Y"], out["elbo_eps"] = orc.flatten(p), Y, ez
    out["elbo_loss"], out["elbo_grads"] = np.array(loss), orc.flatten(g)
    m0, v0 = 0.01 * rng.standard_normal(out["elbo_params"].size), 0.01 * rng.random(out["elbo_params"].size)
    p1, m1, v1 = orc.adam_step(out["elbo_params"], out["elbo_grads"], m0, v0, 4, 1e-3)
    out["adam_m0"], out["adam_v0"], out["adam_p1"], out["adam_m1"], out["adam_v1"] = m0, v0, p1, m1, v1
    Zc = rng.standard_normal((5, z))
    out["dec_Z"], out["dec_F"] = Zc, orc.decoder(p, Zc)
    dF = rng.standard_normal((5, n))
    out["dec_dF"], out["dec_dZ"] = dF, orc.decoder_vjp(p, Zc, dF)
    # --- Philox ------------------------------------------------------------------------------------------------------
    out["randn_f64"] = orc.randn(37, 123456789012345, 9, np.float64)
    out["randn_f32"] = orc.randn(37, 123456789012345, 9, np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
