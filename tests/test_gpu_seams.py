"""GPU tests of the reference's call seams and of the entry points added for them:
Predictive(gp.sample) with per-sample hyper-parameters (batched factors + gpvae_draw_batched),
vae_encoder's apply (gpvae_encoder_fwd), SVI.init/update/evaluate/get_params driven by an EXTERNAL
batch, the literal training loop vs the pipelined StepEngine, regression / gp_regression handed to the
MCMC drivers, chain sharding.  Everything goes ctypes -> C ABI; the checker is the oracle."""
import math
import numpy as np
import pytest
import torch

from oracle import gpvae_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a, tdt):
    return torch.tensor(np.asarray(a), dtype=tdt, device=DEV)


@pytest.mark.parametrize("ndt,tdt,tol", [(np.float64, torch.float64, 1e-12), (np.float32, torch.float32, 1e-5)])
def test_draw_batched_against_oracle(ndt, tdt, tol):
    """gpvae_draw_batched: one draw per factor, ragged n, padded leading dimensions, junk above the
    diagonal, with and without the noise term."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(3)
    for n, nb in ((1, 1), (63, 7), (64, 3), (300, 100), (517, 9)):
        L = np.tril(rng.standard_normal((nb, n, n)))
        E, E2 = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
        sig = rng.uniform(0.01, 0.3, nb)
        buf = torch.zeros((nb, n, n + 3), dtype=tdt, device=DEV)
        buf[:, :, :n] = dev(L, tdt) + torch.triu(torch.full((n, n), 9.0, dtype=tdt, device=DEV), 1)
        F, Y = ops.draw_batched(buf[:, :, :n], dev(E, tdt), dev(E2, tdt), dev(sig, tdt), want_f=True)
        Fo = np.einsum("bij,bj->bi", L, E)
        assert rel(F, Fo) < tol and rel(Y, Fo + sig[:, None] * E2) < tol, (n, nb)
        assert rel(ops.draw_batched(buf[:, :, :n], dev(E, tdt)), Fo) < tol


@pytest.mark.parametrize("ndt,tdt,tol", [(np.float64, torch.float64, 1e-9), (np.float32, torch.float32, 1e-3)])
def test_predictive_per_sample_hyperparameters_c1(ndt, tdt, tol):
    """Predictive(gp.sample, B)(key, x=x) with ls/var/sigma left None (vae.py:226, toy_examples.py:88):
    B = 100 distinct K / L at n = 300 (C1) — f, y and the length / var / noise sites vs the oracle."""
    from gp_vae_amd import ops
    from gp_vae_amd.gp import GP, Predictive, draw_hyper, exp_kernel2
    n, B, jit, seed = 300, 100, 1e-4, 11
    x = np.arange(0, 1, 1 / n)
    gp = GP(kernel=exp_kernel2, jitter=jit, d=1)
    out = Predictive(gp.sample, num_samples=B)(seed, x=torch.tensor(x, dtype=tdt))
    ls, var, sig = draw_hyper(np.random.default_rng(seed), B)  # the documented host stream of the sites
    assert rel(out["length"], ls) < 1e-6 and rel(out["var"], var) < 1e-6 and rel(out["noise"], sig) < 1e-6
    assert int(gp.last_info.abs().sum().item()) == 0
    E = ops.randn((B, n), tdt, DEV, seed=seed, stream_id=1).cpu().numpy().astype(np.float64)
    E2 = ops.randn((B, n), tdt, DEV, seed=seed, stream_id=2).cpu().numpy().astype(np.float64)
    worst_f = worst_o32 = 0.0
    for b in range(B):
        hy = [ndt(v) for v in (var[b], ls[b], sig[b])]  # the device sees the hyper-parameters in its dtype
        L = orc.cholesky_lower(orc.exp_kernel2(x, x, float(hy[0]), float(hy[1]), jit))
        Fo, Yo = orc.gp_draw(L, E[b:b + 1], float(hy[2]), E2[b:b + 1])
        worst_f = max(worst_f, rel(out["f"][b], Fo[0]), rel(out["y"][b], Yo[0]))
        if ndt == np.float32:  # yard-stick: LAPACK's own float32 factor of the same matrix
            K32 = orc.exp_kernel2(x.astype(ndt), x.astype(ndt), hy[0], hy[1], ndt(jit)).astype(ndt)
            F32 = E[b].astype(ndt) @ np.linalg.cholesky(K32).T
            worst_o32 = max(worst_o32, rel(F32, Fo[0]))
    assert worst_f < max(tol, 3 * worst_o32), (worst_f, worst_o32)
    # fixed ls, sampled var / noise: only the sampled sites are returned (gp.py:171-176)
    out2 = Predictive(gp.sample, num_samples=5)(3, x=torch.tensor(x, dtype=tdt), ls=0.2)
    assert "length" not in out2 and out2["var"].shape == (5,) and out2["noise"].shape == (5,)


@pytest.mark.parametrize("ndt,tdt,tol", [(np.float64, torch.float64, 1e-11), (np.float32, torch.float32, 1e-4)])
def test_encoder_apply_and_guide(ndt, tdt, tol):
    from gp_vae_amd import ops
    from gp_vae_amd.gp import GP
    from gp_vae_amd.vae import VAE, encoder_list_from_flat
    rng = np.random.default_rng(4)
    for B, n, h0, h1, z in ((100, 300, 35, 30, 10), (37, 625, 50, 30, 15), (5, 19, 3, 2, 1), (64, 256, 512, 256, 64)):
        p = {k: 8 * v for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
        Y, ez = rng.standard_normal((B, n)), rng.standard_normal((B, z))
        mu_o, sd_o = orc.encoder(p, Y)
        flat = dev(orc.flatten(p), tdt)
        mu, sd, zz = ops.encoder_fwd(flat, dev(Y, tdt), (h0, h1, z), dev(ez, tdt))
        assert rel(mu, mu_o) < tol and rel(sd, sd_o) < tol and rel(zz, mu_o + sd_o * ez) < tol, (B, n)
        vae = VAE(GP(), [h0, h1], z, n, B, 1e-3, 1, 1, 1, np.arange(n) / n, dtype=tdt)
        init, apply = vae.vae_encoder()
        (shape_mu, shape_sd), plist = init(3, (B, n))
        assert shape_mu == (z,) and plist[0][0].shape == (n, h0) and plist[5][1][0][0].shape == (h1, z)
        m2, s2 = apply(encoder_list_from_flat(flat, n, h0, h1, z), dev(Y, tdt))
        assert torch.equal(m2, mu) and torch.equal(s2, sd)
        assert apply(encoder_list_from_flat(flat, n, h0, h1, z), dev(Y[0], tdt))[0].shape == (z,)
        # vae_guide: z ~ Normal(z_loc, z_std) with the key's normals; vae_model returns the observed batch
        zg = vae.vae_guide(dev(Y, tdt), params=flat, rng_key=9)
        eg = ops.randn((B, z), tdt, DEV, seed=9, stream_id=3).cpu().numpy().astype(np.float64)
        assert rel(zg, mu_o + sd_o * eg) < tol
        obs = vae.vae_model(dev(Y, tdt), params=flat, z=zg)
        assert torch.equal(obs, dev(Y, tdt)) and rel(vae.last_f, orc.decoder(p, zg.cpu().numpy().astype(np.float64))) < tol


@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float64, np.float64, 1e-8), (torch.float32, np.float32, 2e-3)])
def test_svi_seam_c1_epoch_matches_oracle(tdt, ndt, tol):
    """A C1 epoch (n=300, B=100, 35/30/10, jitter 1e-4) driven as the reference drives it —
    hyper draw -> Predictive(gp.sample) -> svi.update(state, batch['y']) — NOT through StepEngine;
    losses and parameters vs oracle.train_step on the same inputs; evaluate / get_params."""
    from gp_vae_amd import ops
    from gp_vae_amd.gp import GP, PhiloxKey, Predictive, draw_hyper, exp_kernel2
    from gp_vae_amd.vae import SVI, VAE, Adam, Trace_ELBO
    n, B, dims, lr, jit = 300, 100, (35, 30, 10), 1e-3, 1e-4
    x = np.arange(0, 1, 1 / n)
    gp = GP(kernel=exp_kernel2, jitter=jit, d=1)
    vae = VAE(gp, list(dims[:2]), dims[2], n, B, lr, 1, 6, 2, x, seed=1, dtype=tdt)
    svi = SVI(vae.vae_model, vae.vae_guide, Adam(lr), Trace_ELBO())
    gp_predictive = Predictive(gp.sample, num_samples=B)
    state = svi.init(5, gp_predictive(rng_key=1, x=vae.x)["y"])
    assert state.t == 0 and state.params.numel() == ops.vae_param_count(n, *dims)
    p = orc.unflatten(state.params.cpu().numpy().astype(np.float64), n, *dims)
    p = {k: 5 * v for k, v in p.items()}  # larger weights: gradients well above rounding
    state.params.mul_(5)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(a) for k, a in p.items()}
    rng = np.random.default_rng(0)
    for t in range(1, 7):
        ls, var, sig = (float(ndt(a)) for a in draw_hyper(rng))
        key = PhiloxKey(77, t << 4)
        batch = gp_predictive(key, vae.x, ls=ls, var=var, sigma=sig)
        state, loss = svi.update(state, batch["y"], rng_key=key)
        E, E2, ez = (ops.randn(s, tdt, DEV, seed=77, stream_id=(t << 4) | i).cpu().numpy().astype(np.float64)
                     for i, s in ((1, (B, n)), (2, (B, n)), (3, (B, dims[2]))))
        want, p, m, v = orc.train_step(p, m, v, t, x, (ls, var, sig), E, E2, ez, lr, jit)
        assert abs(loss.item() - want) < tol * abs(want), t
    assert state.t == 6 and rel(state.params, orc.flatten(p)) < tol
    # evaluate leaves the state alone and equals the oracle's loss on an external batch
    Yext, key = np.random.default_rng(1).standard_normal((B, n)), PhiloxKey(5, 0x70)
    before = state.params.clone()
    le = svi.evaluate(state, dev(Yext, tdt), rng_key=key).item()
    ez = ops.randn((B, dims[2]), tdt, DEV, seed=5, stream_id=0x73).cpu().numpy().astype(np.float64)
    pe = orc.unflatten(state.params.cpu().numpy().astype(np.float64), n, *dims)
    assert abs(le - orc.elbo_loss(pe, Yext, ez)) < max(tol, 1e-4) * abs(le) and torch.equal(before, state.params)
    # the default key stream advances with the step count: two updates on one batch use different normals
    l1 = svi.evaluate(state, dev(Yext, tdt)).item()
    state.t += 1
    assert svi.evaluate(state, dev(Yext, tdt)).item() != l1
    state.t -= 1
    got = svi.get_params(state)
    assert [tuple(a.shape) for a in (got["decoder$params"][0][0], got["decoder$params"][4][0])] == [(10, 30), (35, 300)]
    assert got["encoder$params"][0][0].shape == (300, 35)
    # only this model / guide / optimiser / loss
    with pytest.raises(NotImplementedError):
        SVI(lambda b: b, vae.vae_guide, Adam(lr), Trace_ELBO())
    with pytest.raises(NotImplementedError):
        SVI(vae.vae_model, VAE(gp, [35, 30], 10, n, B, lr, 1, 1, 1, x).vae_guide, Adam(lr), Trace_ELBO())


def test_literal_loop_equals_pipelined_engine():
    """VAE(pipelined=False) — draw hyper -> gp_predictive -> svi.update, one launch sequence per step —
    and the pipelined StepEngine (factor stream, batched factors) are the same arithmetic on the same
    keys: epoch_train and eval_test agree (f64: to rounding of the atomics' order)."""
    from gp_vae_amd.gp import GP, exp_kernel2
    from gp_vae_amd.vae import VAE
    n = 96
    x = np.arange(0, 1, 1 / n)
    from gp_vae_amd.gp import draw_hyper
    losses = {}
    for sampler in ("device", draw_hyper):  # gpvae_hyper_sample (VAE's default, what bench.py times) and the host's NumPy draws
        res = []
        for pipelined in (True, False):
            gp = GP(kernel=exp_kernel2, jitter=1e-3, d=1)
            vae = VAE(gp, [12, 9], 4, n, 40, 1e-3, 1, 7, 3, x, seed=3, dtype=torch.float64, pipelined=pipelined,
                      hyper_sampler=sampler)
            st = vae.svi_init()
            st.params.mul_(5)
            loss, st = vae.epoch_train(123, st)
            res.append((loss.item(), st.params.clone(), vae.eval_test(9, st), st.t))
        assert res[0][3] == res[1][3] == 7
        assert abs(res[0][0] - res[1][0]) < 1e-10 * abs(res[0][0])
        assert rel(res[0][1], res[1][1].cpu().numpy()) < 1e-10
        assert abs(res[0][2] - res[1][2]) < 1e-10 * abs(res[0][2])
        losses[sampler if isinstance(sampler, str) else "host"] = res[0][0]
    assert losses["device"] != losses["host"]  # different generators: different (ls, var, sigma) sequences
    assert VAE(GP(kernel=exp_kernel2, jitter=1e-3, d=1), [12, 9], 4, n, 40, 1e-3, 1, 7, 3, x)._engine().hyper_sampler == "device"
    with pytest.raises(ValueError):
        VAE(GP(kernel=exp_kernel2, jitter=1e-3, d=1), [12, 9], 4, n, 40, 1e-3, 1, 7, 3, x, hyper_sampler="host")


def test_step_engine_2d_points_any_strides_and_kernel_guard():
    """a transposed (column-major) [n,2] x — the scripts' array([u,v]).transpose idiom
    (2d_examples.py:564) — must train on the same locations as its contiguous copy, and match the
    oracle; a GP with another kernel must not silently train on the squared-exponential one."""
    from gp_vae_amd import ops, parallel
    from gp_vae_amd.gp import GP, draw_hyper
    from gp_vae_amd.vae import StepEngine, SVIState, VAE
    g = np.arange(0, 1, 1 / 12)
    u, v = np.meshgrid(g, g)
    xt = torch.tensor(np.array([u.flatten(), v.flatten()]), dtype=torch.float64, device=DEV).transpose(1, 0)
    assert not xt.is_contiguous() and xt.shape == (144, 2)
    xc = xt.contiguous()
    n, B, dims = 144, 32, (10, 8, 3)
    out = []
    for x in (xt, xc):
        eng = StepEngine(x, 1e-3, dims, B, 1e-3, dtype=torch.float64, seed=2, group=2)
        p = orc.init_params(np.random.default_rng(0), n, *dims, np.float64)
        st = SVIState(torch.tensor(orc.flatten(p), dtype=torch.float64, device=DEV) * 5)
        out.append((eng.run([(4, 0), (4, 1)], st, train=True).item(), st.params.clone()))
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-12) and rel(out[0][1], out[1][1].cpu().numpy()) < 1e-12
    p = {k: 5 * a for k, a in orc.init_params(np.random.default_rng(0), n, *dims, np.float64).items()}
    m = {k: np.zeros_like(a) for k, a in p.items()}
    vv = {k: np.zeros_like(a) for k, a in p.items()}
    want = 0.0
    for t, key in enumerate([(4, 0), (4, 1)], 1):
        E, E2, ez = (ops.randn(s, torch.float64, DEV, seed=2, stream_id=parallel.philox_stream_id(0, key, i)).cpu().numpy()
                     for i, s in ((1, (B, n)), (2, (B, n)), (3, (B, dims[2]))))
        loss, p, m, vv = orc.train_step(p, m, vv, t, xc.cpu().numpy(), draw_hyper(parallel.step_rng(2, 0, key)), E, E2, ez, 1e-3, 1e-3)
        want += loss
    assert abs(out[0][0] - want) < 1e-8 * abs(want) and rel(out[0][1], orc.flatten(p)) < 1e-8
    with pytest.raises(NotImplementedError):
        StepEngine(xc, 1e-3, dims, B, 1e-3, kernel=lambda *a, **k: None)
    other = GP(kernel=lambda x, z, var, ls, jitter=1e-5: ops.gram_se(x, z, var, ls, jitter), jitter=1e-3, d=2)
    vae = VAE(other, [10, 8], 3, n, B, 1e-3, 1, 2, 1, xc, dtype=torch.float64)
    with pytest.raises(NotImplementedError):
        vae.epoch_train(1, vae.svi_init())
    # ... but the literal loop honours an injected kernel (here: the same arithmetic through a wrapper)
    vae2 = VAE(other, [10, 8], 3, n, B, 1e-3, 1, 2, 1, xc, seed=2, dtype=torch.float64, pipelined=False)
    loss, st = vae2.epoch_train(1, vae2.svi_init())
    assert np.isfinite(loss.item()) and st.t == 2 and eng.failed_factors() == 0


def test_regression_seams_and_chain_sharding(monkeypatch):
    """regression / gp_regression called as the reference calls them: with data they return what the
    MCMC driver is given; through Predictive they return prior / posterior predictive draws;
    gp_run_mcmc(rng_key, x, y, model) keeps the reference's argument list; chains sharded over two
    (emulated) ranks reproduce the single-process chains exactly."""
    from gp_vae_amd import parallel
    from gp_vae_amd.gp import Predictive, exp_kernel2
    from gp_vae_amd.inference import GPInference, Potential, VAEInference
    rng = np.random.default_rng(2)
    n = 40
    x = np.arange(n) / n
    obs = np.sort(rng.permutation(n)[:12])
    y = np.full(n, np.nan); y[obs] = np.sin(6 * x[obs]) + 0.05 * rng.standard_normal(12)
    args = {"num_warmup": 30, "num_samples": 20, "num_chains": 4, "thinning": 1, "max_tree_depth": 5}
    inf = GPInference(exp_kernel2, 1, x, y, obs, args, seed=5, dtype=torch.float64)
    pot = inf.gp_regression(inf.x[inf.obs_idx], inf.y[inf.obs_idx])
    assert isinstance(pot, Potential) and pot.init.shape == (15,)
    q = pot.init[None, :].repeat(3, 1) + 0.01 * torch.randn(3, 15, dtype=torch.float64, device=DEV)
    U, dU = pot(q)
    for c in range(3):
        qq = q[c].cpu().numpy()
        assert abs(U[c].item() - orc.gp_regression_potential(x[obs], y[obs], qq[:12], qq[12], qq[13], qq[14])) < 1e-9 * abs(U[c].item())
    # gpvae_gp_potential (one call) against the composed path (gpvae_mvn_logp + glue), f64 and f32
    U2, dF2, dV2, dL2, dS2 = inf._gp_potential(inf.x[inf.obs_idx].contiguous(), inf.y[inf.obs_idx], None, q[:, :12], q[:, 12], q[:, 13], q[:, 14])
    assert rel(U, U2.cpu().numpy()) < 1e-12 and rel(dU, torch.cat([dF2, dV2[:, None], dL2[:, None], dS2[:, None]], 1).cpu().numpy()) < 1e-9
    inf32 = GPInference(exp_kernel2, 1, x, y, obs, args, seed=5, dtype=torch.float32)
    # (f drawn near the prior scale: a random f would make f^T K^-1 f ~ 1e6 and float32 meaningless)
    Lp = np.linalg.cholesky(orc.exp_kernel2(x[obs], x[obs], 1.0, 0.27, 1e-5))
    q32 = torch.tensor(np.concatenate([(Lp @ rng.standard_normal((12, 3))).T, np.tile([0.0, np.log(0.27), np.log(0.07)], (3, 1))], 1),
                       dtype=torch.float32, device=DEV)
    U32, dU32 = inf32.gp_regression(inf32.x[inf32.obs_idx], inf32.y[inf32.obs_idx])(q32)
    U64, dU64 = pot(q32.double())
    assert rel(U32, U64.cpu().numpy()) < 1e-3 and rel(dU32[:, :12], dU64[:, :12].cpu().numpy()) < 2e-2
    # full-grid model (inference.py:359-397): latent f on all n points, likelihood on the observed ones
    potf = inf.regression(y=inf.y, obs_idx=inf.obs_idx)
    qf = potf.init[None, :] + 0.01 * torch.randn(2, n + 3, dtype=torch.float64, device=DEV)
    Uf, dUf = potf(qf)
    h = 1e-5  # cond(K) ~ 1e6 at jitter 1e-5: the differences carry ~1e-4 of rounding noise
    for j in (3, int(obs[2]), n, n + 1, n + 2):  # an unobserved f, an observed f, the three hyper-parameters
        e = torch.zeros_like(qf); e[:, j] = h
        fd = (potf(qf + e)[0] - potf(qf - e)[0]) / (2 * h)
        assert torch.allclose(dUf[:, j], fd, rtol=1e-3, atol=1e-3), j
    prior = Predictive(inf.regression, num_samples=50)(3)["y_pred"]
    assert prior.shape == (50, n) and torch.isfinite(prior).all()
    post = inf.gp_run_mcmc(7, inf.x[inf.obs_idx], inf.y[inf.obs_idx], model=inf.gp_regression)
    assert post["f"].shape == (80, 12) and post["kernel_var"].shape == (80,)
    pp = Predictive(inf.gp_regression, post)(4, inf.x[inf.obs_idx])["y_pred"]
    assert pp.shape == (80, 12)
    # chain sharding: rank r of 2 runs chains r, r+2 and gets the very chains of the one-process run, bit for bit:
    # the potential sums in a fixed order (no float atomics), so a chain's numbers do not depend on the batch
    # it is evaluated in, and the trees of the batched NUTS are masked per chain
    inf.mcmc_args = {"num_warmup": 12, "num_samples": 8, "num_chains": 4, "thinning": 1, "max_tree_depth": 4}
    whole = inf.gp_run_mcmc(7, inf.x[inf.obs_idx], inf.y[inf.obs_idx], model=inf.gp_regression)["f"].reshape(4, 8, 12)
    again = inf.gp_run_mcmc(7, inf.x[inf.obs_idx], inf.y[inf.obs_idx], model=inf.gp_regression)["f"].reshape(4, 8, 12)
    assert torch.equal(whole, again)
    for r in range(2):
        monkeypatch.setattr(parallel, "dist_info", lambda r=r: (r, 2))
        part = inf.gp_run_mcmc(7, inf.x[inf.obs_idx], inf.y[inf.obs_idx], model=inf.gp_regression)["f"]
        assert inf.chain_ids == [r, r + 2] and part.shape == (16, 12)
        assert torch.equal(part.reshape(2, 8, 12), whole[[r, r + 2]])
    monkeypatch.undo()
    # VAE-prior model
    nz, h1, h0 = 6, 9, 12
    p = orc.init_params(rng, n, h0, h1, nz, np.float64)
    dec = [(50 * p["V1"], p["c1"]), (), (50 * p["V2"], p["c2"]), (), (50 * p["V3"], p["c3"])]
    vinf = VAEInference(None, dec, nz, 1, x, y, obs, dict(args, kernel="hmc"), seed=6, dtype=torch.float64)
    vp = vinf.regression(y=vinf.y, obs_idx=vinf.obs_idx)
    Z, u = rng.standard_normal((5, nz)), np.log(0.1) + 0.1 * rng.standard_normal(5)
    Uv, _ = vp(torch.tensor(np.concatenate([Z, u[:, None]], 1), device=DEV))
    pscaled = dict(p, V1=50 * p["V1"], V2=50 * p["V2"], V3=50 * p["V3"])
    assert rel(Uv, orc.vae_potential(pscaled, Z, u, y, obs)) < 1e-10
    pr = Predictive(vinf.regression, num_samples=30)(1)
    assert pr["y_pred"].shape == (30, n) and pr["z"].shape == (30, nz) and (pr["noise"] >= 0).all()
    smp = vinf.run_mcmc(3)
    assert smp["z"].shape == (80, nz) and smp["f"].shape == (80, n) and smp["noise"].shape == (80,)


def test_decoder_reference_shapes_and_nan_weights():
    """Decoder forward / VJP at the shapes and weight spread of the reference's six decoder files
    (src/test/decoders/*: (10,30),(30,35),(35,300) and (15,30),(30,50),(50,625), std 0.15-0.2; SURVEY.md
    §4).  The files themselves are refused by the safe loaders (tests/golden/make_decoder_fixtures.py), so
    the weights are synthetic.  One of the files is all-NaN (a diverged run): NaN weights must give NaN
    outputs, not a fault."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(70)
    for z, h1, h0, n in ((10, 30, 35, 300), (15, 30, 50, 625)):
        shapes = [(z, h1), (h1,), (h1, h0), (h0,), (h0, n), (n,)]
        w = [rng.normal(0, 0.18, s) for s in shapes]
        p = dict(zip(("V1", "c1", "V2", "c2", "V3", "c3"), w))
        Z = rng.standard_normal((16, z)); Z[0] = 0.0
        dF = rng.standard_normal((16, n))
        for tdt, tol in ((torch.float64, 1e-12), (torch.float32, 1e-5)):
            flat = dev(np.concatenate([a.reshape(-1) for a in w]), tdt)
            assert rel(ops.decoder_fwd(flat, dev(Z, tdt), n, h0, h1), orc.decoder(p, Z)) < tol
            assert rel(ops.decoder_vjp(flat, dev(Z, tdt), dev(dF, tdt), h0, h1), orc.decoder_vjp(p, Z, dF)) < tol
        bad = torch.full_like(flat, float("nan"))
        assert torch.isnan(ops.decoder_fwd(bad, dev(Z, torch.float32), n, h0, h1)).all()


def test_b3_planes_and_draw():
    """bf16 x 3: the planes sum to the float bit for bit (split and Philox producer), and the draw on
    them meets gpvae_draw's float32 bar against the float64 oracle — at least as close as the f32-MFMA
    kernel on the same inputs."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(9)
    X = torch.tensor(rng.standard_normal((2, 130, 70)) * 10 ** rng.uniform(-6, 6, (2, 130, 70)), dtype=torch.float32, device=DEV)
    P = ops.split_b3(X)
    assert P.shape == (2, 3, 130, 70) and torch.equal(P.float().sum(1), X)
    Pl = ops.split_b3(X[0, :70], lower=True)
    assert torch.equal(Pl.float().sum(0), torch.tril(X[0, :70]))
    E3 = ops.randn_b3((256, 384), DEV, seed=12345, stream_id=7)
    E = ops.randn((256, 384), torch.float32, DEV, seed=12345, stream_id=7)
    assert torch.equal(E3.float().sum(0), E)
    assert ops.draw_b3_supported(4096, 4096) and not ops.draw_b3_supported(300, 100) and not ops.draw_b3_supported(4096, 100)
    # 128-row tiles (fewer than 256 tiles of 256 rows, or nb not a multiple of 256) and 256-row tiles (the last case)
    for n, nb in ((128, 256), (384, 256), (1024, 512), (256, 128), (512, 384), (2048, 4096)):
        Lh = np.tril(rng.standard_normal((n, n))) * 0.1
        E2h = rng.standard_normal((nb, n))
        L = torch.tensor(Lh, dtype=torch.float32, device=DEV) + torch.triu(torch.full((n, n), 5.0, device=DEV), 1)
        E3 = ops.randn_b3((nb, n), DEV, seed=3, stream_id=n)
        Eh = E3.float().sum(0).cpu().numpy().astype(np.float64)
        L64 = np.tril(L.cpu().numpy().astype(np.float64))
        Fo = Eh @ L64.T
        E2 = torch.tensor(E2h, dtype=torch.float32, device=DEV)
        F, Y = ops.draw_b3(ops.split_b3(L, lower=True), E3, E2, 0.3, want_f=True)
        Ff = ops.draw(L, E3.float().sum(0))  # the f32-MFMA kernel on the same floats
        e_b3, e_f32 = rel(F, Fo), rel(Ff, Fo)
        assert e_b3 < 1e-5 and e_b3 < 2 * e_f32 + 1e-7, (n, nb, e_b3, e_f32)
        assert rel(Y, Fo + 0.3 * E2.cpu().numpy().astype(np.float64)) < 1e-5
        assert rel(ops.draw_b3(ops.split_b3(L, lower=True), E3), Fo) < 1e-5


def test_b3_draw_full_size_c3():
    """C3 (n = B = 4096): bf16 x 3 draw against the f32-MFMA draw and a float64 product on sampled rows."""
    from gp_vae_amd import ops
    n = B = 4096
    g = np.arange(0, 1, 1 / 64)
    u, v = np.meshgrid(g, g)
    x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1), dtype=torch.float32, device=DEV)
    K = ops.gram_se(x, x, 1.0, 0.2, 1e-3)
    assert int(ops.potrf_(K, zero_upper=False).item()) == 0
    E3 = ops.randn_b3((B, n), DEV, seed=1, stream_id=1)
    E2 = ops.randn((B, n), torch.float32, DEV, seed=1, stream_id=2)
    Y = ops.draw_b3(ops.split_b3(K, lower=True), E3, E2, 0.05)
    E = E3.float().sum(0)
    Yf = ops.draw(K, E, E2, 0.05)
    rows = torch.tensor(np.random.default_rng(0).integers(0, B, 48), device=DEV)
    want = (E[rows].double() @ torch.tril(K).double().T + 0.05 * E2[rows].double()).cpu().numpy()
    e_b3, e_f32 = rel(Y[rows], want), rel(Yf[rows], want)
    assert e_b3 < 1e-5 and e_b3 < 2 * e_f32 + 1e-7, (e_b3, e_f32)


def test_potrf_b3_factor_and_planes():
    """gpvae_potrf_b3: the float32 factor equals gpvae_potrf's to LAPACK-level backward error, and the planes
    it leaves are the factor bit for bit on and below the diagonal — and zero wherever the draw reads
    above it: draws from them equal draws from split_b3(tril(L)).  Non-PD matrix in the batch: info."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(5)
    assert ops.potrf_b3_supported(4096) and ops.potrf_b3_supported(512) and not ops.potrf_b3_supported(300) \
        and not ops.potrf_b3_supported(256)
    # 2304 / 4352: 9 / 17 outer blocks — the trailing updates delayed over groups of 2 / 4 blocks, the last group short
    for n, batch in ((512, 3), (1280, 9), (2304, 3), (4352, 3)):
        x = torch.tensor(rng.random((n, 2)), dtype=torch.float32, device=DEV)
        ls = torch.linspace(0.05, 0.6, batch, device=DEV)
        K = ops.gram_se(x, x, torch.ones(batch, device=DEV), ls, 1e-3).reshape(batch, n, n)
        K[1, 300, 300] = -1.0  # not positive definite from column 301
        A = K.clone()
        junk = torch.full((batch, 3, n, n), 7.0, dtype=torch.bfloat16, device=DEV)  # planes start as garbage
        info, L3 = ops.potrf_b3_(A, junk)
        assert info.cpu().tolist() == [0, 301] + [0] * (batch - 2)
        ref = K.clone()
        assert torch.equal(ops.potrf_(ref, zero_upper=False, nan_on_fail=False), info)
        for b in (0, 2, batch - 1):
            L = torch.tril(A[b]).double()
            Kb = K[b].double()
            assert (torch.linalg.norm(L @ L.T - Kb) / torch.linalg.norm(Kb)).item() < 2e-6
            # planes == factor bit for bit, except where a piece underflows bf16's normal range (entries < 1e-30)
            dif = (torch.tril(L3[b].float().sum(0)) - torch.tril(A[b])).abs()
            assert dif.max().item() < 1e-30 and torch.equal(dif > 0, (dif > 0) & (A[b].tril().abs() < 1e-28))
            E3 = ops.randn_b3((256, n), DEV, seed=b, stream_id=2)
            want = ops.draw_b3(ops.split_b3(A[b], lower=True), E3)
            assert torch.equal(ops.draw_b3(L3[b], E3), want)
            # the plain entry point (planes of the current group of panels in its workspace) does the same arithmetic
            assert torch.equal(A[b].tril(), ref[b].tril())
        # NaN-on-fail: every draw from a failed matrix's planes is NaN
        A2 = K.clone()
        info2, L32 = ops.potrf_b3_(A2, nan_on_fail=True)
        assert torch.isnan(ops.draw_b3(L32[1], ops.randn_b3((256, n), DEV, seed=1))).all()
        assert not torch.isnan(ops.draw_b3(L32[0], ops.randn_b3((256, n), DEV, seed=1))).any()


def test_potrf_share_chip_rows_kernel():
    """GPVAE_POTRF_SHARE_CHIP (the factor stream's calls in StepEngine): per outer block the panel launches cover the
    block's own rows and ONE potrf_rows_kernel launch the rows below it (n = 1024 x 48 matrices: the first two outer
    blocks take that form, the rest the plain one).  A hint, not a change of result: same info, the factor of the
    plain call to rounding (the solve's reciprocals come from another seed), planes == factor bit for bit, LAPACK-level
    backward error, on the three entry points of the planes route."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(11)
    n, batch = 1024, 48
    x = torch.tensor(rng.random((n, 2)), dtype=torch.float32, device=DEV)
    ls = torch.linspace(0.04, 3.0, batch, device=DEV)  # up to nearly rank-one matrices
    K = ops.gram_se(x, x, torch.ones(batch, device=DEV), ls, 1e-3).reshape(batch, n, n)
    K[5, 700, 700] = -1.0  # not positive definite from column 701
    want_info = [0] * batch
    want_info[5] = 701
    ref = K.clone()
    assert ops.potrf_(ref, zero_upper=False, nan_on_fail=False).cpu().tolist() == want_info
    A0 = K.clone()
    assert ops.potrf_(A0, zero_upper=False, nan_on_fail=False, share_chip=True).cpu().tolist() == want_info
    A1 = K.clone()
    info1, L3 = ops.potrf_b3_(A1, share_chip=True)
    A2 = K.clone()
    info2, L2, inv2 = ops.potrf_h2_(A2, share_chip=True)
    assert info1.cpu().tolist() == want_info and info2.cpu().tolist() == want_info
    ok = [b for b in range(batch) if b != 5]  # (the failed matrix holds NaN from column 701 on)
    assert torch.equal(A0[ok].tril(), A1[ok].tril())  # workspace planes and the caller's planes: the same arithmetic
    for b in (0, 17, 47):
        Kb = K[b].double()
        for A in (A1, A2):
            L = torch.tril(A[b]).double()
            assert (torch.linalg.norm(L @ L.T - Kb) / torch.linalg.norm(Kb)).item() < 2e-6
        d = (A1[b].tril() - ref[b].tril()).abs().max().item()
        assert d <= 4e-6 * ref[b].abs().max().item(), d
        dif = (torch.tril(L3[b].float().sum(0)) - torch.tril(A1[b])).abs()
        assert dif.max().item() < 1e-30
        Lh = (L2[b].float().sum(0) * inv2[b]).tril()
        assert rel(Lh, A2[b].tril().cpu().numpy()) < 1e-6


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-12), (torch.float32, 2e-7)])
def test_device_hyper_prior_samplers(tdt, tol):
    """gpvae_hyper_sample against the oracle's restatement (same Philox words, same Marsaglia-Tsang
    accept/reject): InverseGamma(1, .1) / LogNormal / HalfNormal of vae.py:157-159, the kriging model's
    InverseGamma(4, 1), the README's Beta(.1, 2); batching independence; and the engine trained with
    hyper_sampler='device' equals the engine fed the oracle's draws from the host."""
    from gp_vae_amd import ops, parallel
    from gp_vae_amd.vae import StepEngine, SVIState, init_params
    for kw_d, kw_o in (({}, {}), ({"ls_a": 4.0, "ls_b": 1.0}, {"ls_a": 4.0, "ls_b": 1.0}),
                       ({"ls_prior": ops.PRIOR_BETA, "ls_a": 0.1, "ls_b": 2.0}, {"ls_prior": "beta", "ls_a": 0.1, "ls_b": 2.0})):
        ls, var, sig = ops.hyper_sample(300, tdt, DEV, seed=77, stream_id=5, first_index=40, **kw_d)
        lo, vo, so = orc.hyper_sample(300, seed=77, stream_id=5, first_index=40, **kw_o)
        assert rel(ls, lo) < tol and rel(var, vo) < tol and rel(sig, so) < tol, kw_d
    a = ops.hyper_sample(8, tdt, DEV, seed=1, stream_id=2, first_index=5)
    b = ops.hyper_sample(20, tdt, DEV, seed=1, stream_id=2, first_index=0)
    assert all(torch.equal(u, v[5:13]) for u, v in zip(a, b))
    if tdt == torch.float64:
        n, B, dims = 96, 40, (12, 9, 4)
        x = np.arange(0, 1, 1 / n)
        keys = [(2, i) for i in range(5)] + [(3, 0), (3, 1)]  # two epoch keys: two sampler launches in a group
        e1 = StepEngine(torch.tensor(x), 1e-3, dims, B, 1e-3, dtype=tdt, seed=3, group=4, hyper_sampler="device")

        def from_oracle(rng):  # host sampler that replays the device stream: key rides in through a closure
            k = from_oracle.keys.pop(0)
            l, v, s = orc.hyper_sample(1, seed=3, stream_id=parallel.philox_stream_id(0, (k[0], 0), 5), first_index=k[-1])
            return l[0], v[0], s[0]
        from_oracle.keys = list(keys)
        e2 = StepEngine(torch.tensor(x), 1e-3, dims, B, 1e-3, dtype=tdt, seed=3, group=4, hyper_sampler=from_oracle)
        s1, s2 = (SVIState(init_params(n, *dims, 7, tdt, DEV) * 5) for _ in range(2))
        l1, l2 = e1.run(keys, s1).item(), e2.run(keys, s2).item()
        assert abs(l1 - l2) < 1e-9 * abs(l2) and rel(s1.params, s2.params.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_vae_potential_one_launch(tdt, tol):
    """gpvae_vae_potential (gathered output columns, one launch) against the oracle's potential and against the
    composed path (decoder forward + glue + decoder VJP): reference decoder shapes, ragged chain counts,
    many observations; a decoder too wide for LDS falls back to the composed path."""
    from gp_vae_amd import ops
    from gp_vae_amd.inference import VAEInference
    rng = np.random.default_rng(8)
    for chains, n, h0, h1, z, n_obs in ((1024, 300, 35, 30, 10, 10), (37, 625, 50, 30, 15, 60), (5, 64, 3, 2, 1, 64)):
        p = {k: 15 * v for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
        lst = [(p["V1"], p["c1"]), (), (p["V2"], p["c2"]), (), (p["V3"], p["c3"])]
        obs = np.sort(rng.permutation(n)[:n_obs])
        y = np.sin(np.arange(n) / 30.0)
        inf = VAEInference(None, lst, z, 1, np.arange(n) / n, y, obs, {}, dtype=tdt)
        assert ops.vae_potential_supported(tdt, h0, h1, z, n_obs)
        Z, u = rng.standard_normal((chains, z)), np.log(0.1) + 0.3 * rng.standard_normal(chains)
        U, dz, du = inf.potential_and_grad(Z, u)                                   # fused kernel
        U2, dz2, du2 = inf._potential(Z, u, inf.y[inf.obs_idx], inf.obs_idx)       # composed path
        assert rel(U, orc.vae_potential(p, Z, u, y, obs)) < tol
        assert rel(U, U2.cpu().numpy()) < tol and rel(dz, dz2.cpu().numpy()) < tol and rel(du, du2.cpu().numpy()) < tol
        pot = inf.regression(y=inf.y, obs_idx=inf.obs_idx)
        q = torch.tensor(np.concatenate([Z, u[:, None]], 1), dtype=tdt, device=DEV)
        Uq, dUq = pot(q)
        assert torch.equal(Uq, U) and torch.equal(dUq[:, :z], dz) and torch.equal(dUq[:, z], du)
    assert not ops.vae_potential_supported(tdt, 512, 256, 64, 10)
    n, h0, h1, z = 40, 512, 256, 64
    p = orc.init_params(rng, n, h0, h1, z, np.float64)
    lst = [(p["V1"], p["c1"]), (), (p["V2"], p["c2"]), (), (p["V3"], p["c3"])]
    inf = VAEInference(None, lst, z, 1, np.arange(n) / n, np.zeros(n), [1, 5, 9], {}, dtype=tdt)
    Z, u = rng.standard_normal((7, z)), np.full(7, np.log(0.1))
    assert rel(inf.potential_and_grad(Z, u)[0], orc.vae_potential(p, Z, u, np.zeros(n), np.array([1, 5, 9]))) < max(tol, 1e-5)


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-6), (torch.float32, 2e-3)])
def test_nuts_device_transition_matches_the_torch_sampler(tdt, tol):
    """gpvae_nuts_op (the NUTS transition as HIP kernels, one wave per chain) against gp_vae_amd.nuts.nuts (the
    same algorithm as torch operations) on the same normals and uniforms: the same trees, proposals and
    adaptation — positions agree to rounding over warm-up and sampling (few iterations: trajectories are
    chaotic, a rounding difference may eventually flip a tree decision), ragged D (lanes stride the coordinates),
    a potential with divergences in it, and a chain's numbers do not depend on its batch."""
    from gp_vae_amd.nuts import nuts, nuts_device, uniforms_per_iteration
    rng = np.random.default_rng(4)
    for Cn, D, depth in ((7, 3, 4), (5, 70, 5), (33, 130, 3)):
        mu = torch.tensor(rng.standard_normal(D), dtype=tdt, device=DEV)
        sd = torch.tensor(np.exp(rng.uniform(-1.5, 1.0, D)), dtype=tdt, device=DEV)
        calls = [0]

        def pot(q):
            calls[0] += 1
            r = (q - mu) / sd
            U = 0.5 * (r * r).sum(1) + 0.05 * (r ** 4).sum(1)   # not Gaussian: energy errors, some divergence early
            return U, r / sd + 0.2 * r ** 3 / sd
        K = uniforms_per_iteration(depth)

        def noise(rows):
            def fn(it):
                g = torch.Generator().manual_seed(77 + it)
                z = torch.randn((Cn, D), dtype=torch.float64, generator=g).to(DEV, tdt)
                u = torch.rand((Cn, K), dtype=torch.float64, generator=g).to(DEV, tdt)
                return z[rows].contiguous(), u[rows].contiguous()
            return fn
        q0 = torch.tensor(rng.standard_normal((Cn, D)), dtype=tdt, device=DEV)
        allr = torch.arange(Cn, device=DEV)
        its = (6, 4) if tdt == torch.float32 else (25, 10)
        a, ia = nuts(pot, q0, its[0], its[1], max_tree_depth=depth, step_size=0.3, noise_fn=noise(allr))
        b, ib = nuts_device(pot, q0, its[0], its[1], max_tree_depth=depth, step_size=0.3, noise_fn=noise(allr))
        assert a.shape == b.shape == (its[1], Cn, D)
        err = ((a - b).abs().max() / a.abs().max()).item()
        assert err < tol, (Cn, D, depth, err)
        assert torch.allclose(ia["step_size"], ib["step_size"], rtol=max(tol, 1e-6))
        assert torch.allclose(ia["leapfrogs_per_iteration"], ib["leapfrogs_per_iteration"]) and ia["divergent"] == ib["divergent"]
        sub = torch.tensor([1, Cn - 1], device=DEV)
        c, _ = nuts_device(pot, q0[sub], its[0], its[1], max_tree_depth=depth, step_size=0.3, noise_fn=noise(sub))
        # (not bit for bit here: torch's own row sums inside this test potential change their order with the batch
        # shape; with the library's potentials the equality is exact — test_regression_seams_and_chain_sharding)
        assert ((c - b[:, sub]).abs().max() / b.abs().max()).item() < tol


def test_h2_planes_and_draw():
    """fp16 x 2 (opt-in): the two planes hold the scaled float to 2^-22, nothing overflows at the stated bound,
    and the draw from them meets gpvae_draw's float32 bar against the float64 oracle — error of the same size
    as the f32-MFMA and bf16 x 3 kernels on the same inputs — over wide dynamic ranges of L and both tile sizes."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(19)
    X = torch.tensor(rng.standard_normal((2, 130, 72)) * 10 ** rng.uniform(-4, 1, (2, 130, 72)), dtype=torch.float32, device=DEV)
    bound = X.abs().amax(dim=(1, 2))
    P, inv = ops.split_h2(X, bound)
    assert P.shape == (2, 2, 130, 72) and torch.isfinite(P.float()).all()
    back = P.float().sum(1) * inv[:, None, None]
    err = (back - X).abs()
    assert (err <= 2.0 ** -21 * X.abs() + 2.0 ** -38 * bound[:, None, None]).all()
    assert (P[:, 0].float().abs().amax(dim=(1, 2)) <= 2.0 ** 14).all() and (P[:, 0].float().abs().amax(dim=(1, 2)) > 2.0 ** 12).all()
    E2h = ops.randn_h2((256, 384), DEV, seed=12345, stream_id=7)
    E = ops.randn((256, 384), torch.float32, DEV, seed=12345, stream_id=7)
    assert ((E2h.float().sum(0) / 4096.0 - E).abs() <= 2.0 ** -21 * E.abs() + 1e-12).all()
    for n, nb, spread in ((128, 256, 0), (384, 256, 3), (1024, 512, 6), (512, 384, 2), (2048, 4096, 4)):
        Lh = np.tril(rng.standard_normal((n, n))) * 10 ** rng.uniform(-spread, 0, (n, n)) * 3.0
        L = torch.tensor(Lh, dtype=torch.float32, device=DEV) + torch.triu(torch.full((n, n), 5.0, device=DEV), 1)
        L2, inv = ops.split_h2(L, torch.tril(L).abs().max(), lower=True)
        E2h = ops.randn_h2((nb, n), DEV, seed=3, stream_id=n)
        E = ops.randn((nb, n), torch.float32, DEV, seed=3, stream_id=n)
        E2 = ops.randn((nb, n), torch.float32, DEV, seed=4, stream_id=n)
        Fo = E.double() @ torch.tril(L).double().T
        F, Y = ops.draw_h2(L2, inv, E2h, E2, 0.3, want_f=True)
        Ff = ops.draw(L, E)                                            # f32 MFMA
        Fb = ops.draw_b3(ops.split_b3(L, lower=True), ops.randn_b3((nb, n), DEV, seed=3, stream_id=n))
        e_h2, e_f32, e_b3 = rel(F, Fo.cpu().numpy()), rel(Ff, Fo.cpu().numpy()), rel(Fb, Fo.cpu().numpy())
        assert e_h2 < 1e-5 and e_h2 < 3 * e_f32 + 2e-7, (n, nb, e_h2, e_f32, e_b3)
        assert rel(Y, (Fo + 0.3 * E2.double()).cpu().numpy()) < 1e-5
        print("h2 draw n=%d nb=%d: rel err fp16x2 %.2e  f32 MFMA %.2e  bf16x3 %.2e" % (n, nb, e_h2, e_f32, e_b3))


def test_potrf_h2_factor_and_planes():
    """gpvae_potrf_h2: the float32 factor it leaves has a float32-level backward error (within 4x of gpvae_potrf's
    on the same matrix, near-singular K = var 11^T + jitter I included), its planes are the factor times 2^e to
    2^-22 with zeros wherever the draw reads above the diagonal, and draws from them meet the draw's float32 bar.
    Non-PD matrix in the batch: same info as gpvae_potrf."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(6)
    for n, batch in ((512, 3), (1280, 9), (2048, 3)):
        x = torch.tensor(rng.random((n, 2)), dtype=torch.float32, device=DEV)
        ls = torch.linspace(0.05, 0.6, batch, device=DEV)
        var = torch.tensor(10 ** rng.uniform(-2, 2, batch), dtype=torch.float32, device=DEV)
        K = ops.gram_se(x, x, var, ls, 1e-3).reshape(batch, n, n)
        K[batch - 1] = var[batch - 1] * (1.0 + 1e-4 * torch.eye(n, device=DEV))  # near rank one: ls -> infinity
        K[1, 300, 300] = -1.0  # not positive definite from column 301
        A = K.clone()
        junk = torch.full((batch, 2, n, n), 7.0, dtype=torch.float16, device=DEV)
        info, L2, inv = ops.potrf_h2_(A, junk)
        assert info.cpu().tolist() == [0, 301] + [0] * (batch - 2)
        ref = K.clone()
        assert torch.equal(ops.potrf_(ref, zero_upper=False, nan_on_fail=False), info)
        for b in (0, 2, batch - 1):
            L, Lr, Kb = torch.tril(A[b]).double(), torch.tril(ref[b]).double(), K[b].double()
            top = Kb.diagonal().max().item()
            e_h2, e_ref = ((L @ L.T - Kb).abs().max() / top).item(), ((Lr @ Lr.T - Kb).abs().max() / top).item()
            print("potrf_h2 n=%d b=%d: max|LL^T - K| / max K_ii  fp16x2 %.2e  gpvae_potrf %.2e" % (n, b, e_h2, e_ref))
            assert e_h2 < 2e-6 and e_h2 < 4 * e_ref + 5e-8, (n, b, e_h2, e_ref)
            s = 1.0 / inv[b].item()
            assert 2.0 ** 13 < math.sqrt(top) * s <= 2.0 ** 14 * (1 + 1e-6)
            back = L2[b].float().sum(0).double() / s
            lim = 2.0 ** -21 * L.abs() + 2.0 ** -37 * math.sqrt(top)
            assert ((torch.tril(back) - L).abs() <= lim).all()
            E2h = ops.randn_h2((256, n), DEV, seed=b, stream_id=2)
            want = ops.draw_h2(*ops.split_h2(A[b], math.sqrt(top), lower=True), E2h)
            assert torch.equal(ops.draw_h2(L2[b], inv[b:b + 1], E2h), want)
        A2 = K.clone()
        info2, L22, inv2 = ops.potrf_h2_(A2, nan_on_fail=True)
        assert torch.isnan(ops.draw_h2(L22[1], inv2[1:2], ops.randn_h2((256, n), DEV, seed=1))).all()
        assert not torch.isnan(ops.draw_h2(L22[0], inv2[0:1], ops.randn_h2((256, n), DEV, seed=1))).any()


@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float64, np.float64, 1e-14), (torch.float32, np.float32, 2e-7)])
def test_poisson_exp_matches_oracle(tdt, ndt, tol):
    """gpvae_poisson_exp (rate = exp(f), y ~ Poisson(rate): gp.py:249-250): rates to rounding, counts BIT-EXACT
    against the oracle's restatement of the sampler on the same rates, edge values, batching independence."""
    from gp_vae_amd import ops
    rng = np.random.default_rng(23)
    f = np.concatenate([rng.uniform(-7, 2.25, 3000), rng.uniform(2.35, 12, 3000), [-np.inf, 0.0, 25.0, 1000.0, np.nan]]).astype(ndt)
    ft = torch.tensor(f, device=DEV)
    rate, y = ops.poisson_exp(ft, seed=77, stream_id=5)
    rt, yt = rate.cpu().numpy(), y.cpu().numpy()
    want_rate, _ = orc.poisson_exp(f, 77, 5)
    fin = np.isfinite(want_rate)
    assert np.allclose(rt[fin], want_rate[fin], rtol=tol, atol=0) and np.isinf(rt[-2]) and np.isnan(rt[-1]) and rt[-5] == 0
    _, want_y = orc.poisson_exp(f, 77, 5, rate=rt)   # the sampler on the device's own rates: integer parity
    assert y.dtype == torch.int32 and np.array_equal(yt, want_y)
    assert yt[-1] == -1 and yt[-2] == 2 ** 31 - 1 and yt[-5] == 0
    # both branches are exercised and the counts follow their rates
    small, large = rt[:3000], rt[3000:6000]
    assert small.max() < 10 < large.min() and abs((yt[:3000] - small).sum()) < 5 * math.sqrt(small.sum())
    assert abs((yt[3000:6000] - large).sum()) < 5 * math.sqrt(large.sum())
    # element i depends on first_index + i only
    _, y2 = ops.poisson_exp(ft[4000:], seed=77, stream_id=5, first_index=4000)
    assert torch.equal(y2, y[4000:])
    assert ops.poisson_exp(ft[:0], seed=1)[1].numel() == 0


@pytest.mark.parametrize("tdt,ndt,tol", [(torch.float64, np.float64, 1e-9), (torch.float32, np.float32, 2e-4)])
def test_poigp_sample_against_oracle(tdt, ndt, tol):
    """PoiGP.sample (gp.py:197-250) behind Predictive: with the hyper-parameters given, f / rate / y against
    oracle.poigp_sample on the same sample points and normals (1-D and 2-D grids); with them left None, the
    `length` / `var` sites, one aggregated kernel per sample, counts that follow their rates."""
    from gp_vae_amd import ops, gp as gpm
    for d, n, m, B, ls in ((1, 24, 6, 64, 0.05), (2, 5, 4, 32, 0.15)):
        grid = torch.arange(n, dtype=tdt, device=DEV) / n
        prior = gpm.PoiGP(kernel=gpm.exp_kernel2, jitter=1e-4, d=d)
        key = gpm.PhiloxKey(9, base=0x30)
        out = gpm.Predictive(prior.sample, num_samples=B)(key, m, grid, ls=ls, var=0.7, seed=4)
        cells = n ** d
        assert out["f"].shape == (B, cells) and out["y"].dtype == torch.int32 and int(prior.last_info.item()) == 0
        pts = gpm._agg_points(4, d, m, grid).cpu().numpy()
        eps = ops.randn((B, cells), tdt, DEV, seed=9, stream_id=0x31).cpu().numpy().astype(np.float64)
        f, _, _ = orc.poigp_sample(pts.astype(np.float64), 0.7, ls, eps, 1e-4)
        assert rel(out["f"], f) < tol, (d, rel(out["f"], f))
        rt = out["rate"].cpu().numpy()
        assert np.allclose(rt, np.exp(out["f"].cpu().numpy().astype(np.float64)), rtol=10 * np.finfo(ndt).eps)
        _, want_y = orc.poisson_exp(out["f"].cpu().numpy(), 9, 0x34, rate=rt)
        assert np.array_equal(out["y"].cpu().numpy(), want_y)
        # observed counts are passed through (numpyro's obs=)
        obs = torch.ones(cells, dtype=torch.int32, device=DEV)
        assert torch.equal(prior.sample(m, grid, y=obs, ls=ls, var=0.7, rng_key=key, num_samples=2)["y"], obs.expand(2, cells))
        # hyper-parameters left None: per-sample sites and kernels.  The aggregated kernel carries no diagonal
        # jitter (gp.py:119-127 adds it to every entry), so the long length scales of InverseGamma(1, 0.1) give
        # matrices that are not positive definite: NaN draws as in the reference, counts -1
        out2 = gpm.Predictive(prior.sample, num_samples=B)(key, m, grid, seed=4)
        assert out2["length"].shape == (B,) and out2["var"].shape == (B,)
        good = prior.last_info == 0
        assert good.sum().item() >= B // 8
        assert torch.isnan(out2["f"][~good]).all() and (out2["y"][~good] == -1).all()
        i = int(torch.where(good, out2["length"], torch.full_like(out2["length"], 1e9)).argmin().item())  # best conditioned
        f_i, _, _ = orc.poigp_sample(pts.astype(np.float64), float(out2["var"][i]), float(out2["length"][i]), eps[i:i + 1], 1e-4)
        assert rel(out2["f"][i:i + 1], f_i) < 50 * tol
        tot, lam = out2["y"][good].double().sum().item(), out2["rate"][good].double().sum().item()
        assert (out2["y"][good] >= 0).all() and abs(tot - lam) < 6 * math.sqrt(lam)


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_potentials_with_repeated_and_out_of_range_indices(tdt, tol):
    """obs_idx / lik_idx may name a point twice (the reference's f[obs_idx] then counts it twice: inference.py:251-254,
    392-396) — potential and gradient of gpvae_gp_potential and gpvae_vae_potential against the oracle with the
    repeated observation written out, the same bits on every call; an index outside [0, n) is rejected on the host."""
    from gp_vae_amd import ops
    from gp_vae_amd._lib import GpvaeError
    from gp_vae_amd.gp import exp_kernel2
    from gp_vae_amd.inference import GPInference, VAEInference
    rng = np.random.default_rng(4)
    # ---- latent-f GP model on the whole grid, likelihood on f[idx] with repeats
    n, C_ = 40, 3
    x = np.arange(n) / n
    idx = np.array([3, 17, 3, 25, 39, 17, 3, 0])
    yv = np.sin(5 * x[idx]) + 0.1 * rng.standard_normal(len(idx))
    uv, ul, us = rng.normal(0, 0.1, C_), np.log(rng.uniform(0.15, 0.4, C_)), np.log(rng.uniform(0.05, 0.2, C_))
    jit = 1e-5 if tdt == torch.float64 else 1e-2  # float32: a nugget that keeps f^T K^-1 f within the 2e-4 bar
    f = np.stack([np.linalg.cholesky(orc.exp_kernel2(x, x, np.exp(uv[c]), np.exp(ul[c]), jit)) @ rng.standard_normal(n) for c in range(C_)])
    q = dev(np.concatenate([f, uv[:, None], ul[:, None], us[:, None]], 1), tdt)
    xd, yd, idd = dev(x, tdt), dev(yv, tdt), torch.tensor(idx, device=DEV)
    U, dU, info = ops.gp_potential(xd, q, yd, idd, jit)
    U2, dU2, _ = ops.gp_potential(xd, q, yd, idd, jit)
    assert torch.equal(U, U2) and torch.equal(dU, dU2) and int(info.abs().sum()) == 0

    def pot(c, ff, a, b, g):  # oracle: MVN prior on all n points + one likelihood term PER OCCURRENCE in idx
        full = orc.gp_regression_potential(x, ff, ff, a, b, g, jit)  # likelihood residuals zero: its constant terms only
        r = (yv - ff[idx]) / np.exp(g)
        m = len(idx)
        return full + 0.5 * r @ r + (m - n) * g + 0.5 * (m - n) * np.log(2 * np.pi)
    h = 1e-5
    for c in range(C_):
        u0 = pot(c, f[c], uv[c], ul[c], us[c])
        assert abs(U[c].item() - u0) < tol * abs(u0)
        if tdt == torch.float64:
            for k in (3, 17, 5):  # thrice, twice and never observed
                e = np.zeros(n); e[k] = h
                fd = (pot(c, f[c] + e, uv[c], ul[c], us[c]) - pot(c, f[c] - e, uv[c], ul[c], us[c])) / (2 * h)
                assert abs(dU[c, k].item() - fd) < 1e-4 * max(1.0, abs(fd)), (c, k)
            fd = (pot(c, f[c], uv[c], ul[c], us[c] + h) - pot(c, f[c], uv[c], ul[c], us[c] - h)) / (2 * h)
            assert abs(dU[c, n + 2].item() - fd) < 1e-4 * max(1.0, abs(fd))
    # the torch composition of the same model (GPInference._gp_potential) agrees, repeats included
    y_full = np.full(n, np.nan); y_full[idx] = yv
    inf = GPInference(exp_kernel2, 1, x, y_full, idx, {}, dtype=tdt)
    if tdt == torch.float64:  # (the composition's nugget is the reference's fixed 1e-5)
        Ug, dF, dV, dL, dS = inf._gp_potential(xd, yd, idd, q[:, :n], q[:, n], q[:, n + 1], q[:, n + 2])
        assert rel(Ug, U.cpu().numpy()) < tol and rel(dF, dU[:, :n].cpu().numpy()) < 10 * tol
    for bad in ([3, 40], [-1, 2]):
        with pytest.raises(GpvaeError):
            ops.gp_potential(xd, q, yd[:2].contiguous(), torch.tensor(bad, device=DEV))
        with pytest.raises(GpvaeError):
            GPInference(exp_kernel2, 1, x, y_full, bad, {}, dtype=tdt)
    # ---- decoder-prior model: gathered columns, repeats and range
    nd, dims, chains = 30, (9, 7, 4), 5
    p = orc.init_params(np.random.default_rng(1), nd, *dims, np.float64)
    p = {k: 20 * v for k, v in p.items()}
    dec_flat = dev(np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]), tdt)
    oi = np.array([4, 11, 4, 29, 0, 11])
    yo = rng.standard_normal(len(oi))
    zz, lsg = rng.standard_normal((chains, dims[2])), np.log(rng.uniform(0.05, 0.3, chains))
    qd = dev(np.concatenate([zz, lsg[:, None]], 1), tdt)
    assert ops.vae_potential_supported(tdt, dims[0], dims[1], dims[2], len(oi))
    Uv, dUv = ops.vae_potential(dec_flat, qd, dev(yo, tdt), torch.tensor(oi, device=DEV), nd, dims[0], dims[1])

    def vpot(zc, lc):  # the oracle takes y as a full vector: evaluate one occurrence at a time and add the repeats by hand
        sig = np.exp(lc)
        fdec = orc.decoder(p, zc[None])[0]
        r = fdec[oi] - yo
        loglik = np.sum(-0.5 * (r / sig) ** 2 - lc - 0.5 * np.log(2 * np.pi))
        logpz = np.sum(-0.5 * zc ** 2 - 0.5 * np.log(2 * np.pi))
        logps = np.log(2.0) - 0.5 * np.log(2 * np.pi) - np.log(0.1) - 0.5 * (sig / 0.1) ** 2 + lc
        return -(loglik + logpz + logps)
    for c in range(chains):
        u0 = vpot(zz[c], lsg[c])
        assert abs(Uv[c].item() - u0) < tol * max(1.0, abs(u0))
        if tdt == torch.float64:
            e = np.zeros(dims[2]); e[1] = h
            fd = (vpot(zz[c] + e, lsg[c]) - vpot(zz[c] - e, lsg[c])) / (2 * h)
            assert abs(dUv[c, 1].item() - fd) < 1e-4 * max(1.0, abs(fd))
    # without repeats the hand-written potential above IS oracle.vae_potential (pins the restatement used here)
    uniq = np.array([4, 11, 29])
    y_u = np.zeros(nd); y_u[uniq] = yo[[0, 1, 3]]
    oi_keep, yo_keep = oi, yo
    oi, yo = uniq, yo_keep[[0, 1, 3]]
    assert abs(np.ravel(orc.vae_potential(p, zz[:1], lsg[:1], y_u, uniq))[0] - vpot(zz[0], lsg[0])) < 1e-9 * abs(vpot(zz[0], lsg[0]))
    oi, yo = oi_keep, yo_keep
    dec_list = [(dev(p["V1"], tdt), dev(p["c1"], tdt)), (), (dev(p["V2"], tdt), dev(p["c2"], tdt)), (), (dev(p["V3"], tdt), dev(p["c3"], tdt))]
    yfull = np.full(nd, np.nan); yfull[oi] = yo  # last occurrence wins in this vector: only used for shapes below
    for bad in ([4, 30], [-2, 3]):
        with pytest.raises(GpvaeError):
            ops.vae_potential(dec_flat, qd, dev(yo[:2], tdt), torch.tensor(bad, device=DEV), nd, dims[0], dims[1])
        with pytest.raises(GpvaeError):
            VAEInference(None, dec_list, dims[2], 1, np.arange(nd) / nd, yfull, bad, {}, dtype=tdt)
    # the composed path (decoder forward + glue + VJP) handles repeats the same way
    inf = VAEInference(None, dec_list, dims[2], 1, np.arange(nd) / nd, yfull, oi, {}, dtype=tdt)
    Uc, dz, du = inf._potential(qd[:, :dims[2]], qd[:, dims[2]], dev(yo, tdt), torch.tensor(oi, device=DEV))
    assert rel(Uc, Uv.cpu().numpy()) < 10 * tol and rel(dz, dUv[:, :dims[2]].cpu().numpy()) < 50 * tol


@pytest.mark.parametrize("tdt", [torch.float64, torch.float32])
def test_fused_nuts_iteration_matches_the_lockstep_driver(tdt):
    """gpvae_nuts_vae_iteration (one launch per NUTS iteration: every chain runs its own tree to its end, the
    decoder-prior potential inside the wave) against the lock-step driver (gpvae_nuts_op around
    gpvae_vae_potential) on the same normals and uniforms: same trees (leapfrog counts, divergences), same
    adaptation, positions equal to rounding; a chain does not depend on the batch it runs in (ragged chain count:
    the last workgroup is partly empty); a decoder too wide for LDS falls back to the lock-step driver."""
    from gp_vae_amd import _lib
    from gp_vae_amd.inference import VAEInference
    rng = np.random.default_rng(8)
    n, dims = 300, (35, 30, 10)  # the reference's 1-D decoder (toy_examples.py:615-630)
    p = {k: 18 * v for k, v in orc.init_params(np.random.default_rng(2), n, *dims, np.float64).items()}
    dec = [(dev(p["V1"], tdt), dev(p["c1"], tdt)), (), (dev(p["V2"], tdt), dev(p["c2"], tdt)), (), (dev(p["V3"], tdt), dev(p["c3"], tdt))]
    x = np.arange(n) / n
    obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]  # toy_examples.py:199
    y = np.full(n, np.nan)
    y[obs] = orc.decoder(p, rng.standard_normal((1, dims[2])))[0][obs] + 0.05 * rng.standard_normal(len(obs))
    its = (30, 12)
    res = {}
    for driver in ("iteration", "lockstep"):
        args = {"num_warmup": its[0], "num_samples": its[1], "num_chains": 13, "thinning": 1, "max_tree_depth": 6,
                "nuts_driver": driver}
        inf = VAEInference(None, dec, dims[2], 1, x, y, obs, args, seed=3, dtype=tdt)
        pot = inf.regression(y=inf.y, obs_idx=inf.obs_idx)
        assert hasattr(pot, "nuts_iteration")
        smp, info = inf._mcmc(11, pot)
        res[driver] = (smp, info)
    a, b = res["iteration"], res["lockstep"]
    assert a[0].shape == b[0].shape == (13 * its[1], dims[2] + 1) and torch.isfinite(a[0]).all()
    assert torch.equal(a[1]["leapfrogs_per_iteration"], b[1]["leapfrogs_per_iteration"]) and a[1]["divergent"] == b[1]["divergent"]
    assert a[1]["leapfrogs_per_iteration"].max().item() > 3  # real trees, not one-step trajectories
    # bit for bit: the wave-level potential inside the fused kernel evaluates gpvae_vae_potential's expressions in its
    # order (the scalar part is one shared function compiled without fused multiply-adds), and every phase is the
    # same device function in both drivers — anything less would not survive the warm-up's adaptation, which
    # amplifies a last-bit difference of one acceptance statistic about 5x per iteration
    assert torch.equal(a[0], b[0]) and torch.equal(a[1]["step_size"], b[1]["step_size"])
    assert torch.equal(a[1]["inverse_mass"], b[1]["inverse_mass"])
    lib = _lib.load()
    assert lib.gpvae_nuts_vae_supported(0, 35, 30, 10, 10) == 1 and lib.gpvae_nuts_vae_supported(1, 2000, 30, 10, 10) == 0
    wide = orc.init_params(np.random.default_rng(2), 40, 600, 30, 4, np.float64)  # h0 = 600: 18 k floats of V2 alone
    decw = [(dev(wide["V1"], tdt), dev(wide["c1"], tdt)), (), (dev(wide["V2"], tdt), dev(wide["c2"], tdt)), (), (dev(wide["V3"], tdt), dev(wide["c3"], tdt))]
    yw = np.full(40, np.nan); yw[[3, 9]] = 0.1
    infw = VAEInference(None, decw, 4, 1, np.arange(40) / 40, yw, [3, 9], {"num_warmup": 2, "num_samples": 2, "num_chains": 3}, dtype=tdt)
    assert not hasattr(infw.regression(y=infw.y, obs_idx=infw.obs_idx), "nuts_iteration")
    out = infw.run_mcmc(1)
    assert out["z"].shape == (6, 4) and torch.isfinite(out["z"]).all()


def test_whole_run_nuts_kernel_matches_the_per_iteration_driver():
    """gpvae_nuts_vae_run (warm-up, adaptation and sampling of every chain in ONE launch, noise drawn in place from
    the Philox stream) against the per-iteration driver: without warm-up the chains are the same chains (same
    normals, same uniforms, same transitions); with a short warm-up they agree to the rounding of the adaptation
    arithmetic; over a long run the two are two runs of one sampler (posterior moments, acceptance, step sizes)."""
    from gp_vae_amd.inference import VAEInference
    rng = np.random.default_rng(8)
    n, dims = 300, (35, 30, 10)
    p = {k: 18 * v for k, v in orc.init_params(np.random.default_rng(2), n, *dims, np.float64).items()}
    x = np.arange(n) / n
    obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]
    y = np.full(n, np.nan)
    y[obs] = orc.decoder(p, rng.standard_normal((1, dims[2])))[0][obs] + 0.05 * rng.standard_normal(len(obs))

    def run(tdt, driver, warm, ns, chains, thin=1, depth=6):
        dec = [(dev(p["V1"], tdt), dev(p["c1"], tdt)), (), (dev(p["V2"], tdt), dev(p["c2"], tdt)), (), (dev(p["V3"], tdt), dev(p["c3"], tdt))]
        args = {"num_warmup": warm, "num_samples": ns, "num_chains": chains, "thinning": thin, "max_tree_depth": depth,
                "nuts_driver": driver}
        inf = VAEInference(None, dec, dims[2], 1, x, y, obs, args, seed=3, dtype=tdt)
        pot = inf.regression(y=inf.y, obs_idx=inf.obs_idx)
        assert hasattr(pot, "nuts_run")
        smp, info = inf._mcmc(11, pot)
        return smp, info
    for tdt, tol in ((torch.float64, 1e-12), (torch.float32, 1e-5)):
        a, ia = run(tdt, "device", 0, 7, 13, thin=2)
        b, ib = run(tdt, "iteration", 0, 7, 13, thin=2)
        assert a.shape == b.shape == (13 * 4, dims[2] + 1)
        # (means over the sampling iterations: torch divides by multiplying with the reciprocal, the kernel divides)
        assert torch.allclose(ia["leapfrogs_per_iteration"], ib["leapfrogs_per_iteration"], rtol=1e-6) and ia["divergent"] == ib["divergent"]
        assert ((a - b).abs().max() / b.abs().max()).item() < tol, tdt
        assert torch.allclose(ia["accept_prob"], ib["accept_prob"], rtol=1e-5)
    a, ia = run(torch.float64, "device", 8, 4, 13)
    b, ib = run(torch.float64, "iteration", 8, 4, 13)
    assert ((a - b).abs().max() / b.abs().max()).item() < 1e-6
    assert torch.allclose(ia["step_size"], ib["step_size"], rtol=1e-9) and torch.allclose(ia["inverse_mass"], ib["inverse_mass"], rtol=1e-9)
    # a long run with mass-matrix windows: the same sampler statistically
    a, ia = run(torch.float32, "device", 150, 100, 256, depth=8)
    b, ib = run(torch.float32, "iteration", 150, 100, 256, depth=8)
    assert torch.isfinite(a).all() and a.shape == b.shape == (256 * 100, dims[2] + 1)
    sd = b.std(0)
    assert ((a.mean(0) - b.mean(0)).abs() / sd).max().item() < 0.15
    assert ((a.std(0) / sd) - 1).abs().max().item() < 0.15
    assert abs(ia["accept_prob"].mean().item() - ib["accept_prob"].mean().item()) < 0.03 and 0.6 < ia["accept_prob"].mean().item() < 0.95
    assert abs(ia["step_size"].log().mean().item() - ib["step_size"].log().mean().item()) < 0.1
    assert abs(ia["inverse_mass"].log().mean().item() - ib["inverse_mass"].log().mean().item()) < 0.2
