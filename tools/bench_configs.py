"""Measurements at the BASELINE.json configs that are not the bench line.
One JSON object per config on stdout; results are quoted in DESIGN.md §5 and kept in
profiles/r01_configs.json.   usage: python tools/bench_configs.py [c2] [c3s] [c4] [c4cpu] [c5] [f1] [f2]
(parity of these paths against the oracle is the tests' business: tests/test_gpu_parity.py)"""
import sys, os, json, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
from gp_vae_amd.gp import exp_kernel2
from gp_vae_amd.vae import StepEngine, SVIState, init_params, decoder_list_from_flat
from gp_vae_amd.inference import GPInference, VAEInference

dev = torch.device("cuda", 0)
which = set(a.lower() for a in sys.argv[1:]) or {"c2", "c4", "c5"}


def timed(f, reps):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def train_rate(tag, x, jitter, dims, B, steps, hyper_sampler=None, group=16, dtype=torch.float32):
    """draws/s of the whole training step (hyper draw -> Gram -> potrf -> draw -> ELBO -> Adam)."""
    n = x.shape[0]
    kw = {} if hyper_sampler is None else {"hyper_sampler": hyper_sampler}
    eng = StepEngine(x.to(dtype), jitter, dims, B, 1e-3, dtype=dtype, seed=1, group=group, **kw)  # priors of vae.py:157-159
    st = SVIState(init_params(n, *dims, 1, dtype, dev))
    eng.run([(0, i) for i in range(16)], st, train=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run([(1, i) for i in range(steps)], st, train=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"config": tag, "draws_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3,
                      "final_loss": float(eng.loss.item())}), flush=True)


def beta_hyper(rng, size=None):
    """the README's lengthscale prior Beta(0.1, 2.0) (README.md:17) in place of InverseGamma(1, 0.1);
    floored at 1e-3: Beta(0.1, 2) puts mass at ls -> 0, where the Gram matrix is the identity."""
    ls = np.maximum(rng.beta(0.1, 2.0, size=size), 1e-3)
    return ls, np.exp(0.1 * rng.standard_normal(size=size)), np.abs(0.1 * rng.standard_normal(size=size))


if "c2" in which:  # 1D grid n=1024, B=4096, fp32, jitter 1e-4: reference MLP (35,30,10), scaled MLP, README prior
    x1 = torch.arange(1024, dtype=torch.float32, device=dev) / 1024
    train_rate("C2: 1D n=1024, B=4096, VAE 1024-35-30-10, f32", x1, 1e-4, (35, 30, 10), 4096, 128)
    train_rate("C2 scaled: 1D n=1024, B=4096, VAE 1024-512-256-64, f32", x1, 1e-4, (512, 256, 64), 4096, 64)
    train_rate("C2 README prior: ls ~ Beta(0.1, 2.0), VAE 1024-35-30-10, f32", x1, 1e-4, (35, 30, 10), 4096, 128, beta_hyper)

if "c3s" in which:  # the bench grid with the scaled MLP
    g = torch.arange(64, dtype=torch.float32, device=dev) / 64
    u, v = torch.meshgrid(g, g, indexing="xy")
    x2 = torch.stack([u.reshape(-1), v.reshape(-1)], 1).contiguous()
    train_rate("C3 scaled: 64x64 grid, B=4096, VAE 4096-512-256-64, f32", x2, 1e-3, (512, 256, 64), 4096, 32)
    train_rate("C3 f64: 64x64 grid, B=4096, VAE 4096-50-30-15, f64", x2, 1e-3, (50, 30, 15), 4096, 64, dtype=torch.float64)

if "c4cpu" in which:  # host LAPACK dpotrf at growing n, extrapolated ~ n^3 to the C4 size
    import scipy.linalg
    rng = np.random.default_rng(0)
    out = {"config": "C4 CPU baseline: scipy.linalg.cholesky f64 (LAPACK dpotrf), host threads = %s" % os.environ.get("OMP_NUM_THREADS", "default")}
    for nn in (4096, 8192):
        g = rng.standard_normal((nn, 64))
        a = g @ g.T + nn * np.eye(nn)
        t0 = time.perf_counter(); scipy.linalg.cholesky(a, lower=True, overwrite_a=True, check_finite=False); t = time.perf_counter() - t0
        out["potrf_s_n%d" % nn] = t
        out["gflops_n%d" % nn] = nn ** 3 / 3 / t / 1e9
    out["potrf_s_n32768_extrapolated"] = out["potrf_s_n8192"] * 64
    print(json.dumps(out), flush=True)

if "c4" in which:  # kriging, 192x192 grid, 32768 observations, fp64
    g = np.arange(192) / 192.0
    u, v = np.meshgrid(g, g)
    x = np.stack([u.ravel(), v.ravel()], 1)
    rng = np.random.default_rng(0)
    obs = np.sort(rng.permutation(x.shape[0])[:32768])
    y = np.sin(3 * np.pi * x[:, 0]) + 0.3 * np.cos(9 * np.pi * x[:, 1]) + 0.5 * np.sin(7 * np.pi * (x[:, 0] + x[:, 1])) \
        + 0.05 * rng.standard_normal(x.shape[0])
    xt = torch.tensor(x, dtype=torch.float64, device=dev)
    x_obs = xt[torch.tensor(obs, device=dev)]
    mask = np.ones(x.shape[0], bool); mask[obs] = False
    x_test = xt[torch.tensor(np.nonzero(mask)[0], device=dev)]
    y_obs = torch.tensor(y[obs], dtype=torch.float64, device=dev)
    n_obs, n_test = x_obs.shape[0], x_test.shape[0]
    # factorisation alone
    A = ops.gram_se(x_obs, x_obs, 1.0, 0.2, 1e-5)
    A.diagonal().add_(0.05)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = ops.potrf_(A, zero_upper=False, nan_on_fail=False)
    torch.cuda.synchronize(); t_potrf = time.perf_counter() - t0
    ok = bool((info == 0).all().item())
    del A
    xi = ops.randn((n_test,), torch.float64, dev, seed=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mean, samp, var_out, info = ops.krig_predict(x_obs, x_test, y_obs, 1.0, 0.2, 0.05, 1e-5, xi)
    torch.cuda.synchronize(); t_pred = time.perf_counter() - t0
    truth = np.sin(3 * np.pi * x[mask, 0]) + 0.3 * np.cos(9 * np.pi * x[mask, 1]) + 0.5 * np.sin(7 * np.pi * (x[mask, 0] + x[mask, 1]))
    rmse = float(np.sqrt(np.mean((mean.cpu().numpy() - truth) ** 2)))
    flop = n_obs ** 3 / 3
    print(json.dumps({"config": "C4: kriging 192x192 grid, n_obs=32768, n_test=%d, f64" % n_test,
                      "potrf_s": t_potrf, "potrf_tflops": flop / t_potrf / 1e12, "potrf_frac_of_f64_peak": flop / t_potrf / 1e12 / 78.6,
                      "pd": ok, "predict_s": t_pred,
                      "predict_tflops": (flop + n_obs ** 2 * (n_test + 1) + 2 * n_obs * n_test) / t_pred / 1e12,
                      "rmse_vs_noise_free_truth": rmse, "min_var": float(var_out.min().item()),
                      "info": int(info.item())}), flush=True)

if "c5" in which:  # potential-gradient evaluations of the VAE-prior regression, 1024 chains
    n, dims = 300, (35, 30, 10)
    flat = init_params(n, *dims, 1, torch.float32, dev)
    dec = decoder_list_from_flat(flat, n, *dims)
    x = np.arange(0, 1, 1 / n)
    yv = np.sin(6 * x)
    obs = [20, 23, 100, 110, 117, 130, 133, 140, 170, 190]
    inf = VAEInference(None, dec, dims[2], 1, x, yv, obs, {}, dtype=torch.float32)
    chains = 1024
    z = ops.randn((chains, dims[2]), torch.float32, dev, seed=5)
    ls = torch.full((chains,), math.log(0.1), device=dev)
    t = timed(lambda: inf.potential_and_grad(z, ls), 50)
    print(json.dumps({"config": "C5: VAE-prior regression potential+gradient, 1024 chains, decoder 10-30-35-300, f32",
                      "grad_evals_per_s": chains / t, "ms_per_batched_eval": t * 1e3}), flush=True)

if "f2" in which:  # batched small kriging: the reference's vmap over posterior samples (inference.py:511-521)
    g = np.arange(25) / 25.0
    u, v = np.meshgrid(g, g)
    x = np.stack([u.ravel(), v.ravel()], 1).astype(np.float32)      # 25x25 grid (2d_examples.py scale)
    rng = np.random.default_rng(0)
    for n_obs in (100, 400):
        obs = np.sort(rng.permutation(x.shape[0])[:n_obs])
        mask = np.ones(x.shape[0], bool); mask[obs] = False
        y = np.sin(4 * x.sum(1)) + 0.1 * rng.standard_normal(x.shape[0])
        S = 8000
        var = rng.lognormal(0, 0.1, S); ls = rng.uniform(0.05, 0.5, S); sig = np.abs(rng.normal(0, 0.1, S)) + 1e-3
        xo = torch.tensor(x[obs], device=dev); xt = torch.tensor(x[mask], device=dev)
        yo = torch.tensor(y[obs], dtype=torch.float32, device=dev)
        xi = ops.randn((S, int(mask.sum())), torch.float32, dev, seed=9)
        t = timed(lambda: ops.krig_predict_batched(xo, xt, yo, var, ls, sig, 1e-5, xi), 3)
        m, p, vv, info = ops.krig_predict_batched(xo, xt, yo, var, ls, sig, 1e-5, xi)
        nl = 100
        tl = timed(lambda: [ops.krig_predict(xo, xt, yo, float(var[s]), float(ls[s]), float(sig[s]), 1e-5, xi[s]) for s in range(nl)], 1) / nl
        flop = S * (n_obs ** 3 / 3 + n_obs ** 2 * (int(mask.sum()) + 1))
        print(json.dumps({"config": "f2: batched kriging, 25x25 grid, n_obs=%d, n_test=%d, S=%d draws, f32" % (n_obs, int(mask.sum()), S),
                          "problems_per_s": S / t, "ms_total": t * 1e3, "tflops": flop / t / 1e12,
                          "one_at_a_time_problems_per_s": 1 / tl,
                          "pd_fraction": float((info == 0).float().mean().item())}), flush=True)

if "f1" in which:  # gp_regression potential + gradient for a batch of chains
    from gp_vae_amd.gp import exp_kernel2
    rng = np.random.default_rng(0)
    for n_obs, chains in ((100, 1024), (400, 1024)):
        g = np.arange(25) / 25.0
        u, v = np.meshgrid(g, g)
        x = np.stack([u.ravel(), v.ravel()], 1)
        obs = np.sort(rng.permutation(x.shape[0])[:n_obs])
        y = np.full(x.shape[0], np.nan); y[obs] = np.sin(4 * x[obs].sum(1))
        inf = GPInference(exp_kernel2, 2, x, y, obs, {}, dtype=torch.float32)
        f = torch.tensor(rng.standard_normal((chains, n_obs)) * 0.3, dtype=torch.float32, device=dev)
        uv = torch.zeros(chains, device=dev); ul = torch.full((chains,), math.log(0.3), device=dev); us = torch.full((chains,), math.log(0.1), device=dev)
        t = timed(lambda: inf.potential_and_grad(f, uv, ul, us), 5)
        flop = chains * (n_obs ** 3 / 3 + 2.0 * n_obs ** 3)
        print(json.dumps({"config": "f1: gp_regression potential+gradient, n_obs=%d, %d chains, f32" % (n_obs, chains),
                          "grad_evals_per_s": chains / t, "ms_per_batched_eval": t * 1e3, "tflops": flop / t / 1e12,
                          "pd_fraction": float((inf.last_info == 0).float().mean().item())}), flush=True)
