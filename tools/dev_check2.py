"""Developer GPU check: vae_elbo / adam / decoder / kriging vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
from oracle import gpvae_oracle as orc
dev = "cuda"
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for (B, n, h0, h1, z) in ((100, 300, 35, 30, 10), (37, 625, 50, 30, 15), (256, 1024, 64, 32, 16)):
        rng = np.random.default_rng(1)
        p = orc.init_params(rng, n, h0, h1, z, dt)
        for k in p: p[k] = (p[k] * 10).astype(dt)   # bigger weights => relus both ways
        Y = rng.standard_normal((B, n)).astype(dt); ez = rng.standard_normal((B, z)).astype(dt)
        loss_o, g_o = orc.elbo_loss_and_grads({k: v.astype(np.float64) for k, v in p.items()}, Y.astype(np.float64), ez.astype(np.float64))
        flat = torch.tensor(orc.flatten(p), device=dev)
        grads = torch.empty_like(flat)
        loss = ops.vae_elbo(flat, torch.tensor(Y, device=dev), torch.tensor(ez, device=dev), (h0, h1, z), grads=grads)
        g_d = orc.unflatten(grads.cpu().numpy(), n, h0, h1, z)
        print(dt.__name__, (B, n, h0, h1, z), "loss rel", abs(loss.item() - loss_o) / abs(loss_o),
              "grad rel", {k: float("%.2e" % rel(g_d[k], g_o[k])) for k in orc.PARAM_NAMES}, flush=True)
        loss2 = ops.vae_elbo(flat, torch.tensor(Y, device=dev), torch.tensor(ez, device=dev), (h0, h1, z))
        print("   fwd-only loss rel", abs(loss2.item() - loss_o) / abs(loss_o))
        # adam
        m = rng.standard_normal(flat.numel()).astype(dt) * 0.01; v = np.abs(rng.standard_normal(flat.numel())).astype(dt) * 0.01
        po, mo, vo = orc.adam_step(orc.flatten(p), grads.cpu().numpy(), m, v, 3, 1e-3)
        pt, mt, vt = flat.clone(), torch.tensor(m, device=dev), torch.tensor(v, device=dev)
        ops.adam_(pt, grads, mt, vt, 3, 1e-3)
        print("   adam rel", rel(pt.cpu().numpy(), po), rel(mt.cpu().numpy(), mo), rel(vt.cpu().numpy(), vo))
        # decoder
        dec = np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")])
        Zc = rng.standard_normal((129, z)).astype(dt); dF = rng.standard_normal((129, n)).astype(dt)
        Fd = ops.decoder_fwd(torch.tensor(dec, device=dev), torch.tensor(Zc, device=dev), n, h0, h1)
        dZd = ops.decoder_vjp(torch.tensor(dec, device=dev), torch.tensor(Zc, device=dev), torch.tensor(dF, device=dev), h0, h1)
        p64 = {k: v.astype(np.float64) for k, v in p.items()}
        print("   decoder fwd rel", rel(Fd.cpu().numpy(), orc.decoder(p64, Zc.astype(np.float64))),
              "vjp rel", rel(dZd.cpu().numpy(), orc.decoder_vjp(p64, Zc.astype(np.float64), dF.astype(np.float64))), flush=True)
    # kriging
    for n, d, nobs in ((300, 1, 10), (625, 2, 100), (1500, 1, 700)):
        rng = np.random.default_rng(2)
        if d == 1: x = np.arange(0, 1, 1 / n)[:n].astype(dt)
        else:
            g = int(round(n ** 0.5)); gg = np.arange(0, 1, 1 / g); u, v = np.meshgrid(gg, gg); x = np.stack([u.ravel(), v.ravel()], 1).astype(dt)
        obs = np.sort(rng.permutation(n)[:nobs])
        y = np.sin(6 * x.reshape(n, -1).sum(1)).astype(dt) + (0.05 * rng.standard_normal(n)).astype(dt)
        xi = rng.standard_normal(n - nobs).astype(dt)
        mo, so, vo = orc.gp_prediction(x.astype(np.float64), y.astype(np.float64), obs, 1.0, 0.2, 0.05, xi.astype(np.float64))
        xt = torch.tensor(x, device=dev)
        mask = np.ones(n, bool); mask[obs] = False
        md, sd, vd, info = ops.krig_predict(xt[torch.tensor(obs, device=dev)], xt[torch.tensor(np.where(mask)[0], device=dev)],
                                            torch.tensor(y[obs], device=dev), 1.0, 0.2, 0.05, 1e-5, torch.tensor(xi, device=dev))
        print(dt.__name__, "krig", (n, d, nobs), "info", info.item(), "mean rel", rel(md.cpu().numpy(), mo), "var rel", rel(vd.cpu().numpy(), vo),
              "sample rel", rel(sd.cpu().numpy(), so), flush=True)
# timing C3 VAE
B = n = 4096; h0, h1, z = 50, 30, 15
P = ops.vae_param_count(n, h0, h1, z)
flat = ops.randn((P,), seed=5) * 0.01
Y = ops.randn((B, n), seed=6); ez = ops.randn((B, z), seed=7)
grads = torch.empty_like(flat); work = ops.VaeWorkspace(B, n, h0, h1, z, torch.float32, dev); loss = torch.empty(1, device=dev)
m = torch.zeros_like(flat); v = torch.zeros_like(flat)
def step():
    ops.vae_elbo(flat, Y, ez, (h0, h1, z), grads=grads, loss=loss, work=work); ops.adam_(flat, grads, m, v, 1, 1e-3)
step(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("vae step C3 ms", (time.perf_counter() - t) / 10 * 1e3, flush=True)
