import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
from oracle import gpvae_oracle as orc
dev = "cuda"
def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
dt = np.float64
n, nobs = 300, 10
rng = np.random.default_rng(2)
x = np.arange(0, 1, 1 / n)[:n].astype(dt)
obs = np.sort(rng.permutation(n)[:nobs])
y = np.sin(6 * x).astype(dt)
mo, so, vo = orc.gp_prediction(x, y, obs, 1.0, 0.2, 0.05, None)
mask = np.ones(n, bool); mask[obs] = False
xt = torch.tensor(x, device=dev)
xo = xt[torch.tensor(obs, device=dev)]; xs = xt[torch.tensor(np.where(mask)[0], device=dev)]
yo = torch.tensor(y[obs], device=dev)
A = ops.gram_se(xo, xo, 1.0, 0.2, 1e-5).clone(); A += 0.05 * torch.eye(nobs, device=dev, dtype=A.dtype)
Kpx = ops.gram_se(xs, xo, 1.0, 0.2, 1e-5)
print("kxx rel", rel(A.cpu().numpy(), orc.exp_kernel2(x[obs], x[obs], 1.0, 0.2) + 0.05 * np.eye(nobs)))
print("kpx rel", rel(Kpx.cpu().numpy(), orc.exp_kernel2(x[mask], x[obs], 1.0, 0.2)))
X = torch.cat([Kpx, yo[None, :]], 0).contiguous()
ops.potrf_(A, X)
mean = X[:-1] @ X[-1]
print("manual mean rel", rel(mean.cpu().numpy(), mo))
md, sd, vd, info = ops.krig_predict(xo, xs, yo, 1.0, 0.2, 0.05, 1e-5, None)
print("krig mean rel", rel(md.cpu().numpy(), mo), "vs manual", rel(md.cpu().numpy(), mean.cpu().numpy()))
print(md[:5].cpu().numpy(), mean[:5].cpu().numpy(), mo[:5])
