"""Developer smoke run on the GPU box: gram / potrf / draw / randn vs the oracle (not a pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib
avail = os.environ.get("GPVAE_DEV_SYMS")
if avail:
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in avail.split(",")}
from gp_vae_amd import ops
from oracle import gpvae_oracle as orc

dev = "cuda"
def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

for dt, tdt in ((np.float32, torch.float32), (np.float64, torch.float64)):
    for n, d in ((16, 1), (300, 1), (625, 2), (1024, 1)):
        if d == 1:
            x = np.arange(0, 1, 1 / n).astype(dt)[:n]
        else:
            g = int(round(n ** 0.5)); gg = np.arange(0, 1, 1 / g); u, v = np.meshgrid(gg, gg)
            x = np.stack([u.ravel(), v.ravel()], 1).astype(dt)
        jit = 1e-4 if d == 1 else 1e-3
        K = orc.exp_kernel2(x, x, 1.1, 0.2, jit)
        xt = torch.tensor(x, device=dev)
        Kd = ops.gram_se(xt, xt, 1.1, 0.2, jit)
        print(dt.__name__, n, d, "gram rel", rel(Kd.cpu().numpy(), K), flush=True)
        Lo = orc.cholesky_lower(K.astype(np.float64))
        A = Kd.clone()
        nx = 70
        rng = np.random.default_rng(0)
        Xh = rng.standard_normal((nx, n)).astype(dt)
        Xd = torch.tensor(Xh, device=dev)
        info = ops.potrf_(A, Xd)
        torch.cuda.synchronize()
        Ld = A.cpu().numpy().astype(np.float64)
        print("   info", info.cpu().numpy(), "potrf rel", rel(Ld, Lo),
              "bwd", rel(Ld @ Ld.T, K.astype(np.float64)), "upper0", float(np.abs(np.triu(Ld, 1)).max()), flush=True)
        Xo = orc.right_solve_lt(Lo, Xh.astype(np.float64))
        print("   trailing rows rel", rel(Xd.cpu().numpy(), Xo), flush=True)
        E = rng.standard_normal((200, n)).astype(dt); E2 = rng.standard_normal((200, n)).astype(dt)
        Fo, Yo = orc.gp_draw(Ld.astype(dt), E, 0.05, E2)
        Y = ops.draw(A, torch.tensor(E, device=dev), torch.tensor(E2, device=dev), 0.05)
        print("   draw rel", rel(Y.cpu().numpy(), Yo), flush=True)
    z = ops.randn((1001,), tdt, dev, seed=1234567890123, stream_id=7).cpu().numpy()
    zo = orc.randn(1001, 1234567890123, 7, dt)
    print(dt.__name__, "randn maxabs", float(np.abs(z - zo).max()), "mean/std", z.mean(), z.std(), flush=True)

# timing at C3 size
n = 4096
g = np.arange(0, 1, 1 / 64); u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1).astype(np.float32), device=dev)
K = torch.empty((n, n), device=dev)
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
def tm(f, it=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it
t_g = tm(lambda: ops.gram_se(x, x, 1.0, 0.3, 1e-3, out=K))
print("gram 4096 ms", t_g * 1e3, "GB/s", n * n * 4 / t_g / 1e9, flush=True)
def fac():
    ops.gram_se(x, x, 1.0, 0.3, 1e-3, out=K); return ops.potrf_(K, zero_upper=False, nan_on_fail=False)
t_f = tm(fac)
print("gram+potrf 4096 ms", t_f * 1e3, "TF", n ** 3 / 3 / (t_f - t_g) / 1e12, "info", fac().item(), flush=True)
t_d = tm(lambda: ops.draw(K, E, E2, 0.1))
print("draw 4096x4096 ms", t_d * 1e3, "TF", n * n * 4096 / t_d / 1e12, flush=True)
Kb = torch.empty((16, n, n), device=dev)
ls = torch.linspace(0.05, 1.0, 16, device=dev)
def facb():
    ops.gram_se(x, x, torch.ones(16, device=dev), ls, 1e-3, out=Kb); return ops.potrf_(Kb, zero_upper=False, nan_on_fail=False)
t_gb = tm(lambda: ops.gram_se(x, x, torch.ones(16, device=dev), ls, 1e-3, out=Kb), 3)
t_fb = tm(facb, 3)
print("batched16 potrf ms", (t_fb - t_gb) * 1e3, "TF", 16 * n ** 3 / 3 / (t_fb - t_gb) / 1e12, "info", facb().cpu().numpy(), flush=True)
