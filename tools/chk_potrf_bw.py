"""Backward error max |L L^T - K| of the float32 factorisations at C3 over the length-scale ladder: gpvae_potrf_b3
(bf16 x 3 trailing updates), gpvae_potrf with one extra row (forces the f32-MFMA trailing updates), LAPACK spotrf;
where the largest error sits and whether the diagonal error is biased.  python tools/chk_potrf_bw.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gp_vae_amd import ops

G = int(os.environ.get("GRID", "64"))
n, jit = G * G, float(os.environ.get("JIT", "1e-3"))
g = np.arange(0, 1, 1 / G)
u, v = np.meshgrid(g, g)
x = torch.tensor(np.stack([u.ravel(), v.ravel()], 1), dtype=torch.float32, device="cuda")
ladder = [0.02, 0.05, 0.3, 1.1, 4.4, 55.0, 1e3]
if len(sys.argv) > 1:
    ladder = [float(a) for a in sys.argv[1:]]


def blocked_numpy(K, NB=256):
    """right-looking blocked Cholesky in float32 numpy (sgemm accumulates in f32): what a plain f32 blocked algorithm gives"""
    A = K.copy()
    for j in range(0, n, NB):
        e = min(n, j + NB)
        A[j:e, j:e] = np.linalg.cholesky(A[j:e, j:e])
        if e < n:
            import scipy.linalg
            A[e:, j:e] = scipy.linalg.solve_triangular(A[j:e, j:e], A[e:, j:e].T, lower=True).T.astype(np.float32)
            A[e:, e:] -= A[e:, j:e] @ A[e:, j:e].T
    return np.tril(A)


def bw(L, K):
    Ld = torch.tril(L).double()
    R = Ld @ Ld.T - K.double()
    a = R.abs()
    i = int(a.argmax().item())
    d = torch.diagonal(R)
    return a.max().item(), (i // n, i % n), d.mean().item(), d.abs().max().item(), torch.tril(R, -1).abs().max().item(), R.pow(2).mean().sqrt().item()


for ls in ladder:
    K = ops.gram_se(x, x, 1.0, ls, jit)
    A = K.clone()
    info = ops.potrf_b3_(A)[0] if ops.potrf_b3_supported(n) else ops.potrf_(A, zero_upper=False, nan_on_fail=False)
    B = K.clone()
    X = torch.zeros((1, n), dtype=torch.float32, device="cuda")
    info2 = ops.potrf_(B, X, zero_upper=False, nan_on_fail=False)
    Ll = torch.tensor(np.linalg.cholesky(K.cpu().numpy()), device="cuda")
    print("ls %-8g info %d %d" % (ls, info.item(), info2.item()))
    Lb = torch.tensor(blocked_numpy(K.cpu().numpy()), device="cuda")
    L64 = torch.tensor(np.linalg.cholesky(K.cpu().numpy().astype(np.float64)), device="cuda")  # f64 factor of the SAME f32 matrix
    for name, L in (("b3", A), ("f32mfma", B), ("lapack", Ll), ("np-block", Lb)):
        m, at, dmean, dmax, omax, rms = bw(L, K)
        fwd = (torch.linalg.norm(torch.tril(L).double() - L64) / torch.linalg.norm(L64)).item()
        print("   %-8s max %.3e at %s  diag mean %+.3e max %.3e  offdiag max %.3e  rms %.3e  |L-L64|/|L64| %.3e" % (name, m, at, dmean, dmax, omax, rms, fwd))
