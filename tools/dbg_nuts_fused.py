"""Debug: one fused NUTS iteration at max_depth 1 / 3, then compare the kernel's last potential evaluation with ops.vae_potential,
and the lock-step ops on the same buffers."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops, _lib, nuts as N
from oracle import gpvae_oracle as orc
lib = _lib.load()
tdt = torch.float64
dev = "cuda"
n, (h0, h1, z) = 300, (35, 30, 10)
p = {k: 18 * v for k, v in orc.init_params(np.random.default_rng(2), n, h0, h1, z, np.float64).items()}
dec = torch.tensor(np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]), dtype=tdt, device=dev)
obs = torch.tensor([20, 23, 100, 110, 117, 130, 133, 140, 170, 190], device=dev)
y = torch.tensor(np.random.default_rng(1).standard_normal(10), dtype=tdt, device=dev)
Cn, D = 13, z + 1
md = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = N.uniforms_per_iteration(md)
g = torch.Generator().manual_seed(5)
q = (0.3 * torch.randn((Cn, D), dtype=torch.float64, generator=g)).to(dev, tdt)
q[:, z] = np.log(0.07)
U, gr = ops.vae_potential(dec, q, y, obs, n, h0, h1)
def mk():
    vec = lambda: torch.zeros((Cn, D), dtype=tdt, device=dev)
    sca = lambda: torch.zeros(Cn, dtype=tdt, device=dev)
    flg = lambda: torch.zeros(Cn, dtype=torch.int32, device=dev)
    t = {k: vec() for k in ("ql", "pl", "gl", "qr", "pr", "gr", "q_prop", "g_prop", "p_sum", "qe", "pe", "ge", "sub_psum", "sq", "sg", "ph", "qn")}
    t.update({k: sca() for k in ("U_prop", "logW", "H0", "sum_acc", "n_prop", "accept", "es", "sub_logW", "sU")})
    t.update({k: flg() for k in ("active", "diverged", "sub_stop", "building", "right")})
    t["ck_p"] = torch.zeros((md + 1, Cn, D), dtype=tdt, device=dev); t["ck_s"] = torch.zeros((md + 1, Cn, D), dtype=tdt, device=dev)
    t["flags"] = torch.zeros(2, dtype=torch.int32, device=dev)
    t["z"] = torch.randn((Cn, D), dtype=torch.float64, generator=torch.Generator().manual_seed(9)).to(dev, tdt)
    t["unif"] = torch.rand((Cn, K), dtype=torch.float64, generator=torch.Generator().manual_seed(10)).to(dev, tdt)
    t["eps"] = torch.full((Cn,), 0.05, dtype=tdt, device=dev)
    t["q"], t["U"], t["g"] = q.clone(), U.clone(), gr.clone()
    t["minv"] = torch.ones((Cn, D), dtype=tdt, device=dev)
    t["Un"], t["gn"] = sca(), vec()
    b = N._Buffers()
    for k, v in t.items():
        setattr(b, k, v.data_ptr())
    b.ldu, b.ldg = K, D
    return t, b
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ta, ba = mk()
rc = lib.gpvae_nuts_vae_iteration(st, 1, Cn, md, C.byref(ba), n, h0, h1, z, C.c_void_p(dec.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(obs.data_ptr()), 10, 0.1)
assert rc == 0, lib.gpvae_last_error()
torch.cuda.synchronize()
U2, g2 = ops.vae_potential(dec, ta["qn"], y, obs, n, h0, h1)
print("fused potential vs gpvae_vae_potential at the last leaf: U", (ta["Un"] - U2).abs().max().item(), "g", (ta["gn"] - g2).abs().max().item())
# lock-step on fresh buffers
tb, bb = mk()
def op(code, depth=0, leaf=0, lo=0, hi=0):
    rc = lib.gpvae_nuts_op(st, 1, code, Cn, D, md, depth, leaf, lo, hi, C.byref(bb)); assert rc == 0, lib.gpvae_last_error()
op(0)
for depth in range(md):
    op(1, depth)
    leaves = 1 << depth
    for leaf in range(leaves):
        if leaf == 0: op(2, depth)
        Un, gn = ops.vae_potential(dec, tb["qn"], y, obs, n, h0, h1)
        tb["Un"].copy_(Un); tb["gn"].copy_(gn)
        lo, hi = N._ckpt_range(leaf)
        op(3 if leaf + 1 == leaves else 6, depth, leaf, lo, hi)
    op(4, depth)
op(5)
torch.cuda.synchronize()
for k in ("q", "U", "g", "accept", "n_prop", "active", "logW", "p_sum", "ql", "qr", "q_prop", "sub_logW", "H0", "es", "sum_acc"):
    d = (ta[k].double() - tb[k].double()).abs().max().item()
    print("%-10s max diff %.3e   fused %s  lockstep %s" % (k, d, ta[k].flatten()[:3].tolist(), tb[k].flatten()[:3].tolist()))
print("---- several iterations, both drivers from the same state")
ta, ba = mk(); tb, bb = mk()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    zz = torch.randn((Cn, D), dtype=torch.float64, generator=torch.Generator().manual_seed(100 + it)).to(dev, tdt)
    uu = torch.rand((Cn, K), dtype=torch.float64, generator=torch.Generator().manual_seed(200 + it)).to(dev, tdt)
    for t in (ta, tb):
        t["z"].copy_(zz); t["unif"].copy_(uu)
    rc = lib.gpvae_nuts_vae_iteration(st, 1, Cn, md, C.byref(ba), n, h0, h1, z, C.c_void_p(dec.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(obs.data_ptr()), 10, 0.1)
    assert rc == 0
    op(0)
    for depth in range(md):
        op(1, depth)
        leaves = 1 << depth
        for leaf in range(leaves):
            if leaf == 0: op(2, depth)
            Un, gn = ops.vae_potential(dec, tb["qn"], y, obs, n, h0, h1)
            tb["Un"].copy_(Un); tb["gn"].copy_(gn)
            lo, hi = N._ckpt_range(leaf)
            op(3 if leaf + 1 == leaves else 6, depth, leaf, lo, hi)
        op(4, depth)
    op(5)
    torch.cuda.synchronize()
    print(it, "accept diff %.3e sum_acc diff %.3e U diff %.3e" % ((ta["accept"] - tb["accept"]).abs().max().item(), (ta["sum_acc"] - tb["sum_acc"]).abs().max().item(), (ta["U"] - tb["U"]).abs().max().item()))
    print(it, "q diff %.3e" % (ta["q"] - tb["q"]).abs().max().item(), "n_prop fused", ta["n_prop"].tolist(), "lockstep", tb["n_prop"].tolist())
