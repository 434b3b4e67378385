"""Dev experiment: batched potrf replayed from a HIP graph vs launched kernel by kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096
g = int(np.sqrt(n)); xs = torch.linspace(0, 1, g, device=dev)
X = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2).contiguous()
for batch in (1, 16, 32):
    ls = torch.linspace(0.05, 1.0, batch, device=dev); var = torch.ones(batch, device=dev)
    K = ops.gram_se(X, X, var, ls, jitter=1e-3)
    A = K.clone()
    from gp_vae_amd import _lib
    lib = _lib.load()
    wb = lib.gpvae_potrf_workspace_bytes(0, n, 0, batch)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    info = torch.zeros(batch, dtype=torch.int32, device=dev)
    def call():
        ops.potrf_(A, zero_upper=False, nan_on_fail=False, work=work)
    call(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        call()
    def tm(f):
        ts = []
        for it in range(4):
            A.copy_(K); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))
    t0 = tm(call); t1 = tm(gr.replay)
    L = A.clone(); A.copy_(K); call(); torch.cuda.synchronize()
    print(f"batch {batch}: launches {t0:.3f} ms, graph {t1:.3f} ms  ({batch*n**3/3/t1/1e9:.1f} TF) same={torch.equal(L, A)}", flush=True)
