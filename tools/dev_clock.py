import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib
avail = os.environ.get("GPVAE_DEV_SYMS")
if avail:
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in avail.split(",")}
from gp_vae_amd import ops
lib = _lib.load()
dev = "cuda"
def clock():
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    lib.gpvae_dev_clock(C.c_void_p(torch.cuda.current_stream().cuda_stream), 20000, C.c_void_p(out.data_ptr()))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[0] / o[1] * 100.0
print("idle clock MHz", clock(), clock(), flush=True)
n = 4096
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
L = torch.tril(ops.randn((n, n), seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
evs[0].record()
for i in range(100):
    ops.draw(L, E, E2, sig)
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(100)]
print("draw ms first10", [round(t, 3) for t in ts[:10]], "last5", [round(t, 3) for t in ts[-5:]], flush=True)
print("TF last", n * n * 4096 / (ts[-1] * 1e-3) / 1e12)
print("clock right after load MHz", clock(), flush=True)
# clock while draw is running on another stream
s2 = torch.cuda.Stream()
for i in range(20): ops.draw(L, E, E2, sig)
with torch.cuda.stream(s2):
    c = clock()
print("clock during load MHz", c, flush=True)
