import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib
avail = os.environ.get("GPVAE_DEV_SYMS")
if avail:
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k in avail.split(",")}
from gp_vae_amd import ops
lib = _lib.load(); dev = "cuda"
st = torch.zeros(8, dtype=torch.int64, device=dev)
lib.gpvae_dev_set_stamps(C.c_void_p(st.data_ptr()))
for dt in (torch.float32, torch.float64):
    for n in (4096, 4096 - 192):   # last panel launch has K = 192 (n=4096) or K = 0
        g = torch.arange(n, device=dev, dtype=dt) / n
        K = ops.gram_se(g, g, 1.0, 0.05, 1e-3)
        ops.potrf_(K, zero_upper=False, nan_on_fail=False)
        torch.cuda.synchronize()
        s = st.cpu().numpy()
        print(dt, n, "phase cycles (gemm, toLDS, diag, trsm, store):", np.diff(s[:6]), flush=True)
# first panel of a matrix where all 16 WGs exist: use small outer: n=1088 -> last launch c0=1024,K=0 ; 
