import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib, ops
lib = _lib.load(); dev = "cuda"
st = torch.zeros(8 * 128, dtype=torch.int64, device=dev)
lib.gpvae_dev_set_stamps(C.c_void_p(st.data_ptr()))
for dt in (torch.float32,):
    n = 4096
    g = torch.arange(n, device=dev, dtype=dt) / n
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    K = ops.gram_se(g, g, torch.ones(batch, device=dev, dtype=dt), torch.full((batch,), 0.05, device=dev, dtype=dt), 1e-3)
    ops.potrf_(K, zero_upper=False, nan_on_fail=False)
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(128, 8)
    print(dt, "panel launch: phase cycles (copy+gemm, subtract, diag, trsm, store)")
    for j in (0, 1, 2, 3, 4, 5, 6, 7, 32, 33, 34, 35, 60, 61, 62, 63):
        print(j, np.diff(s[j, :6]), 'issue', s[j,6]-s[j,0], 'stored', s[j,7]-s[j,0], flush=True)
