import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib, ops
lib = _lib.load(); dev = "cuda"; n = 4096
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
L = torch.tril(ops.randn((n, n), seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev)
st = torch.zeros(64, dtype=torch.int64, device=dev)
for _ in range(3): ops.draw(L, E, E2, sig)
lib.gpvae_dev_set_dstamps(C.c_void_p(st.data_ptr()))
ops.draw(L, E, E2, sig); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(8, 8)[:6, :6]
print("per iteration: issue-loads, mfma, barrier1, store, barrier2 | iter total")
for i in range(6):
    d = np.diff(s[i]); nxt = (s[i + 1][0] - s[i][0]) if i < 5 else 0
    print(d, nxt)
