import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import _lib, ops
lib = _lib.load(); dev = "cuda"; n = 4096
E = ops.randn((4096, n), seed=1); E2 = ops.randn((4096, n), seed=2)
L = torch.tril(ops.randn((n, n), seed=3)).contiguous()
sig = torch.full((1,), 0.1, device=dev)
st = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
for _ in range(5): ops.draw(L, E, E2, sig)
lib.gpvae_dev_set_dstamps(C.c_void_p(st.data_ptr()))
ops.draw(L, E, E2, sig); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(1024, 8)
w0 = s[:, 4].min()
ws = (s[:, 4] - w0) / 100.0; we = (s[:, 5] - w0) / 100.0   # us
freq = (s[:, 3] - s[:, 0]) / np.maximum(s[:, 5] - s[:, 4], 1) * 100.0  # MHz
print("kernel span us", we.max(), " mean shader MHz per block", freq.mean(), freq.min(), freq.max())
hw = s[:, 6]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = (hw >> 27) & 0xf  # approximate fields
ct = 31 - np.arange(1024) // 32
pro = (s[:, 1] - s[:, 0]); loop = (s[:, 2] - s[:, 1]); epi = (s[:, 3] - s[:, 2])
tiles = (ct + 1) * 4
print("blk  ct  start_us  end_us  prologue  loop_cyc/tile  epilogue  hwid")
for b in list(range(0, 1024, 64)) + [1023]:
    print(b, ct[b], round(ws[b], 1), round(we[b], 1), pro[b], int(loop[b] / tiles[b]), epi[b], hex(hw[b]))
print("mean cycles/tile by round: first 512:", (loop[:512] / tiles[:512]).mean(), " rest:", (loop[512:] / tiles[512:]).mean())
print("sum over blocks of loop cycles / (1024 SIMD-slots...) :", loop.sum(), " total tiles", tiles.sum())
order = np.argsort(we)
print("last finishing blocks:", [(int(b), int(ct[b]), round(ws[b],1), round(we[b],1)) for b in order[-8:]])
print("start time histogram (us):", np.histogram(ws, bins=10)[0], np.histogram(ws, bins=10)[1].round(0))
