"""As time_draw_pad.py but with the engine's real operands: L from gpvae_potrf_b3 on a C3 Gram matrix (planes written
by the factorisation with row stride n + pad), E from gpvae_randn_b3_rows.  (dev tool)"""
import sys, ctypes as C
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
n = B = 4096
vp = lambda t: C.c_void_p(t.data_ptr())
x = torch.tensor(bench.grid_points(), device=dev, dtype=torch.float32)
E2 = ops.randn((B, n), torch.float32, dev, seed=2)
sig = torch.full((1,), 0.1, device=dev)
Y = torch.empty((B, n), device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for ls in (0.1, 1.0):
    for rep in range(2):
        for pad in (0, 64, 192):
            ldp = n + pad
            K = ops.gram_se(x, x, 1.0, ls, bench.JITTER).reshape(1, n, n)
            L3 = torch.zeros((1, 3, n, ldp), dtype=torch.bfloat16, device=dev)
            info = torch.zeros(1, dtype=torch.int32, device=dev)
            work = torch.empty(lib.gpvae_potrf_b3_workspace_bytes(n, 1) + 256, dtype=torch.uint8, device=dev)
            assert lib.gpvae_potrf_b3(st, n, 1, vp(K), n, n * n, vp(info), 0, vp(work), vp(L3), ldp, n * ldp, 3 * n * ldp) == 0
            E3 = torch.zeros((3, B, ldp), dtype=torch.bfloat16, device=dev)
            assert lib.gpvae_randn_b3_rows(st, B, n, 1, 7, vp(E3), ldp, B * ldp) == 0
            def run():
                rc = lib.gpvae_draw_b3(st, n, B, vp(L3), ldp, n * ldp, vp(E3), ldp, B * ldp, vp(E2), n, vp(sig), None, 0, vp(Y), n)
                assert rc == 0, lib.gpvae_last_error()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize()
            print("ls %.1f rep %d pad %3d: %.4f ms" % (ls, rep, pad, e0.elapsed_time(e1) / 20), flush=True)
