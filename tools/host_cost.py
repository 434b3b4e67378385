"""Host cost of the step's ABI calls at the C1 shapes (n = 300, B = 100; any argument: C2, n = 1024, B = 4096):
time to ENQUEUE and time to finish, per call."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_vae_amd import ops
from gp_vae_amd.vae import StepEngine, SVIState, init_params
dev = torch.device("cuda", 0)
n, B, dims = (300, 100, (35, 30, 10)) if len(sys.argv) < 2 else (1024, 4096, (35, 30, 10))
x = torch.arange(n, dtype=torch.float32, device=dev) / n
out = {}
for overlap in (True, False):
    eng = StepEngine(x, 1e-4, dims, B, 1e-3, dtype=torch.float32, seed=1, group=16, overlap=overlap)
    st = SVIState(init_params(n, *dims, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(64)], st, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.run([(1, i) for i in range(512)], st, train=True); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out["engine_overlap_%s" % overlap] = {"host_ms": (t1 - t0) / 512 * 1e3, "total_ms": (t2 - t0) / 512 * 1e3}
eng.run([(0, i) for i in range(4)], st, train=True)
torch.cuda.synchronize()
def host(f, reps=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return {"host_us": (t1 - t0) / reps * 1e6, "total_us": (t2 - t0) / reps * 1e6}
out["normals(3 launches)"] = host(lambda: eng._normals((0, 1), 0, eng._h_main))
def draw():
    if eng.b3:
        return eng._lib.gpvae_draw_b3(eng._h_main, n, B, eng._pL3[0][0], n, n * n, eng._pE3, n, B * n, eng._pE2, n,
                                      eng._psig[0][0], None, 0, eng._pY[0], n)
    return eng._lib.gpvae_draw(eng._h_main, eng._dtc, n, B, eng._pK[0][0], eng.ld, eng._pE, n, eng._pE2, n,
                               eng._psig[0][0], None, 0, eng._pY[0], n)
out["draw(1)"] = host(draw)
out["elbo(8)"] = host(lambda: eng._learn_start(0, st, True, 0))
out["adam(1)"] = host(lambda: eng._learn_finish(None, st, True))
ev = torch.cuda.Event()
s2 = torch.cuda.Stream()
out["event record+wait"] = host(lambda: (ev.record(eng._main), s2.wait_event(ev)))
out["randn x1"] = host(lambda: eng._lib.gpvae_randn(eng._h_main, eng._dtc, B * 10, 1, 5, eng._pez[0]))
print(json.dumps(out, indent=1))
