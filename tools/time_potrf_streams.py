"""Dev experiment: batched potrf split over k HIP streams (panel latency of one part under the
trailing updates of the others)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gp_vae_amd import ops
dev = "cuda"; n = 4096
g = int(np.sqrt(n)); xs = torch.linspace(0, 1, g, device=dev)
X = torch.stack(torch.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2).contiguous()
for batch in (16, 32):
    ls = torch.linspace(0.05, 1.0, batch, device=dev); var = torch.ones(batch, device=dev)
    K = ops.gram_se(X, X, var, ls, jitter=1e-3)
    A = K.clone()
    for parts in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        per = batch // parts
        def run():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(cur)
            for i, s in enumerate(streams):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    ops.potrf_(A[i * per:(i + 1) * per], zero_upper=False, nan_on_fail=False)
                e = torch.cuda.Event(); e.record(s); cur.wait_event(e)
        ts = []
        for it in range(4):
            A.copy_(K); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts[1:]))
        print(f"batch {batch} parts {parts}: {t:.3f} ms  {batch*n**3/3/t/1e9:.1f} TF", flush=True)
