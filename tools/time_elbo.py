"""ELBO forward + backward at the C3 shape, timed alone (HIP events), per kernel via rocprofv3 if wanted."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_vae_amd import ops
from gp_vae_amd.vae import init_params
dev = torch.device("cuda", 0)
B, n, dims = 4096, 4096, (50, 30, 15)
p = init_params(n, *dims, 1, torch.float32, dev)
Y = ops.randn((B, n), seed=1); ez = ops.randn((B, dims[2]), seed=2)
g = torch.empty_like(p); w = ops.VaeWorkspace(B, n, *dims, torch.float32, dev); loss = torch.empty(1, device=dev)
for _ in range(5): ops.vae_elbo(p, Y, ez, dims, 0.1, grads=g, loss=loss, work=w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.vae_elbo(p, Y, ez, dims, 0.1, grads=g, loss=loss, work=w)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"variant": os.environ.get("GPVAE_A1V", "0"), "elbo_us": e0.elapsed_time(e1) / 50 * 1e3, "loss": loss.item()}))
