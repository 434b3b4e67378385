"""f16x2 engine: stop at a step whose loss is off and look at its draw (dev tool)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from gp_vae_amd import ops
from gp_vae_amd.vae import StepEngine, SVIState, init_params

dev = torch.device("cuda", 0)
x = torch.tensor(bench.grid_points(), device=dev)
last = int(sys.argv[1]) if len(sys.argv) > 1 else 42
n = bench.N
eng = StepEngine(x, bench.JITTER, bench.DIMS, bench.B, bench.LR, dtype=torch.float32, seed=1,
                 hyper_sampler=bench.bench_hyper, group=16, mfma="f16x2")
st = SVIState(init_params(bench.N, *bench.DIMS, 1, torch.float32, dev))
eng.run([(0, i) for i in range(5)], st, train=True)
eng.run([(1, i) for i in range(last + 1)], st, train=True)
torch.cuda.synchronize()
print("last losses", eng.loss_steps.cpu().numpy()[-4:])
slot, i, buf = (last // 16) & 1, last % 16, last & 1
Y = eng.Ybuf[buf]
print("step %d slot %d i %d: max|Y| %.4e  finite %s  params max %.3e  ez max %.3e" %
      (last, slot, i, Y.abs().max().item(), bool(torch.isfinite(Y).all()), st.params.abs().max().item(), eng.ezbuf[buf].abs().max().item()))
hd = eng.hyper_dev[slot]
print("hyper var %.4g ls %.4g sigma %.4g  inv %.4e" % (hd[0, i].item(), hd[1, i].item(), hd[2, i].item(), eng.L2inv[slot][i].item()))
L = torch.tril(eng.Kbuf[slot][i][:, :n]).double()
Eh = eng.Eh.float().sum(0).double() / 4096.0
Yo = Eh @ L.T + hd[2, i].double() * eng.E2.double()
print("Y vs recomputation from the engine's buffers: max diff %.4e  max|Yo| %.4e" % ((Y.double() - Yo).abs().max().item(), Yo.abs().max().item()))
Y2 = ops.draw_h2(eng.L2buf[slot][i], eng.L2inv[slot][i:i + 1], eng.Eh, eng.E2, hd[2, i:i + 1])
print("draw_h2 again from the same buffers: max diff to Yo %.4e" % (Y2.double() - Yo).abs().max().item())
print("max|E| %.3f  max|E2| %.3f" % (Eh.abs().max().item(), eng.E2.abs().max().item()))
K = ops.gram_se(x.float(), x.float(), hd[0, i:i + 1], hd[1, i:i + 1], bench.JITTER).reshape(n, n).double()
print("engine factor: max|LL^T - K| / max K_ii = %.3e" % ((L @ L.T - K).abs().max() / K.diagonal().max()).item())
A = K.float().clone()
ops.potrf_(A, zero_upper=False, nan_on_fail=False)
Lr = torch.tril(A).double()
print("plain potrf:   max|LL^T - K| / max K_ii = %.3e   max|L - Lref| %.3e" %
      (((Lr @ Lr.T - K).abs().max() / K.diagonal().max()).item(), (L - Lr).abs().max().item()))
Yr = Eh @ Lr.T + hd[2, i].double() * eng.E2.double()
print("max|Y - Y(plain factor)| %.3e ;  row norms of Y: engine %.4e plain %.4e" %
      ((Y.double() - Yr).abs().max().item(), Y.double().norm(dim=1).max().item(), Yr.norm(dim=1).max().item()))
ez = eng.ezbuf[buf]
la = ops.vae_elbo(st.params, Y, ez, bench.DIMS, eng.obs_sigma).item()
lb = ops.vae_elbo(st.params, Yr.float().contiguous(), ez, bench.DIMS, eng.obs_sigma).item()
mu, sd = ops.encoder_fwd(st.params, Y, bench.DIMS)[:2]
mur, sdr = ops.encoder_fwd(st.params, Yr.float().contiguous(), bench.DIMS)[:2]
print("ELBO loss (params after the step) on the engine's Y %.5e, on Y from the plain factor %.5e" % (la, lb))
print("encoder: max|mu| %.3e max sd %.3e   (plain factor: %.3e %.3e)" % (mu.abs().max().item(), sd.max().item(), mur.abs().max().item(), sdr.max().item()))
