"""GP-prior VAE trainer — mirrors /root/reference/src/model/vae.py on top of libgpvae_hip.

`VAE` keeps the reference constructor (vae.py:39-52) and `fit` / `epoch_train` / `eval_test`
contract (fit returns the decoder parameter list `[(W,b),(),(W,b),(),(W,b)]`, vae.py:257).  What
NumPyro's SVI/Trace_ELBO/Adam and Predictive(GP.sample) do per step is done by five entry points of
the C ABI: gpvae_gram_se -> gpvae_potrf -> gpvae_draw -> gpvae_vae_elbo -> gpvae_adam.

MI355X-first structure of the step loop (`StepEngine`):
  * the Gram build + Cholesky of step t+1 does not depend on the network, so it runs on its own HIP
    stream one step AHEAD of the draw/ELBO/Adam chain of step t (double-buffered factor); the
    latency-bound panel sweep hides behind the MFMA/HBM-bound kernels;
  * hyper-parameters (ls, var, sigma) are drawn on the host (vae.py:157-159 priors) and reach the
    device as three scalars per step through pinned staging;
  * standard normals come from the library's Philox kernel keyed by (seed, rank, step);
  * data parallel: each rank owns its (ls,var,sigma), factor and B draws; ONE all-reduce (RCCL) of
    the flat gradient per step, then every rank applies the same Adam update.  The loss is a SUM
    over the batch (vae.py:165-166), so summing gradients over ranks equals one step on the
    concatenated batch.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from ._lib import GpvaeError, check
from .ops import _dt, _p, _stream
from .gp import Predictive, _as_device, _seed_of, draw_hyper
from .parallel import allreduce_sum_start, allreduce_wait, allreduce_sum_, dist_info, philox_stream_id, step_rng


class SVIState():
    """What numpyro's SVIState carries for this model: parameters, Adam moments, step count."""

    def __init__(self, params, m=None, v=None, t=0):
        self.params = params
        self.m = torch.zeros_like(params) if m is None else m
        self.v = torch.zeros_like(params) if v is None else v
        self.t = t


def init_params(n, h0, h1, z, seed, dtype, device):
    """stax.randn() weights and default biases: 1e-2 * N(0,1) (vae.py:76-95), flat ABI layout."""
    P = ops.vae_param_count(n, h0, h1, z)
    flat = ops.randn((P,), dtype, device, seed=seed, stream_id=0x51A7)
    flat.mul_(0.01)
    return flat


def decoder_list_from_flat(flat, n, h0, h1, z):
    """decoder$params in the stax list layout the reference pickles (toy_examples.py:164-165)."""
    o = n * h0 + h0 + h0 * h1 + h1 + 2 * (h1 * z + z)
    V1 = flat[o:o + z * h1].reshape(z, h1); o += z * h1
    c1 = flat[o:o + h1]; o += h1
    V2 = flat[o:o + h1 * h0].reshape(h1, h0); o += h1 * h0
    c2 = flat[o:o + h0]; o += h0
    V3 = flat[o:o + h0 * n].reshape(h0, n); o += h0 * n
    c3 = flat[o:o + n]
    return [(V1, c1), (), (V2, c2), (), (V3, c3)]


def decoder_flat_from_list(params_list, dtype=None, device=None):
    (V1, c1), _, (V2, c2), _, (V3, c3) = params_list
    parts = [_as_device(a, dtype).reshape(-1) for a in (V1, c1, V2, c2, V3, c3)]
    return torch.cat(parts).contiguous()


class StepEngine():
    """One training/evaluation step = gram -> potrf -> draw -> ELBO (-> all-reduce -> Adam).

    The factors of the next `group` steps are independent of the network, so they are built as ONE
    batched Gram + batched Cholesky launch sequence on a side stream (throughput mode of the panel
    sweep instead of `group` latency-bound single factorisations), one group ahead of the
    draw/ELBO/Adam chain; 2 x group factor slots live in HBM (group=8, n=4096, f32: 1 GiB)."""

    def __init__(self, x, jitter, dims, batch_size, lr, dtype=torch.float32, seed=0, obs_sigma=0.1,
                 hyper_sampler=draw_hyper, overlap=True, group=8):
        self.x = _as_device(x, dtype)
        self.n = self.x.shape[0]
        self.d = self.x.shape[1] if self.x.dim() == 2 else 1
        self.h0, self.h1, self.z = dims
        self.B = batch_size
        self.lr = lr
        self.jitter = jitter
        self.dtype = dtype
        self.seed = int(seed)
        self.obs_sigma = obs_sigma
        self.hyper_sampler = hyper_sampler
        self.rank, self.world = dist_info()
        if (dist.is_available() and dist.is_initialized() and overlap
                and int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < 8):
            # the collective's stream would share a hardware queue with the factor stream and wait
            # behind whole factor groups (gp_vae_amd/__init__.py)
            import warnings
            warnings.warn("gp_vae_amd: GPU_MAX_HW_QUEUES < 8 with torch.distributed active — set it to 8 "
                          "before the first HIP call, or the factor stream no longer overlaps the training steps")
        dev = self.x.device
        n, B = self.n, self.B
        self.G = max(1, int(group))
        self.ld = (n + 3) // 4 * 4  # 16-byte aligned rows for the vector loads
        G = self.G
        self.Kbuf = [torch.empty((G, n, self.ld), dtype=dtype, device=dev) for _ in range(2)]
        self.hyper_dev = [torch.empty((3, G), dtype=dtype, device=dev) for _ in range(2)]  # var|ls|sigma
        self.hyper_pin = [torch.empty((3, G), dtype=dtype).pin_memory() for _ in range(2)]
        self.info = [torch.zeros(G, dtype=torch.int32, device=dev) for _ in range(2)]
        wbytes = _lib.load().gpvae_potrf_workspace_bytes(_dt(self.x), n, 0, G)
        self.pwork = [torch.empty(max(16, wbytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.E = torch.empty((B, n), dtype=dtype, device=dev)
        self.E2 = torch.empty((B, n), dtype=dtype, device=dev)
        # two draw buffers: step t+1 is drawn while step t's gradient all-reduce is in flight
        self.Ybuf = [torch.empty((B, n), dtype=dtype, device=dev) for _ in range(2)]
        self.ezbuf = [torch.empty((B, self.z), dtype=dtype, device=dev) for _ in range(2)]
        self.grads = torch.empty(ops.vae_param_count(n, self.h0, self.h1, self.z), dtype=dtype,
                                 device=dev)
        self.loss = torch.zeros(1, dtype=dtype, device=dev)        # the last step's loss
        self.loss_steps = self.loss
        self.loss_sum = torch.zeros(1, dtype=dtype, device=dev)
        self.work = ops.VaeWorkspace(B, n, self.h0, self.h1, self.z, dtype, dev)
        self.overlap = overlap
        self.s_gp = torch.cuda.Stream(device=dev) if overlap else None
        self.ev_ready = [torch.cuda.Event() for _ in range(2)]   # factors of group slot s are ready
        self.ev_free = [torch.cuda.Event() for _ in range(2)]    # group slot s has been consumed
        self.ev_copied = [torch.cuda.Event() for _ in range(2)]  # slot s's hyper upload is done
        self.draw_events = None  # bench.py: list of (start, end) HIP events around gpvae_draw

    # -- stage A: the factors of one group of steps (own stream) ---------------------------------------
    def _issue_factors(self, keys, slot):
        g = len(keys)
        pin = self.hyper_pin[slot]
        self.ev_copied[slot].synchronize()  # the staging block's previous upload has left the host
        for i, key in enumerate(keys):
            ls, var, sigma = self.hyper_sampler(step_rng(self.seed, self.rank, key))
            pin[0, i], pin[1, i], pin[2, i] = float(var), float(ls), float(sigma)
        for i in range(g, self.G):  # unused slots of a short last group: harmless values
            pin[0, i], pin[1, i], pin[2, i] = 1.0, 1.0, 0.0
        stream = self.s_gp if self.overlap else torch.cuda.current_stream()
        if self.overlap:
            stream.wait_event(self.ev_free[slot])  # the draws that read this slot's factors are done
        with torch.cuda.stream(stream):
            hd = self.hyper_dev[slot]
            hd.copy_(pin, non_blocking=True)
            self.ev_copied[slot].record(stream)
            K = self.Kbuf[slot]
            lib = _lib.load()
            check(lib.gpvae_gram_se(_stream(), _dt(K), self.d, self.n, self.n, _p(self.x),
                                    _p(self.x), g, _p(hd[0]), _p(hd[1]), float(self.jitter), _p(K),
                                    self.ld, self.n * self.ld), "gpvae_gram_se")
            # the draw masks the strict upper triangle itself: no zeroing / NaN-fill pass needed
            check(lib.gpvae_potrf(_stream(), _dt(K), self.n, g, _p(K), self.ld, self.n * self.ld,
                                  0, C.c_void_p(0), 0, 0, _p(self.info[slot]), 0, _p(self.pwork[slot])),
                  "gpvae_potrf")
            self.ev_ready[slot].record(stream)

    # -- stage B: draws + ELBO (+ update) on the current stream ----------------------------------------
    def _produce(self, key, slot, i, last_of_group, buf):
        """Normals and the B draws of one step into draw buffer `buf` (main stream)."""
        main = torch.cuda.current_stream()
        if self.overlap and i == 0:
            main.wait_event(self.ev_ready[slot])
        ops.randn(None, out=self.E, seed=self.seed, stream_id=philox_stream_id(self.rank, key, 1))
        ops.randn(None, out=self.E2, seed=self.seed, stream_id=philox_stream_id(self.rank, key, 2))
        ops.randn(None, out=self.ezbuf[buf], seed=self.seed, stream_id=philox_stream_id(self.rank, key, 3))
        K = self.Kbuf[slot][i]
        sig = self.hyper_dev[slot][2, i:i + 1]
        lib = _lib.load()
        if self.draw_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(main)
        check(lib.gpvae_draw(_stream(), _dt(K), self.n, self.B, _p(K), self.ld, _p(self.E), self.n,
                             _p(self.E2), self.n, _p(sig), None, 0, _p(self.Ybuf[buf]), self.n),
              "gpvae_draw")
        if self.draw_events is not None:
            ev[1].record(main)
            self.draw_events.append(ev)
        if self.overlap and last_of_group:
            self.ev_free[slot].record(main)

    def _learn_start(self, buf, state, train, t=0):
        """ELBO (+ gradient) of the step in draw buffer `buf`; starts the one collective of the path
        (flat-gradient SUM, RCCL over xGMI) and returns its handle."""
        self.loss = self.loss_steps[t:t + 1]  # one slot per step: summed once at the end of run()
        ops.vae_elbo(state.params, self.Ybuf[buf], self.ezbuf[buf], (self.h0, self.h1, self.z),
                     self.obs_sigma, grads=self.grads if train else None, loss=self.loss, work=self.work)
        return allreduce_sum_start(self.grads) if train else None

    def _learn_finish(self, handle, state, train):
        if train:
            allreduce_wait(handle)
            state.t += 1
            ops.adam_(state.params, self.grads, state.m, state.v, state.t, self.lr)

    def run(self, keys, state, train=True):
        """Run the steps named by `keys` (tuples of ints feeding the per-step RNG streams).
        Returns the device scalar holding the summed loss of this rank."""
        self.loss_sum.zero_()
        keys = [tuple(k) for k in keys]
        if not keys:
            return self.loss_sum
        self.loss_steps = torch.zeros(len(keys), dtype=self.dtype, device=self.x.device)
        groups = [keys[i:i + self.G] for i in range(0, len(keys), self.G)]
        if self.overlap:
            for s in range(2):
                self.ev_free[s].record(torch.cuda.current_stream())
        self._issue_factors(groups[0], 0)
        # software pipeline over steps: draw(t+1) does not depend on the parameters, so it is
        # enqueued between the start of step t's gradient all-reduce and the Adam step that waits
        # for it — the collective's latency hides behind the draw.  Same arithmetic in any order.
        pending = None  # (handle,) of the step whose Adam update is still to come
        t = 0
        for gi, grp in enumerate(groups):
            slot = gi & 1
            if gi + 1 < len(groups):
                self._issue_factors(groups[gi + 1], (gi + 1) & 1)
            for i, key in enumerate(grp):
                self._produce(key, slot, i, i == len(grp) - 1, t & 1)
                if pending is not None:
                    self._learn_finish(pending[0], state, train)
                pending = (self._learn_start(t & 1, state, train, t),)
                t += 1
        if pending is not None:
            self._learn_finish(pending[0], state, train)
        torch.sum(self.loss_steps, dim=0, keepdim=True, out=self.loss_sum)
        return self.loss_sum


class VAE():
    """2-hidden-layer VAE trained on GP draws (reference constructor: vae.py:39-52)."""

    def __init__(self, gp, hidden_dims, z_dim, out_dim, batch_size, learning_rate, num_epochs,
                 num_train, num_test, x, seed=0, dtype=torch.float32):
        self.gp = gp
        assert len(hidden_dims) == 2  # vae.py:55
        self.hidden_dims = hidden_dims
        self.z_dim = z_dim
        self.out_dim = out_dim
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs
        self.num_train = num_train
        self.num_test = num_test
        self.x = _as_device(x, dtype)
        self.seed = seed
        self.dtype = dtype
        self.gp_predictive = None
        self.engine = None
        self.test_loss_list = []

    # stax-style (init, apply) pair: apply(params_list, z) -> decoder output (inference.py:246)
    def vae_decoder(self):
        h0, h1 = self.hidden_dims

        def init_fun(rng_key, input_shape):
            flat = init_params(self.out_dim, h0, h1, self.z_dim, _seed_of(rng_key), self.dtype,
                               self.x.device)
            return (self.out_dim,), decoder_list_from_flat(flat, self.out_dim, h0, h1, self.z_dim)

        def apply_fun(params, inputs, **kwargs):
            flat = decoder_flat_from_list(params, self.dtype)
            zz = _as_device(inputs, self.dtype)
            single = zz.dim() == 1
            out = ops.decoder_fwd(flat, zz.reshape(-1, self.z_dim), self.out_dim, h0, h1)
            return out[0] if single else out

        return init_fun, apply_fun

    def _engine(self):
        if self.engine is None:
            h0, h1 = self.hidden_dims
            self.engine = StepEngine(self.x, self.gp.jitter, (h0, h1, self.z_dim), self.batch_size,
                                     self.learning_rate, self.dtype, self.seed)
        return self.engine

    def epoch_train(self, rng_key, svi_state):
        """num_train steps of draw-hyper -> GP batch -> svi.update (vae.py:139-169)."""
        k = _seed_of(rng_key) & 0x7FFFFFFF
        loss = self._engine().run([(k, i) for i in range(self.num_train)], svi_state, train=True)
        return loss, svi_state

    def eval_test(self, rng_key, svi_state):
        """Mean per-sample test loss over num_test fresh batches (vae.py:172-201)."""
        k = _seed_of(rng_key) & 0x7FFFFFFF
        loss = self._engine().run([(k, i) for i in range(self.num_test)], svi_state, train=False)
        if self._engine().world > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss = loss / self._engine().world
        return float(loss.item()) / self.batch_size / max(self.num_test, 1)

    def svi_init(self):
        """svi.init (vae.py:228): fresh parameters, zero Adam moments; identical on every rank."""
        h0, h1 = self.hidden_dims
        return SVIState(init_params(self.out_dim, h0, h1, self.z_dim, self.seed, self.dtype,
                                    self.x.device))

    def fit(self, plot_loss=False):
        """Train (vae.py:203-257).  Returns svi.get_params(state)['decoder$params']."""
        self.gp_predictive = Predictive(self.gp.sample, num_samples=self.batch_size)
        state = self.svi_init()
        rng = np.random.default_rng(self.seed)
        self.test_loss_list = []
        for i in range(self.num_epochs):
            k_train, k_test = (int(v) for v in rng.integers(0, 2 ** 31 - 1, size=2))
            t_start = time.time()
            _, state = self.epoch_train(k_train, state)
            test_loss = self.eval_test(k_test, state)
            self.test_loss_list += [test_loss]
            if self._engine().rank == 0:
                print("Epoch {}: loss = {} ({:.2f} s.)".format(i, test_loss, time.time() - t_start))
            if np.isnan(test_loss):  # vae.py:245
                break
        self.state = state
        h0, h1 = self.hidden_dims
        return decoder_list_from_flat(state.params, self.out_dim, h0, h1, self.z_dim)
