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
from .gp import PhiloxKey, Predictive, _as_device, _key_parts, _seed_of, draw_hyper, exp_kernel2
from .parallel import allreduce_sum_start, allreduce_wait, allreduce_sum_, dist_info, philox_stream_id, step_rng


class SVIState():
    """What numpyro's SVIState carries for this model: parameters, Adam moments, step count and the
    key (`seed`) the per-update latent normals are derived from."""

    def __init__(self, params, m=None, v=None, t=0, seed=0):
        self.params = params
        self.m = torch.zeros_like(params) if m is None else m
        self.v = torch.zeros_like(params) if v is None else v
        self.t = t
        self.seed = int(seed)


class Adam():
    """numpyro.optim.Adam(step_size, b1=0.9, b2=0.999, eps=1e-8) (vae.py:212): the hyper-parameters
    of gpvae_adam."""

    def __init__(self, step_size, b1=0.9, b2=0.999, eps=1e-8):
        self.step_size, self.b1, self.b2, self.eps = float(step_size), float(b1), float(b2), float(eps)


class Trace_ELBO():
    """numpyro.infer.Trace_ELBO() as the reference uses it (vae.py:217): one particle."""

    def __init__(self, num_particles=1):
        if num_particles != 1:
            raise NotImplementedError("gpvae_vae_elbo is the one-particle ELBO of vae.py:213-218")
        self.num_particles = 1


_SVI_STREAM = 0x5E1 << 44  # Philox stream base of SVI's own latent normals (update t -> base | t << 4)


class SVI():
    """Call-compatible stand-in for numpyro.infer.SVI(model, guide, optim, loss) at the reference's
    call sites (vae.py:213-218, 228, 165, 194, 257): `init(key, batch) -> state`,
    `update(state, batch) -> (state, loss)`, `evaluate(state, batch) -> loss`,
    `get_params(state)["decoder$params"]`.

    It is not a probabilistic-programming engine: `model` and `guide` must be the bound
    `vae_model` / `vae_guide` of one :class:`VAE` — the program whose ELBO, gradient and Adam step
    gpvae_vae_elbo / gpvae_adam compute.  Anything else raises.  `update` takes an EXTERNALLY supplied
    batch [B, n]; with torch.distributed initialised the flat gradient is summed over ranks first (the
    one collective of the path), so every rank applies the same step.  The state is updated in place
    and returned."""

    def __init__(self, model, guide, optim, loss, **static_kwargs):
        vae = getattr(model, "__self__", None)
        if (not isinstance(vae, VAE) or getattr(guide, "__self__", None) is not vae
                or getattr(model, "__func__", None) is not VAE.vae_model
                or getattr(guide, "__func__", None) is not VAE.vae_guide):
            raise NotImplementedError("SVI here runs VAE.vae_model / VAE.vae_guide of one VAE object (the "
                                      "ELBO gpvae_vae_elbo implements); other programs need NumPyro")
        if not isinstance(optim, Adam):
            raise NotImplementedError("SVI: the optimiser of this path is Adam (vae.py:212)")
        if not isinstance(loss, Trace_ELBO):
            raise NotImplementedError("SVI: the loss of this path is Trace_ELBO() (vae.py:217)")
        if static_kwargs:
            raise NotImplementedError("SVI: static model arguments are not part of this path")
        self.vae, self.optim, self.loss = vae, optim, loss
        self.model, self.guide = model, guide
        self._grads = None
        self._work = None

    def _dims(self):
        v = self.vae
        return v.hidden_dims[0], v.hidden_dims[1], v.z_dim

    def _batch(self, batch):
        v = self.vae
        Y = _as_device(batch, v.dtype)
        Y = Y.reshape(Y.shape[0], -1)  # vae.py:107,129
        if Y.shape[1] != v.out_dim:
            raise GpvaeError("SVI: batch has %d columns, the VAE's out_dim is %d" % (Y.shape[1], v.out_dim))
        return Y if Y.stride(1) == 1 else Y.contiguous()

    def init(self, rng_key, batch):
        """Fresh 1e-2*N(0,1) parameters (stax.randn, vae.py:76-95) keyed by rng_key, zero Adam moments.
        The batch only fixes the shapes, as in numpyro.module's initialisation."""
        self._batch(batch)
        v = self.vae
        seed = _seed_of(rng_key)
        h0, h1, z = self._dims()
        return SVIState(init_params(v.out_dim, h0, h1, z, seed, v.dtype, v.x.device), seed=seed)

    def _eps(self, state, B, rng_key):
        v = self.vae
        if rng_key is None:
            seed, base = state.seed, _SVI_STREAM | ((state.t + 1) << 4)
        else:
            seed, base = _key_parts(rng_key)
        return ops.randn((B, v.z_dim), v.dtype, v.x.device, seed=seed, stream_id=base | 3)

    def _elbo(self, state, batch, rng_key, train):
        Y = self._batch(batch)
        dims = self._dims()
        if self._work is None or self._work.key != (Y.shape[0], Y.shape[1]) + dims + (Y.dtype,):
            self._work = ops.VaeWorkspace(Y.shape[0], Y.shape[1], *dims, Y.dtype, Y.device)
        if train and (self._grads is None or self._grads.numel() != state.params.numel()):
            self._grads = torch.empty_like(state.params)
        eps = self._eps(state, Y.shape[0], rng_key)
        return ops.vae_elbo(state.params, Y, eps, dims, self.vae.obs_sigma,
                            grads=self._grads if train else None, work=self._work)

    def update(self, state, batch, rng_key=None):
        """One step: loss = -ELBO(batch) and its gradient, [all-reduce], Adam (vae.py:165).
        Returns (state, loss) with loss a device scalar [1].  rng_key (optional, a PhiloxKey) names the
        latent normals; by default they follow the state's own key and step count."""
        loss = self._elbo(state, batch, rng_key, True)
        allreduce_sum_(self._grads)
        state.t += 1
        o = self.optim
        ops.adam_(state.params, self._grads, state.m, state.v, state.t, o.step_size, o.b1, o.b2, o.eps)
        return state, loss

    def evaluate(self, state, batch, rng_key=None):
        """loss = -ELBO(batch) at the current parameters, no update (vae.py:194)."""
        return self._elbo(state, batch, rng_key, False)

    def get_params(self, state):
        """{'encoder$params', 'decoder$params'} in the stax list layouts (vae.py:257)."""
        v = self.vae
        h0, h1, z = self._dims()
        return {"encoder$params": encoder_list_from_flat(state.params, v.out_dim, h0, h1, z),
                "decoder$params": decoder_list_from_flat(state.params, v.out_dim, h0, h1, z)}


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


def encoder_list_from_flat(flat, n, h0, h1, z):
    """encoder$params as stax.serial(Dense, Relu, Dense, Relu, FanOut(2), parallel(Dense,
    serial(Dense, Exp))) nests them (vae.py:75-85)."""
    o = 0
    W1 = flat[o:o + n * h0].reshape(n, h0); o += n * h0
    b1 = flat[o:o + h0]; o += h0
    W2 = flat[o:o + h0 * h1].reshape(h0, h1); o += h0 * h1
    b2 = flat[o:o + h1]; o += h1
    Wm = flat[o:o + h1 * z].reshape(h1, z); o += h1 * z
    bm = flat[o:o + z]; o += z
    Ws = flat[o:o + h1 * z].reshape(h1, z); o += h1 * z
    bs = flat[o:o + z]
    return [(W1, b1), (), (W2, b2), (), (), [(Wm, bm), [(Ws, bs), ()]]]


def encoder_flat_from_list(params_list, dtype=None):
    (W1, b1), _, (W2, b2), _, _, ((Wm, bm), ((Ws, bs), _)) = params_list
    parts = [_as_device(a, dtype).reshape(-1) for a in (W1, b1, W2, b2, Wm, bm, Ws, bs)]
    return torch.cat(parts).contiguous()


def decoder_flat_from_list(params_list, dtype=None, device=None):
    (V1, c1), _, (V2, c2), _, (V3, c3) = params_list
    parts = [_as_device(a, dtype).reshape(-1) for a in (V1, c1, V2, c2, V3, c3)]
    return torch.cat(parts).contiguous()


def save_decoder_params(path, params_list, format="npz"):
    """Write the decoder parameters `VAE.fit()` returns — the stax list [(W,b),(),(W,b),(),(W,b)] the
    reference pickles after training (toy_examples.py:161-165, 2d_examples.py) — to `path`.

    format="npz" (default): six plain arrays V1,c1,V2,c2,V3,c3, readable with numpy.load(allow_pickle=
    False) — what load_decoder_params reads.  format="pickle": the reference's own file format (the
    same list with numpy arrays, pickle protocol 4), for handing a decoder trained here to the
    reference's scripts, whose `pickle.load` + `decoder(decoder_params, z)` accept it as is."""
    (V1, c1), _, (V2, c2), _, (V3, c3) = params_list
    arrs = [np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a))
            for a in (V1, c1, V2, c2, V3, c3)]
    if format == "npz":
        with open(path, "wb") as fh:
            np.savez(fh, V1=arrs[0], c1=arrs[1], V2=arrs[2], c2=arrs[3], V3=arrs[4], c3=arrs[5])
    elif format == "pickle":
        import pickle
        with open(path, "wb") as fh:
            pickle.dump([(arrs[0], arrs[1]), (), (arrs[2], arrs[3]), (), (arrs[4], arrs[5])], fh, protocol=4)
    else:
        raise ValueError("format must be 'npz' or 'pickle'")


def load_decoder_params(path, dtype=None, device=None):
    """Read a decoder written by save_decoder_params(format="npz") back as the stax list (tensors on
    `device`).  Pickle files are not read here — unpickling runs code from the file, and the reference's
    decoder files hold jax device arrays that only a full unpickler with jax installed can rebuild;
    convert them once, where they are trusted, with `numpy.savez` of the six arrays."""
    with open(path, "rb") as fh:
        if fh.read(4) != b"PK\x03\x04":
            raise GpvaeError("load_decoder_params reads the .npz files save_decoder_params writes; %r is not "
                             "one (a pickle is not loaded: convert it to six arrays V1,c1,V2,c2,V3,c3)" % (path,))
    with np.load(path, allow_pickle=False) as z:
        names = ("V1", "c1", "V2", "c2", "V3", "c3")
        missing = [k for k in names if k not in z.files]
        if missing:
            raise GpvaeError("decoder file %r lacks arrays %s" % (path, missing))
        a = [z[k] for k in names]
    if not (a[0].ndim == a[2].ndim == a[4].ndim == 2 and a[0].shape[1] == a[2].shape[0] == a[1].shape[0]
            and a[2].shape[1] == a[4].shape[0] == a[3].shape[0] and a[4].shape[1] == a[5].shape[0]):
        raise GpvaeError("decoder file %r: layer shapes do not chain: %s" % (path, [x.shape for x in a]))
    t = [_as_device(x, dtype) if device is None else torch.as_tensor(x, dtype=dtype, device=device) for x in a]
    return [(t[0], t[1]), (), (t[2], t[3]), (), (t[4], t[5])]


_SIDE_STREAMS = {}


def _side_streams(dev):
    """The factor stream and the normals stream of every StepEngine on `dev`: created once per process.
    HIP deals streams to GPU_MAX_HW_QUEUES hardware queues in creation order, and a side stream that lands on the
    main stream's queue serialises behind it: with a fresh pair per engine every SECOND engine of a process ran
    the C3 step in 0.90-0.96 instead of 0.81 ms (tools/soak_c3.py; 4 queues, the boxes' setting).  Engines of one
    process therefore share the first pair, whose placement is the one bench.py measures."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SIDE_STREAMS[key]


class StepEngine():
    """One training/evaluation step = gram -> potrf -> draw -> ELBO (-> all-reduce -> Adam).

    The factors of the next `group` steps are independent of the network, so they are built as ONE
    batched Gram + batched Cholesky launch sequence on a side stream (throughput mode of the panel
    sweep instead of `group` latency-bound single factorisations), one group ahead of the
    draw/ELBO/Adam chain; 2 x group factor slots live in HBM (group=16, n=4096, f32: 2 GiB; the default
    group is 16, less when a slot would exceed 8 GiB)."""

    def __init__(self, x, jitter, dims, batch_size, lr, dtype=torch.float32, seed=0, obs_sigma=0.1,
                 hyper_sampler=draw_hyper, overlap=True, group=None, kernel=exp_kernel2, mfma="auto",
                 first_group=None, group_normals=True):
        self.first_group = first_group  # int or sequence: sizes of the first factor groups; None: default_ramp
        self.default_ramp = ()
        if isinstance(hyper_sampler, str) and hyper_sampler != "device":
            raise ValueError("hyper_sampler: a callable rng -> (ls, var, sigma) or 'device'")
        if kernel is not exp_kernel2:
            # the engine calls gpvae_gram_se itself: another kernel would silently train on SE
            raise NotImplementedError("StepEngine builds its Gram matrices with gpvae_gram_se (exp_kernel2); "
                                      "use VAE(..., pipelined=False) to train through an injected kernel")
        # [n] or [n,d] row-major, as gpvae_gram_se reads it: a transposed / strided view (the 2-D
        # scripts' array([u,v]).transpose idiom, 2d_examples.py:564) is copied, not reinterpreted
        self.x = ops._points(_as_device(x, dtype))
        self.n = self.x.shape[0]
        self.d = self.x.shape[1]
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
        self.ld = (n + 3) // 4 * 4  # 16-byte aligned rows for the vector loads
        if group is None:
            # a factorisation call carries the latency of its dependent panel launches whatever its batch, so
            # full groups of 16 measured best at C3; larger problems take what 8 GiB per slot holds
            group = max(2, min(16, (8 << 30) // max(1, n * self.ld * self.x.element_size())))
        self.G = max(1, int(group))
        G = self.G
        self.Kbuf = [torch.empty((G, n, self.ld), dtype=dtype, device=dev) for _ in range(2)]
        self.hyper_dev = [torch.empty((3, G), dtype=dtype, device=dev) for _ in range(2)]  # var|ls|sigma
        self.hyper_pin = [torch.empty((3, G), dtype=dtype).pin_memory() for _ in range(2)]
        self.info = [torch.zeros(G, dtype=torch.int32, device=dev) for _ in range(2)]
        self.info_fail = torch.zeros(1, dtype=torch.int32, device=dev)  # factors reported not PD so far
        wbytes = _lib.load().gpvae_potrf_workspace_bytes(_dt(self.x), n, 0, G)
        self.pwork = [torch.empty(max(16, wbytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        # float32 draws run on the bf16 matrix pipe when the shapes allow ("bf16 x 3": exact 3-way bf16
        # split of both operands, six piece products, f32 accumulation — float32-accurate at 2.5x the f32
        # MFMA rate, csrc/b3.hpp): E is generated straight as bf16 planes, L is split on the factor stream.
        # mfma="f16x2" (opt-in): the draw on two fp16 planes per operand — half the matrix work of bf16 x 3; the
        # operands are rounded to 22 bits, measured errors are those of the f32-MFMA kernel or below
        # (include/gpvae.h, tests/test_gpu_seams.py::test_h2_planes_and_draw).  gpvae_potrf_h2 leaves the factor
        # as such planes and runs its trailing updates on them.
        if mfma not in ("auto", "f32", "bf16x3", "f16x2"):
            raise ValueError("mfma must be 'auto', 'f32', 'bf16x3' or 'f16x2'")
        self.b3 = (mfma != "f32" and dtype == torch.float32 and self.ld == n and ops.draw_b3_supported(n, B)
                   and ops.potrf_b3_supported(n))
        if mfma in ("bf16x3", "f16x2") and not self.b3:
            raise GpvaeError("%s draws need float32, n %% 256 == 0, n >= 512 and batch_size %% 128 == 0" % mfma)
        self.h2 = self.b3 and mfma == "f16x2"
        if self.h2:
            self.b3 = False
            self.Eh = torch.empty((2, B, n), dtype=torch.float16, device=dev)
            self.L2buf = [torch.empty((G, 2, n, n), dtype=torch.float16, device=dev) for _ in range(2)]
            self.L2inv = [torch.empty(G, dtype=torch.float32, device=dev) for _ in range(2)]
        if self.b3:
            self.E3 = torch.empty((3, B, n), dtype=torch.bfloat16, device=dev)
            self.L3buf = [torch.empty((G, 3, n, n), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        self.E = None if (self.b3 or self.h2) else torch.empty((B, n), dtype=dtype, device=dev)
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
        self.s_gp, self.s_rng = _side_streams(dev) if overlap else (None, None)
        # Normals of a whole GROUP of steps at a time (round 3).  Generated per step on the side stream, the normals
        # cost the main stream two cross-stream event operations per step (a record after the draw, a wait before the
        # next one), and the kernel trace shows an 11-12 us hole in the main stream at each of them: 23 us of a 0.6 ms
        # step.  They depend on nothing but (seed, rank, key), so the side stream now writes the normals of all the
        # steps of a group into per-slot buffers as soon as the slot is free — ONE wait per group on the main stream,
        # beside the one for the factors.  Costs G x (E planes + E' + eps_z) per slot (C3: 2 x 2.5 GiB of 288);
        # engines whose buffers would exceed 16 GiB keep the per-step scheme.  Same Philox streams, same values.
        es_ = self.x.element_size()
        per_step = B * n * ((6 if self.b3 else 4 if self.h2 else es_) + es_) + B * self.z * es_
        self.gn = bool(overlap and group_normals and 2 * G * per_step <= (16 << 30)
                       and os.environ.get("GPVAE_GROUP_NORMALS", "1") != "0")  # the variable: A/B measurements
        if self.gn:
            if self.b3:
                self.E3g = [torch.empty((G, 3, B, n), dtype=torch.bfloat16, device=dev) for _ in range(2)]
            elif self.h2:
                self.Ehg = [torch.empty((G, 2, B, n), dtype=torch.float16, device=dev) for _ in range(2)]
            else:
                self.Eg = [torch.empty((G, B, n), dtype=dtype, device=dev) for _ in range(2)]
            self.E2g = [torch.empty((G, B, n), dtype=dtype, device=dev) for _ in range(2)]
            self.ezg = [torch.empty((G, B, self.z), dtype=dtype, device=dev) for _ in range(2)]
            self.ev_rng_ready = [torch.cuda.Event() for _ in range(2)]
        self.ev_drawn, self.ev_rng = torch.cuda.Event(), torch.cuda.Event()
        self._rng_pending = False
        self.ev_ready = [torch.cuda.Event() for _ in range(2)]   # factors of group slot s are ready
        self.ev_free = [torch.cuda.Event() for _ in range(2)]    # group slot s has been consumed
        self.ev_copied = [torch.cuda.Event() for _ in range(2)]  # slot s's hyper upload is done
        self.draw_events = None  # bench.py: list of (start, end) HIP events around gpvae_draw
        # ... around every draw_events_every-th draw: an event record costs the stream ~5.6 us (kernel trace: a hole
        # of that size on either side of every bracketed draw, 11 us of a 0.55 ms step when every launch is timed)
        self.draw_events_every = 1
        self._draw_calls = 0
        # The small configurations (C1, C2) are bound by the host's launch rate, so the per-step path below
        # makes no tensor views and no wrapper calls: raw device pointers are taken once here, the library is
        # called with explicit stream handles (no stream context switches), six ABI calls per step
        self._lib = _lib.load()
        self._dtc = _dt(self.x)
        es = self.x.element_size()
        vp = C.c_void_p
        self._pK = [[vp(self.Kbuf[s].data_ptr() + i * n * self.ld * es) for i in range(G)] for s in range(2)]
        self._psig = [[vp(self.hyper_dev[s].data_ptr() + (2 * G + i) * es) for i in range(G)] for s in range(2)]
        if self.b3:
            self._pL3 = [[vp(self.L3buf[s].data_ptr() + i * 3 * n * n * 2) for i in range(G)] for s in range(2)]
            self._pE3 = vp(self.E3.data_ptr())
        elif self.h2:
            self._pL2 = [[vp(self.L2buf[s].data_ptr() + i * 2 * n * n * 2) for i in range(G)] for s in range(2)]
            self._pL2inv = [[vp(self.L2inv[s].data_ptr() + i * 4) for i in range(G)] for s in range(2)]
            self._pEh = vp(self.Eh.data_ptr())
        else:
            self._pE = vp(self.E.data_ptr())
        self._pE2 = vp(self.E2.data_ptr())
        if self.gn:
            Eg = self.E3g if self.b3 else (self.Ehg if self.h2 else self.Eg)
            Ebytes = B * n * (6 if self.b3 else 4 if self.h2 else es)
            self._pEg = [[vp(Eg[s].data_ptr() + i * Ebytes) for i in range(G)] for s in range(2)]
            self._pE2g = [[vp(self.E2g[s].data_ptr() + i * B * n * es) for i in range(G)] for s in range(2)]
            self._pezg = [[vp(self.ezg[s].data_ptr() + i * B * self.z * es) for i in range(G)] for s in range(2)]
        self._cur_ez = [None, None]  # eps_z of the step in draw buffer 0 / 1
        self._gn_run = False         # run() sets it: this run takes its normals from the group buffers
        self.gn_at = 8               # ... issued this many sixteenths of a group after the group before it starts
        self._pY = [vp(t.data_ptr()) for t in self.Ybuf]
        self._pez = [vp(t.data_ptr()) for t in self.ezbuf]
        self._pgrads = vp(self.grads.data_ptr())
        self._pwork = vp(self.work.buf.data_ptr())
        self._h_rng = vp(self.s_rng.cuda_stream) if overlap else None
        self._main = torch.cuda.current_stream(dev)  # run() re-reads the current stream
        self._h_main = vp(self._main.cuda_stream)

    # -- stage A: the factors of one group of steps (own stream) ---------------------------------------
    def _issue_factors(self, keys, slot, share_chip=False):
        """share_chip: the group is factored beside the steps of the group before it (GPVAE_POTRF_SHARE_CHIP = 4)."""
        g = len(keys)
        pflags = 4 if share_chip else 0
        device_hyper = isinstance(self.hyper_sampler, str)
        if not device_hyper:
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
            if device_hyper:
                # hyper_sampler="device": (ls, var, sigma) of step (k, i) is sample i of the Philox stream
                # of (rank, k) — drawn by gpvae_hyper_sample on the factor stream, no host round trip
                runs, lo = [], 0
                for i in range(1, g + 1):  # runs of consecutive steps of one epoch key: one launch each
                    if i == g or keys[i][0] != keys[lo][0] or keys[i][-1] != keys[i - 1][-1] + 1:
                        runs.append((lo, i))
                        lo = i
                for lo, hi in runs:
                    cnt = self.G - lo if hi == g else hi - lo  # the tail also fills a short group's unused slots
                    ops.hyper_sample(cnt, seed=self.seed, stream_id=philox_stream_id(self.rank, (keys[lo][0], 0), 5),
                                     first_index=keys[lo][-1], out=hd[:, lo:lo + cnt])
            else:
                hd.copy_(pin, non_blocking=True)
                self.ev_copied[slot].record(stream)
            K = self.Kbuf[slot]
            lib = _lib.load()
            check(lib.gpvae_gram_se(_stream(), _dt(K), self.d, self.n, self.n, _p(self.x),
                                    _p(self.x), g, _p(hd[0]), _p(hd[1]), float(self.jitter), _p(K),
                                    self.ld, self.n * self.ld), "gpvae_gram_se")
            # the draw masks the strict upper triangle itself: no zeroing / NaN-fill pass needed
            if self.b3:  # factor + its bf16 planes in one sweep (the panel launches write the planes)
                check(lib.gpvae_potrf_b3(_stream(), self.n, g, _p(K), self.ld, self.n * self.ld, _p(self.info[slot]),
                                         pflags, _p(self.pwork[slot]), _p(self.L3buf[slot]), self.n, self.n * self.n,
                                         3 * self.n * self.n), "gpvae_potrf_b3")
            elif self.h2:  # the same with fp16 x 2 planes of L * 2^e, e from the largest diagonal entry
                check(lib.gpvae_potrf_h2(_stream(), self.n, g, _p(K), self.ld, self.n * self.ld, _p(self.info[slot]),
                                         pflags, _p(self.pwork[slot]), _p(self.L2buf[slot]), self.n, self.n * self.n,
                                         2 * self.n * self.n, _p(self.L2inv[slot])), "gpvae_potrf_h2")
            else:
                check(lib.gpvae_potrf(_stream(), _dt(K), self.n, g, _p(K), self.ld, self.n * self.ld,
                                      0, C.c_void_p(0), 0, 0, _p(self.info[slot]), pflags, _p(self.pwork[slot])),
                      "gpvae_potrf")
            self.info_fail.add_(self.info[slot][:g].ne(0).sum().to(torch.int32))
            self.ev_ready[slot].record(stream)

    # -- stage B: draws + ELBO (+ update) on the current stream ----------------------------------------
    def _normals(self, key, buf, h, dst=None):
        """The step's standard normals (Philox streams keyed by rank, key, purpose — parallel.philox_stream_id):
        E (as bf16 planes on the bf16 x 3 path), E', eps_z into draw buffer `buf` — or into dst = (E, E', eps_z)
        pointers of a group slot —, on the stream with handle h."""
        lib, n, B, seed = self._lib, self.n, self.B, self.seed
        sid = ((self.rank & 0xFF) << 56) | ((int(key[0]) & 0xFFFFFF) << 28) | ((int(key[-1]) & 0xFFFFFF) << 4)
        pE, pE2, pez = dst if dst is not None else (
            self._pE3 if self.b3 else (self._pEh if self.h2 else self._pE), self._pE2, self._pez[buf])
        if self.b3:
            rc = lib.gpvae_randn_b3(h, B * n, seed, sid | 1, pE, B * n)
        elif self.h2:
            rc = lib.gpvae_randn_h2(h, B * n, seed, sid | 1, pE, B * n)
        else:
            rc = lib.gpvae_randn(h, self._dtc, B * n, seed, sid | 1, pE)
        rc = rc or lib.gpvae_randn(h, self._dtc, B * n, seed, sid | 2, pE2)
        rc = rc or lib.gpvae_randn(h, self._dtc, B * self.z, seed, sid | 3, pez)
        if rc:
            check(rc, "gpvae_randn")

    def _issue_normals(self, keys, slot):
        """Group mode: the normals of every step of the group into slot `slot`, on the normals stream."""
        self.s_rng.wait_event(self.ev_free[slot])  # the steps that read this slot's buffers are done
        for i, key in enumerate(keys):
            self._normals(key, 0, self._h_rng, (self._pEg[slot][i], self._pE2g[slot][i], self._pezg[slot][i]))
        self.ev_rng_ready[slot].record(self.s_rng)

    def _produce(self, key, slot, i, last_of_group, buf, next_key=None):
        """The B draws of one step into draw buffer `buf` (main stream).  The normals of the NEXT step are
        generated on a side stream as soon as this draw has read E / E' — beside the ELBO chain of this
        step, whose small latency-bound kernels leave most CUs idle (they were 36 us of the main
        stream's critical path)."""
        main, hm = self._main, self._h_main
        if self.overlap and i == 0:
            main.wait_event(self.ev_ready[slot])
        gn = self.gn and self._gn_run
        if gn:
            if self.overlap and i == 0:
                main.wait_event(self.ev_rng_ready[slot])  # the whole group's normals
            pE, pE2 = self._pEg[slot][i], self._pE2g[slot][i]
            self._cur_ez[buf] = self._pezg[slot][i]
        else:
            pE, pE2 = (self._pE3 if self.b3 else (self._pEh if self.h2 else self._pE)), self._pE2
            self._cur_ez[buf] = self._pez[buf]
            if self._rng_pending:
                main.wait_event(self.ev_rng)      # this step's normals came from the side stream
                self._rng_pending = False
            else:
                self._normals(key, buf, hm)
        lib, n = self._lib, self.n
        ev = None
        if self.draw_events is not None:
            if self._draw_calls % self.draw_events_every == 0:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(main)
            self._draw_calls += 1
        if self.b3:
            rc = lib.gpvae_draw_b3(hm, n, self.B, self._pL3[slot][i], n, n * n, pE, n, self.B * n,
                                   pE2, n, self._psig[slot][i], None, 0, self._pY[buf], n)
        elif self.h2:
            rc = lib.gpvae_draw_h2(hm, n, self.B, self._pL2[slot][i], n, n * n, self._pL2inv[slot][i], pE, n,
                                   self.B * n, pE2, n, self._psig[slot][i], None, 0, self._pY[buf], n)
        else:
            rc = lib.gpvae_draw(hm, self._dtc, n, self.B, self._pK[slot][i], self.ld, pE, n,
                                pE2, n, self._psig[slot][i], None, 0, self._pY[buf], n)
        if rc:
            check(rc, "gpvae_draw")
        if ev is not None:
            ev[1].record(main)
            self.draw_events.append(ev)
        if gn:
            return  # the slot is released after the group's last ELBO (run): it also reads the slot's eps_z
        if self.overlap and next_key is not None:
            self.ev_drawn.record(main)
            self.s_rng.wait_event(self.ev_drawn)
            self._normals(next_key, buf ^ 1, self._h_rng)
            self.ev_rng.record(self.s_rng)
            self._rng_pending = True
        if self.overlap and last_of_group:
            self.ev_free[slot].record(main)

    def _learn_start(self, buf, state, train, t=0):
        """ELBO (+ gradient) of the step in draw buffer `buf`; starts the one collective of the path
        (flat-gradient SUM, RCCL over xGMI) and returns its handle."""
        rc = self._lib.gpvae_vae_elbo(self._h_main, self._dtc, self.B, self.n, self.h0, self.h1, self.z, self._pparams,
                                      self._pY[buf], self.n, self._cur_ez[buf], float(self.obs_sigma),
                                      C.c_void_p(self._loss_base + t * self._loss_es),
                                      self._pgrads if train else None, self._pwork)
        if rc:
            check(rc, "gpvae_vae_elbo")
        return allreduce_sum_start(self.grads) if (train and self._collective) else None

    def _learn_finish(self, handle, state, train):
        if train:
            if handle is not None:
                allreduce_wait(handle)
            state.t += 1
            rc = self._lib.gpvae_adam(self._h_main, self._dtc, self._nparams, self._pparams, self._pgrads, self._pm,
                                      self._pv, int(state.t), float(self.lr), 0.9, 0.999, 1e-8)
            if rc:
                check(rc, "gpvae_adam")

    def time_draw_alone(self, reps=16):
        """Average duration (ms, HIP events) of the step's draw launched back to back with nothing beside it on
        the chip — the kernel's own time, as a kernel trace reports it; bench.py quotes it next to the in-loop
        figure, whose events also see the wait for CUs the factor stream holds."""
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        self._main, self._h_main = main, C.c_void_p(main.cuda_stream)
        keep, self.draw_events = self.draw_events, []
        was, self.overlap = self.overlap, False
        was_gn = self._gn_run  # (a run in group mode left real normals in slot 0's buffers: the draw reads those)
        try:
            for r in range(reps + 2):
                self._rng_pending = True  # keep the normals already in the buffers: time the draw only
                self._produce((0, 0), 0, 0, False, 0)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self.draw_events[2:]]
        finally:
            self.overlap, self.draw_events, self._rng_pending, self._gn_run = was, keep, False, was_gn
        return sum(ms) / len(ms)

    def failed_factors(self):
        """Number of Gram matrices gpvae_potrf reported as not positive definite since the engine was
        built (synchronises).  Such a step trains on a NaN batch, as the reference's does (SURVEY.md
        §2.1-11); fit() reports it with the epoch instead of leaving a bare NaN loss."""
        if self.overlap:
            self.s_gp.synchronize()
        return int(self.info_fail.item())

    def run(self, keys, state, train=True):
        """Run the steps named by `keys` (tuples of ints feeding the per-step RNG streams).
        Returns the device scalar holding the summed loss of this rank."""
        self.loss_sum.zero_()
        keys = [tuple(k) for k in keys]
        if not keys:
            return self.loss_sum
        self.loss_steps = torch.zeros(len(keys), dtype=self.dtype, device=self.x.device)
        self.loss = self.loss_steps[-1:]  # the last step's loss
        # per-run raw handles for the step path (the current stream is the main stream of this run)
        self._main = torch.cuda.current_stream()
        self._h_main = C.c_void_p(self._main.cuda_stream)
        self._loss_base, self._loss_es = self.loss_steps.data_ptr(), self.loss_steps.element_size()
        if state.params.device != self.x.device or state.params.dtype != self.dtype or not state.params.is_contiguous():
            raise GpvaeError("StepEngine.run: state.params must be a contiguous %s tensor on %s" % (self.dtype, self.x.device))
        self._pparams, self._pm, self._pv = (C.c_void_p(a.data_ptr()) for a in (state.params, state.m, state.v))
        self._nparams = state.params.numel()
        from .parallel import _collective_on
        self._collective = _collective_on()
        # the first group is exposed (nothing to overlap it with).  Measured on 20-step runs at C3 (ms per
        # step): a full group of 16: 0.98; first group of 2 / 4 / 8: 1.08 / 1.06 / 1.05; ramps 4,4,8: 1.16,
        # 4,8,8: 1.11, 8,8: 1.08, 2,4,8,16: 1.20; one group of 20: 1.01 — every factorisation call carries the
        # latency of its 64 dependent panel launches whatever its batch, so fewer, fuller groups win;
        # `first_group` (an int or a sequence of sizes) overrides
        sched = self.first_group
        if sched is None:
            sched = self.default_ramp if len(keys) > self.G else ()
        elif isinstance(sched, int):
            sched = (sched,)
        groups, pos = [], 0
        for g in sched:  # the sizes of the first groups (a ramp), then full groups
            g = max(1, min(int(g), self.G))
            groups.append(keys[pos:pos + g])
            pos += g
            if pos >= len(keys):
                break
        groups += [keys[i:i + self.G] for i in range(pos, len(keys), self.G)]
        groups = [g for g in groups if g]
        if self.overlap:
            for s in range(2):
                self.ev_free[s].record(torch.cuda.current_stream())
        self._gn_run = self.gn and self.overlap
        self._rng_pending = False
        self._issue_factors(groups[0], 0)
        if self._gn_run:
            self._issue_normals(groups[0], 0)
        # software pipeline over steps: draw(t+1) does not depend on the parameters, so it is
        # enqueued between the start of step t's gradient all-reduce and the Adam step that waits
        # for it — the collective's latency hides behind the draw.  Same arithmetic in any order.
        pending = None  # (handle,) of the step whose Adam update is still to come
        t = 0
        for gi, grp in enumerate(groups):
            slot = gi & 1
            if gi + 1 < len(groups):  # factored beside this group's steps: the first group of a run has the chip to itself
                self._issue_factors(groups[gi + 1], (gi + 1) & 1, share_chip=self.overlap)
            for i, key in enumerate(grp):
                # the next group's normals go out half a group after its factors, so that the two bursts of
                # side-stream work do not fall on the same steps (measured: no difference to the step, 0.768-0.774 ms
                # for 0 / 4 / 8 / 12 sixteenths; kept for the shorter queue on the normals stream)
                if self._gn_run and gi + 1 < len(groups) and i == self.gn_at * len(grp) // 16:
                    self._issue_normals(groups[gi + 1], (gi + 1) & 1)
                nxt = grp[i + 1] if i + 1 < len(grp) else (groups[gi + 1][0] if gi + 1 < len(groups) else None)
                self._produce(key, slot, i, i == len(grp) - 1, t & 1, nxt)
                if pending is not None:
                    self._learn_finish(pending[0], state, train)
                pending = (self._learn_start(t & 1, state, train, t),)
                if self._gn_run and i == len(grp) - 1:
                    self.ev_free[slot].record(self._main)  # factors, normals and eps_z of the slot have been read
                t += 1
        if pending is not None:
            self._learn_finish(pending[0], state, train)
        torch.sum(self.loss_steps, dim=0, keepdim=True, out=self.loss_sum)
        return self.loss_sum


class VAE():
    """2-hidden-layer VAE trained on GP draws (reference constructor: vae.py:39-52).

    `pipelined=True` (default) runs epoch_train / eval_test on :class:`StepEngine` (factor stream one
    group ahead, batched factors); `pipelined=False` runs the reference's literal loop
    draw hyper -> gp_predictive -> svi.update.  Same keys, same arithmetic, same results."""

    obs_sigma = 0.1  # vae.py:117

    def __init__(self, gp, hidden_dims, z_dim, out_dim, batch_size, learning_rate, num_epochs,
                 num_train, num_test, x, seed=0, dtype=torch.float32, pipelined=True, mfma="auto",
                 hyper_sampler="device"):
        self.gp = gp
        self.mfma = mfma  # StepEngine's matrix-pipe route for float32 draws: "auto" | "f32" | "bf16x3" | "f16x2"
        # where each step's (ls, var, sigma) come from (vae.py:157-159 priors): "device" — gpvae_hyper_sample on the
        # factor stream, the configuration bench.py times — or a callable rng -> (ls, var, sigma) run on the host
        # (gp.draw_hyper: NumPy's generators).  The pipelined engine and the literal loop draw the same values.
        if isinstance(hyper_sampler, str) and hyper_sampler != "device":
            raise ValueError("hyper_sampler: a callable rng -> (ls, var, sigma) or 'device'")
        self.hyper_sampler = hyper_sampler
        assert len(hidden_dims) == 2  # vae.py:55
        self.hidden_dims = hidden_dims
        self.z_dim = z_dim
        self.out_dim = out_dim
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs
        self.num_train = num_train
        self.num_test = num_test
        x = _as_device(x, dtype)
        self.x = x if x.is_contiguous() else x.contiguous()
        self.seed = seed
        self.dtype = dtype
        self.pipelined = bool(pipelined)
        self.gp_predictive = None
        self.svi = None
        self.engine = None
        self.state = None
        self.test_loss_list = []

    def _flat_dims(self):
        return self.hidden_dims[0], self.hidden_dims[1], self.z_dim

    # stax-style (init, apply) pairs -------------------------------------------------------------------
    def vae_encoder(self):
        """(init_fun, apply_fun) of the encoder (vae.py:72-85): apply(params, batch) -> (z_loc, z_std)."""
        h0, h1, z = self._flat_dims()

        def init_fun(rng_key, input_shape):
            flat = init_params(self.out_dim, h0, h1, z, _seed_of(rng_key), self.dtype, self.x.device)
            return ((z,), (z,)), encoder_list_from_flat(flat, self.out_dim, h0, h1, z)

        def apply_fun(params, inputs, **kwargs):
            flat = encoder_flat_from_list(params, self.dtype)
            Y = _as_device(inputs, self.dtype)
            single = Y.dim() == 1
            mu, sd = ops.encoder_fwd(flat, Y.reshape(-1, self.out_dim).contiguous(), (h0, h1, z))
            return (mu[0], sd[0]) if single else (mu, sd)

        return init_fun, apply_fun

    def vae_decoder(self):
        """(init_fun, apply_fun) of the decoder (vae.py:87-96): apply(params, z) -> f (inference.py:246)."""
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

    def _params(self, params):
        if params is not None:
            return params
        if self.state is None:
            raise GpvaeError("vae_model / vae_guide outside SVI need parameters: pass params= or call "
                             "fit() / svi.init() first")
        return self.state.params

    def vae_guide(self, batch, params=None, rng_key=None):
        """Inference with the encoder (vae.py:120-137): z ~ Normal(z_loc, z_std), one row per sample.
        `params`: flat parameter vector (default: the current SVI state's)."""
        Y = _as_device(batch, self.dtype)
        Y = Y.reshape(Y.shape[0], -1).contiguous()
        seed, base = _key_parts(rng_key)
        eps = ops.randn((Y.shape[0], self.z_dim), self.dtype, Y.device, seed=seed, stream_id=base | 3)
        return ops.encoder_fwd(self._params(params), Y, self._flat_dims(), eps)[2]

    def vae_model(self, batch, params=None, z=None, rng_key=None):
        """Generation with the decoder (vae.py:98-117): z ~ N(0, I) (or the guide's z when replayed),
        f = decoder(z), obs ~ Normal(f, 0.1) observed at `batch` — returns the observed site's value,
        i.e. the batch, as numpyro.sample(..., obs=batch) does; f is kept in `self.last_f`."""
        Y = _as_device(batch, self.dtype)
        Y = Y.reshape(Y.shape[0], -1)
        if z is None:
            seed, base = _key_parts(rng_key)
            z = ops.randn((Y.shape[0], self.z_dim), self.dtype, Y.device, seed=seed, stream_id=base | 4)
        h0, h1, zd = self._flat_dims()
        flat = self._params(params)
        dec = flat[flat.numel() - (zd * h1 + h1 + h1 * h0 + h0 + h0 * self.out_dim + self.out_dim):]
        self.last_f = ops.decoder_fwd(dec.contiguous(), _as_device(z, self.dtype).reshape(-1, zd),
                                      self.out_dim, h0, h1)
        return Y

    # ---- training -------------------------------------------------------------------------------------
    def _engine(self):
        if self.engine is None:
            h0, h1 = self.hidden_dims
            self.engine = StepEngine(self.x, self.gp.jitter, (h0, h1, self.z_dim), self.batch_size,
                                     self.learning_rate, self.dtype, self.seed, obs_sigma=self.obs_sigma,
                                     kernel=self.gp.kernel, mfma=self.mfma, hyper_sampler=self.hyper_sampler)
        return self.engine

    def _ensure_svi(self):
        if self.svi is None:
            self.svi = SVI(self.vae_model, self.vae_guide, Adam(self.learning_rate), Trace_ELBO())
        if self.gp_predictive is None:
            self.gp_predictive = Predictive(self.gp.sample, num_samples=self.batch_size)

    def _literal_steps(self, keys, svi_state, train):
        """The reference's loop body (vae.py:150-167 / 182-196), one step per key: draw
        (ls, var, sigma), batch = gp_predictive(key, x, ls, var, sigma)['y'], svi.update / evaluate.
        Keys, host RNG and Philox streams are the ones StepEngine uses."""
        self._ensure_svi()
        rank, _ = dist_info()
        total = torch.zeros(1, dtype=self.dtype, device=self.x.device)
        for key in keys:
            if isinstance(self.hyper_sampler, str):  # "device": sample key[-1] of the (rank, epoch key) Philox stream
                ls, var, sigma = (float(t.item()) for t in ops.hyper_sample(
                    1, self.dtype, self.x.device, seed=self.seed, stream_id=philox_stream_id(rank, (key[0], 0), 5),
                    first_index=key[-1]))
            else:
                ls, var, sigma = self.hyper_sampler(step_rng(self.seed, rank, key))
            pk = PhiloxKey(self.seed, philox_stream_id(rank, key, 0))
            batch = self.gp_predictive(pk, self.x, ls=float(ls), var=float(var), sigma=float(sigma))
            if train:
                svi_state, loss = self.svi.update(svi_state, batch["y"], rng_key=pk)
            else:
                loss = self.svi.evaluate(svi_state, batch["y"], rng_key=pk)
            total += loss
        return total, svi_state

    def _steps(self, keys, svi_state, train):
        if self.pipelined:
            return self._engine().run(keys, svi_state, train=train), svi_state
        return self._literal_steps(keys, svi_state, train)

    def epoch_train(self, rng_key, svi_state):
        """num_train steps of draw-hyper -> GP batch -> svi.update (vae.py:139-169)."""
        k = _seed_of(rng_key) & 0x7FFFFFFF
        return self._steps([(k, i) for i in range(self.num_train)], svi_state, True)

    def eval_test(self, rng_key, svi_state):
        """Mean per-sample test loss over num_test fresh batches (vae.py:172-201)."""
        k = _seed_of(rng_key) & 0x7FFFFFFF
        loss, _ = self._steps([(k, i) for i in range(self.num_test)], svi_state, False)
        _, world = dist_info()
        if world > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss = loss / world
        return float(loss.item()) / self.batch_size / max(self.num_test, 1)

    def svi_init(self):
        """svi.init (vae.py:228) with the VAE's own seed; identical on every rank."""
        self._ensure_svi()
        shape_only = torch.empty((self.batch_size, self.out_dim), dtype=self.dtype, device=self.x.device)
        return self.svi.init(self.seed, shape_only)

    def fit(self, plot_loss=False):
        """Train (vae.py:203-257).  Returns svi.get_params(state)['decoder$params']."""
        self.svi = SVI(self.vae_model, self.vae_guide, Adam(self.learning_rate), Trace_ELBO())
        self.gp_predictive = Predictive(self.gp.sample, num_samples=self.batch_size)
        n = self.x.shape[0]
        if self.batch_size * n * n * self.x.element_size() <= 2 ** 30:
            # initialise with a sample batch (vae.py:226): batch_size GPs with their own hyper-parameters
            sample_batch = self.gp_predictive(rng_key=PhiloxKey(self.seed, 0x1A17 << 4), x=self.x)["y"]
        else:
            # svi.init reads only the batch's shape; batch_size distinct n x n factors are skipped when
            # they would not fit a GiB
            sample_batch = torch.empty((self.batch_size, self.out_dim), dtype=self.dtype, device=self.x.device)
        state = self.svi.init(self.seed, sample_batch)
        self.state = state
        rng = np.random.default_rng(self.seed)
        self.test_loss_list = []
        rank, _ = dist_info()
        for i in range(self.num_epochs):
            k_train, k_test = (int(v) for v in rng.integers(0, 2 ** 31 - 1, size=2))
            t_start = time.time()
            _, state = self.epoch_train(k_train, state)
            test_loss = self.eval_test(k_test, state)
            self.test_loss_list += [test_loss]
            if rank == 0:
                print("Epoch {}: loss = {} ({:.2f} s.)".format(i, test_loss, time.time() - t_start))
            if np.isnan(test_loss):  # vae.py:245
                bad = self.engine.failed_factors() if self.engine is not None else 0
                if rank == 0 and bad:
                    print("gp_vae_amd: %d Gram matrices were not positive definite up to epoch %d (jitter %g)"
                          % (bad, i, self.gp.jitter))
                break
        self.state = state
        return self.svi.get_params(state)["decoder$params"]
