"""Thin torch-tensor front end of the C ABI (include/gpvae.h).

torch is used for device memory and streams only: every function here passes raw device pointers
of caller-visible tensors to libgpvae_hip.so on torch's current HIP stream.  No arithmetic happens
in torch and nothing falls back to the CPU: a CPU tensor raises.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import F32, F64, GpvaeError, POTRF_NAN_ON_FAIL, POTRF_SHARE_CHIP, POTRF_ZERO_UPPER, check

_DT = {torch.float32: F32, torch.float64: F64}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise GpvaeError("unsupported dtype %s (float32/float64 only)" % t.dtype)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise GpvaeError("libgpvae_hip needs device (HIP) tensors; got a %s tensor" % t.device)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def scalar(v, like=None, dtype=None, device=None):
    """Device scalar (shape [1]) for the ABI's `const void *var` style arguments."""
    dtype = dtype or like.dtype
    device = device or like.device
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=dtype).reshape(-1).contiguous()
    return torch.tensor([float(v)], dtype=dtype, device=device)


def _points(x):
    """check_d of the reference (gp.py:30-41): a 1-D array is a column of 1-D locations."""
    if x.dim() == 1:
        x = x.reshape(-1, 1)
    if x.dim() != 2 or x.shape[1] not in (1, 2):
        raise GpvaeError("locations must be [n] or [n,d] with d in {1,2}; got %s" % (tuple(x.shape),))
    return x.contiguous()


def gram_se(x, z, var, ls, jitter=1.0e-5, out=None):
    """K[b] = var[b]*exp(-0.5*(|x_i-z_j|/ls[b])^2) (+jitter*I iff square).  var/ls: scalars or [batch].

    Returns [n_x, n_z] for scalar hyper-parameters, [batch, n_x, n_z] for vectors."""
    _dev(x, z)
    if x.dim() != z.dim():
        raise GpvaeError("x and z must have the same number of dimensions (gp.py:34)")
    x, z = _points(x), _points(z)
    if x.shape[1] != z.shape[1] or x.dtype != z.dtype:
        raise GpvaeError("x and z disagree in d or dtype")
    batched = isinstance(var, torch.Tensor) and var.numel() > 1 or \
        isinstance(ls, torch.Tensor) and ls.numel() > 1
    var_t, ls_t = scalar(var, x), scalar(ls, x)
    batch = max(var_t.numel(), ls_t.numel())
    if var_t.numel() != batch:
        var_t = var_t.expand(batch).contiguous()
    if ls_t.numel() != batch:
        ls_t = ls_t.expand(batch).contiguous()
    n_x, n_z = x.shape[0], z.shape[0]
    if out is None:
        out = torch.empty((batch, n_x, n_z), dtype=x.dtype, device=x.device)
    K = out.view(batch, n_x, n_z)
    lib = _lib.load()
    check(lib.gpvae_gram_se(_stream(), _dt(x), x.shape[1], n_x, n_z, _p(x), _p(z), batch, _p(var_t),
                            _p(ls_t), float(jitter), _p(K), K.stride(1), K.stride(0)),
          "gpvae_gram_se")
    return K if batched else K[0]


def gram_agg(pts, var, ls, jitter=1.0e-6):
    """Aggregated kernel from sample points pts [n_cells, m, d] (agg_kernel_grid, gp.py:119-127)."""
    _dev(pts)
    if pts.dim() != 3 or pts.shape[2] not in (1, 2):
        raise GpvaeError("pts must be [n_cells, m, d]")
    pts = pts.contiguous()
    n_cells, m, d = pts.shape
    K = torch.empty((n_cells, n_cells), dtype=pts.dtype, device=pts.device)
    lib = _lib.load()
    var_t, ls_t = scalar(var, pts), scalar(ls, pts)  # keep alive until the launch is enqueued
    check(lib.gpvae_gram_agg(_stream(), _dt(pts), d, n_cells, m, _p(pts), _p(var_t), _p(ls_t),
                             float(jitter), _p(K), K.stride(0)), "gpvae_gram_agg")
    return K


def potrf_(A, X=None, zero_upper=True, nan_on_fail=True, work=None, share_chip=False):
    """In-place lower Cholesky of A [n,n] or [batch,n,n]; X ([.., m, n]) becomes X L^-T.

    Returns the int32 info tensor [batch] (device; 0 = ok, j = leading minor j not PD)."""
    _dev(A, X)
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    if A3.shape[1] != A3.shape[2] or A3.stride(2) != 1:
        raise GpvaeError("A must be [batch,n,n] with unit column stride")
    batch, n = A3.shape[0], A3.shape[1]
    m = 0
    X3 = None
    if X is not None:
        X3 = X if X.dim() == 3 else X.unsqueeze(0)
        if X3.shape[0] != batch or X3.shape[2] != n or X3.stride(2) != 1 or X3.dtype != A.dtype:
            raise GpvaeError("X must be [batch,m,n], same dtype, unit column stride")
        m = X3.shape[1]
    info = torch.empty((batch,), dtype=torch.int32, device=A.device)
    flags = ((POTRF_ZERO_UPPER if zero_upper else 0) | (POTRF_NAN_ON_FAIL if nan_on_fail else 0)
             | (POTRF_SHARE_CHIP if share_chip else 0))  # share_chip: a hint for calls that run beside other work
    lib = _lib.load()
    if work is None:
        work = torch.empty(max(16, lib.gpvae_potrf_workspace_bytes(_dt(A), n, m, batch)),
                           dtype=torch.uint8, device=A.device)
    check(lib.gpvae_potrf(_stream(), _dt(A), n, batch, _p(A3), A3.stride(1), A3.stride(0), m,
                          _p(X3), X3.stride(1) if m else 0, X3.stride(0) if m else 0, _p(info),
                          flags, _p(work)), "gpvae_potrf")
    return info


_side_streams = {}


def potrf_overlapped_(A, parts=2, zero_upper=True, nan_on_fail=True):
    """potrf_ of a batch [batch,n,n] issued as `parts` sub-batches on side streams (forked from and
    joined back to the current stream): the latency-bound panel launches of one sub-batch run under
    the MFMA-bound trailing updates of the other.  Same results as potrf_ (the matrices are
    independent); pays from batch 32 up at n = 4096 (+3-5 %), loses below.  Returns info [batch]."""
    _dev(A)
    if A.dim() != 3:
        raise GpvaeError("potrf_overlapped_: A must be [batch,n,n]")
    batch = A.shape[0]
    parts = max(1, min(int(parts), batch))
    if parts == 1:
        return potrf_(A, zero_upper=zero_upper, nan_on_fail=nan_on_fail)
    key = (A.device.index, parts)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=A.device) for _ in range(parts)]
    cur = torch.cuda.current_stream(A.device)
    fork = torch.cuda.Event()
    fork.record(cur)
    bounds = [batch * i // parts for i in range(parts + 1)]
    infos = []
    for i, s in enumerate(_side_streams[key]):
        s.wait_event(fork)
        with torch.cuda.stream(s):
            sub = A[bounds[i]:bounds[i + 1]]
            sub.record_stream(s)
            infos.append(potrf_(sub, zero_upper=zero_upper, nan_on_fail=nan_on_fail))
            infos[-1].record_stream(cur)
        join = torch.cuda.Event()
        join.record(s)
        cur.wait_event(join)
    return torch.cat(infos)


def trsm_(L, X, trans=True):
    """In place X <- X L^-T (trans, every row x becomes the y with L y = x) or X <- X L^-1 (every row
    becomes the y with L^T y = x) against an existing lower factor L [n,n]; X [m,n].  Both in turn give
    X A^-1 for A = L L^T.  The hot paths do not need this (their solves ride gpvae_potrf's extra rows)."""
    _dev(L, X)
    n = L.shape[0]
    if L.dim() != 2 or L.shape != (n, n) or X.dim() != 2 or X.shape[1] != n or L.stride(1) != 1 or X.stride(1) != 1:
        raise GpvaeError("trsm_: L must be [n,n], X [m,n], unit column strides")
    if X.dtype != L.dtype:
        raise GpvaeError("trsm_: dtype mismatch")
    lib = _lib.load()
    check(lib.gpvae_trsm(_stream(), _dt(L), 1 if trans else 0, n, X.shape[0], _p(L), L.stride(0) if n > 0 else 1,
                         _p(X), X.stride(0) if X.shape[0] > 0 else max(n, 1)), "gpvae_trsm")
    return X


def draw(L, E, E2=None, sigma=None, want_f=False):
    """Y = E L^T (+ sigma*E2).  L [n,n] lower factor, E/E2 [nb,n].  Returns Y or (F, Y)."""
    _dev(L, E, E2)
    n, nb = L.shape[0], E.shape[0]
    if L.shape != (n, n) or E.shape[1] != n or L.stride(1) != 1 or E.stride(1) != 1:
        raise GpvaeError("draw: L must be [n,n], E [nb,n], unit column strides")
    if E2 is not None and (E2.shape != E.shape or E2.stride(1) != 1):
        raise GpvaeError("draw: E2 must match E")
    Y = torch.empty((nb, n), dtype=L.dtype, device=L.device)
    F = torch.empty((nb, n), dtype=L.dtype, device=L.device) if want_f else None
    sig = scalar(sigma, L) if E2 is not None else None
    lib = _lib.load()
    check(lib.gpvae_draw(_stream(), _dt(L), n, nb, _p(L), L.stride(0), _p(E), E.stride(0), _p(E2),
                         E2.stride(0) if E2 is not None else 0, _p(sig), _p(F),
                         F.stride(0) if want_f else 0, _p(Y), Y.stride(0)), "gpvae_draw")
    return (F, Y) if want_f else Y


def draw_batched(L, E, E2=None, sigma=None, want_f=False):
    """One draw per factor: Y[b] = L[b] E[b] (+ sigma[b]*E2[b]).  L [nb,n,n] lower factors, E/E2 [nb,n],
    sigma [nb] (or a scalar).  Returns Y or (F, Y).  (Predictive(gp.sample) with sampled
    hyper-parameters: vae.py:226, toy_examples.py:88.)"""
    _dev(L, E, E2)
    if L.dim() != 3 or L.shape[1] != L.shape[2] or L.stride(2) != 1:
        raise GpvaeError("draw_batched: L must be [nb,n,n] with unit column stride")
    nb, n = L.shape[0], L.shape[1]
    if tuple(E.shape) != (nb, n) or E.stride(1) != 1 or E.dtype != L.dtype:
        raise GpvaeError("draw_batched: E must be [nb,n], same dtype, unit column stride")
    if E2 is not None and (E2.shape != E.shape or E2.stride(1) != 1 or E2.dtype != L.dtype):
        raise GpvaeError("draw_batched: E2 must match E")
    Y = torch.empty((nb, n), dtype=L.dtype, device=L.device)
    F = torch.empty((nb, n), dtype=L.dtype, device=L.device) if want_f else None
    sig = None
    if E2 is not None:
        sig = scalar(sigma, L)
        if sig.numel() == 1:
            sig = sig.expand(nb).contiguous()
        if sig.numel() != nb:
            raise GpvaeError("draw_batched: sigma needs one entry per factor")
    lib = _lib.load()
    check(lib.gpvae_draw_batched(_stream(), _dt(L), n, nb, _p(L), L.stride(1), L.stride(0), _p(E),
                                 E.stride(0), _p(E2), E2.stride(0) if E2 is not None else 0, _p(sig),
                                 _p(F), F.stride(0) if want_f else 0, _p(Y), Y.stride(0)),
          "gpvae_draw_batched")
    return (F, Y) if want_f else Y


def randn(shape, dtype=torch.float32, device="cuda", seed=0, stream_id=0, out=None):
    """Philox4x32-10 / Box-Muller standard normals; element i depends only on (seed, stream_id, i)."""
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=device)
    _dev(out)
    if not out.is_contiguous():
        raise GpvaeError("randn: out must be contiguous")
    lib = _lib.load()
    check(lib.gpvae_randn(_stream(), _dt(out), out.numel(), int(seed), int(stream_id), _p(out)),
          "gpvae_randn")
    return out


# ---- fp16 x 2: the draw on two fp16 planes per operand (opt-in; include/gpvae.h) --------------------------
def split_h2(X, bound, lower=False, out=None, inv_scale=None):
    """float32 [rows, cols] or [batch, rows, cols] -> (planes [.., 2, rows, cols] fp16 of X * 2^e, inv_scale [batch]
    = 2^-e); `bound`: per-matrix upper bound of |entries| (scalar or [batch] tensor)."""
    _dev(X, out)
    X3 = X if X.dim() == 3 else X.unsqueeze(0)
    if X3.dtype != torch.float32 or X3.stride(2) != 1:
        raise GpvaeError("split_h2: X must be float32 with unit column stride")
    batch, rows, cols = X3.shape
    if out is None:
        out = torch.empty((batch, 2, rows, cols), dtype=torch.float16, device=X.device)
    o = out.view(batch, 2, rows, cols)
    b = bound if isinstance(bound, torch.Tensor) else torch.full((batch,), float(bound), device=X.device)
    b = b.to(device=X.device, dtype=torch.float32).reshape(-1).contiguous()
    if b.numel() != batch:
        raise GpvaeError("split_h2: one bound per matrix")
    if inv_scale is None:
        inv_scale = torch.empty(batch, dtype=torch.float32, device=X.device)
    check(_lib.load().gpvae_split_h2(_stream(), rows, cols, batch, _p(X3), X3.stride(1), X3.stride(0), int(bool(lower)),
                                     _p(b), _p(o), o.stride(2), o.stride(1), o.stride(0), _p(inv_scale)), "gpvae_split_h2")
    return (o if X.dim() == 3 else o[0]), inv_scale


def randn_h2(shape, device="cuda", seed=0, stream_id=0, out=None):
    """The float32 normals of randn(shape, seed, stream_id), times 2^12, as fp16 planes [2, *shape]."""
    if out is None:
        out = torch.empty((2,) + tuple(shape), dtype=torch.float16, device=device)
    _dev(out)
    if not out.is_contiguous() or out.dtype != torch.float16 or out.shape[0] != 2:
        raise GpvaeError("randn_h2: out must be a contiguous float16 [2, ...] tensor")
    count = out[0].numel()
    check(_lib.load().gpvae_randn_h2(_stream(), count, int(seed), int(stream_id), _p(out), count), "gpvae_randn_h2")
    return out


def draw_h2(L2, inv_scale, E2h, E2=None, sigma=None, want_f=False, out=None):
    """Y = E L^T (+ sigma*E2) from fp16 x 2 operands: L2 [2,n,n] with its inv_scale [1] (split_h2), E2h [2,nb,n]
    (randn_h2), E2 [nb,n] float32.  Returns Y or (F, Y), float32."""
    _dev(L2, E2h, E2, out, inv_scale)
    n, nb = L2.shape[1], E2h.shape[1]
    Y = torch.empty((nb, n), dtype=torch.float32, device=L2.device) if out is None else out
    F = torch.empty((nb, n), dtype=torch.float32, device=L2.device) if want_f else None
    sig = scalar(sigma, Y) if E2 is not None else None
    check(_lib.load().gpvae_draw_h2(_stream(), n, nb, _p(L2), L2.stride(1), L2.stride(0), _p(inv_scale), _p(E2h),
                                    E2h.stride(1), E2h.stride(0), _p(E2), E2.stride(0) if E2 is not None else 0,
                                    _p(sig), _p(F), n if want_f else 0, _p(Y), Y.stride(0)), "gpvae_draw_h2")
    return (F, Y) if want_f else Y


PRIOR_INVGAMMA, PRIOR_BETA = 0, 1


def hyper_sample(count, dtype=torch.float32, device="cuda", seed=0, stream_id=0, first_index=0,
                 ls_prior=PRIOR_INVGAMMA, ls_a=1.0, ls_b=0.1, var_scale=0.1, sigma_scale=0.1, out=None):
    """(ls, var, sigma) [count] from the hyper-priors on the device (vae.py:157-159 by default:
    InverseGamma(1, .1), LogNormal(0, .1), HalfNormal(.1)).  out: optional [3, >=count] tensor whose rows
    var | ls | sigma (StepEngine's layout) are filled."""
    if out is None:
        out = torch.empty((3, count), dtype=dtype, device=device)
    _dev(out)
    if out.dim() != 2 or out.shape[0] != 3 or out.shape[1] < count or (out.shape[1] > 1 and out.stride(1) != 1):
        raise GpvaeError("hyper_sample: out must be [3, >= count] with contiguous rows")
    check(_lib.load().gpvae_hyper_sample(_stream(), _dt(out), int(count), int(seed), int(stream_id), int(first_index),
                                         int(ls_prior), float(ls_a), float(ls_b), float(var_scale),
                                         float(sigma_scale), _p(out[1]), _p(out[0]), _p(out[2])),
          "gpvae_hyper_sample")
    return out[1, :count], out[0, :count], out[2, :count]


def poisson_exp(f, seed=0, stream_id=0, first_index=0, want_y=True):
    """rate = exp(f) and y ~ Poisson(rate) (int32) elementwise: the observation of PoiGP.sample (gp.py:249-250).
    Element i of the flattened f draws from Philox counter first_index + i of (seed, stream_id).
    Returns (rate, y) shaped like f (y None when want_y is False)."""
    _dev(f)
    f = f.contiguous()
    rate = torch.empty_like(f)
    y = torch.empty(f.shape, dtype=torch.int32, device=f.device) if want_y else None
    check(_lib.load().gpvae_poisson_exp(_stream(), _dt(f), f.numel(), int(seed), int(stream_id), int(first_index),
                                        _p(f), _p(rate), _p(y)), "gpvae_poisson_exp")
    return rate, y


# ---- bf16 x 3 (b3) operands: three bf16 planes that sum to the float exactly --------------------------
def split_b3(X, lower=False, out=None):
    """float32 X [rows, cols] or [batch, rows, cols] -> bf16 planes [3, rows, cols] / [batch, 3, rows, cols]
    (lower: zeros above the diagonal)."""
    _dev(X, out)
    if X.dtype != torch.float32 or X.stride(-1) != 1:
        raise GpvaeError("split_b3: X must be float32 with unit column stride")
    X3 = X if X.dim() == 3 else X.unsqueeze(0)
    batch, rows, cols = X3.shape
    if out is None:
        out = torch.empty((batch, 3, rows, cols), dtype=torch.bfloat16, device=X.device)
    o = out.view(batch, 3, rows, cols)
    check(_lib.load().gpvae_split_b3(_stream(), rows, cols, batch, _p(X3), X3.stride(1), X3.stride(0),
                                     1 if lower else 0, _p(o), o.stride(2), o.stride(1), o.stride(0)),
          "gpvae_split_b3")
    return o if X.dim() == 3 else o[0]


def potrf_b3_supported(n):
    return bool(_lib.load().gpvae_potrf_b3_supported(int(n)))


def potrf_b3_(A, L3=None, zero_upper=False, nan_on_fail=False, work=None, info=None, share_chip=False):
    """In-place float32 Cholesky of A [batch,n,n] (or [n,n]) that also leaves the factor as bf16 x 3 planes
    L3 [batch,3,n,n] for draw_b3 (written by the panel launches themselves: no split pass).
    Returns (info, L3)."""
    _dev(A, L3)
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    batch, n = A3.shape[0], A3.shape[1]
    if A3.dtype != torch.float32 or A3.shape[2] != n or A3.stride(2) != 1:
        raise GpvaeError("potrf_b3_: A must be float32 [batch,n,n] with unit column stride")
    if L3 is None:
        L3 = torch.empty((batch, 3, n, n), dtype=torch.bfloat16, device=A.device)
    o = L3.view(batch, 3, n, n)
    if info is None:
        info = torch.empty((batch,), dtype=torch.int32, device=A.device)
    lib = _lib.load()
    if work is None:
        work = torch.empty(max(16, lib.gpvae_potrf_b3_workspace_bytes(n, batch)), dtype=torch.uint8, device=A.device)
    flags = ((POTRF_ZERO_UPPER if zero_upper else 0) | (POTRF_NAN_ON_FAIL if nan_on_fail else 0)
             | (POTRF_SHARE_CHIP if share_chip else 0))  # share_chip: a hint for calls that run beside other work
    check(lib.gpvae_potrf_b3(_stream(), n, batch, _p(A3), A3.stride(1), A3.stride(0), _p(info), flags, _p(work),
                             _p(o), o.stride(2), o.stride(1), o.stride(0)), "gpvae_potrf_b3")
    return info, (o if A.dim() == 3 else o[0])


def potrf_h2_(A, L2=None, inv_scale=None, zero_upper=False, nan_on_fail=False, work=None, info=None, share_chip=False):
    """potrf_b3_ with the factor left as fp16 x 2 planes L2 [batch,2,n,n] of L * 2^e_b for draw_h2 and the trailing
    updates on those planes; inv_scale [batch] float32 = 2^-e_b.  Returns (info, L2, inv_scale)."""
    _dev(A, L2, inv_scale)
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    batch, n = A3.shape[0], A3.shape[1]
    if A3.dtype != torch.float32 or A3.shape[2] != n or A3.stride(2) != 1:
        raise GpvaeError("potrf_h2_: A must be float32 [batch,n,n] with unit column stride")
    if L2 is None:
        L2 = torch.empty((batch, 2, n, n), dtype=torch.float16, device=A.device)
    if inv_scale is None:
        inv_scale = torch.empty((batch,), dtype=torch.float32, device=A.device)
    o = L2.view(batch, 2, n, n)
    if info is None:
        info = torch.empty((batch,), dtype=torch.int32, device=A.device)
    lib = _lib.load()
    if work is None:
        work = torch.empty(max(16, lib.gpvae_potrf_h2_workspace_bytes(n, batch)), dtype=torch.uint8, device=A.device)
    flags = ((POTRF_ZERO_UPPER if zero_upper else 0) | (POTRF_NAN_ON_FAIL if nan_on_fail else 0)
             | (POTRF_SHARE_CHIP if share_chip else 0))  # share_chip: a hint for calls that run beside other work
    check(lib.gpvae_potrf_h2(_stream(), n, batch, _p(A3), A3.stride(1), A3.stride(0), _p(info), flags, _p(work),
                             _p(o), o.stride(2), o.stride(1), o.stride(0), _p(inv_scale)), "gpvae_potrf_h2")
    return info, (o if A.dim() == 3 else o[0]), inv_scale


def randn_b3(shape, device="cuda", seed=0, stream_id=0, out=None):
    """The float32 normals of randn(shape, seed, stream_id) as bf16 planes [3, *shape]."""
    if out is None:
        out = torch.empty((3,) + tuple(shape), dtype=torch.bfloat16, device=device)
    _dev(out)
    if not out.is_contiguous() or out.dtype != torch.bfloat16 or out.shape[0] != 3:
        raise GpvaeError("randn_b3: out must be a contiguous bfloat16 [3, ...] tensor")
    count = out[0].numel()
    check(_lib.load().gpvae_randn_b3(_stream(), count, int(seed), int(stream_id), _p(out), count),
          "gpvae_randn_b3")
    return out


def draw_b3_supported(n, nb):
    return bool(_lib.load().gpvae_draw_b3_supported(int(n), int(nb)))


def draw_b3(L3, E3, E2=None, sigma=None, want_f=False, out=None):
    """Y = E L^T (+ sigma*E2) from b3 operands: L3 [3,n,n] (zeros above the diagonal), E3 [3,nb,n]
    (bf16 planes), E2 [nb,n] float32.  Returns Y or (F, Y), float32."""
    _dev(L3, E3, E2, out)
    n, nb = L3.shape[1], E3.shape[1]
    Y = torch.empty((nb, n), dtype=torch.float32, device=L3.device) if out is None else out
    F = torch.empty((nb, n), dtype=torch.float32, device=L3.device) if want_f else None
    sig = scalar(sigma, Y) if E2 is not None else None
    check(_lib.load().gpvae_draw_b3(_stream(), n, nb, _p(L3), L3.stride(1), L3.stride(0), _p(E3), E3.stride(1),
                                    E3.stride(0), _p(E2), E2.stride(0) if E2 is not None else 0, _p(sig),
                                    _p(F), F.stride(0) if want_f else 0, _p(Y), Y.stride(0)), "gpvae_draw_b3")
    return (F, Y) if want_f else Y


# ---- a9 kriging ---------------------------------------------------------------------------------------
def krig_predict(x_obs, x_test, y_obs, var, ls, sigma, jitter=1.0e-5, xi=None):
    """(mean, sample, var) at x_test given observations; literal gp_prediction arithmetic.

    Returns also the int32 info tensor of the factorisation."""
    _dev(x_obs, x_test, y_obs, xi)
    x_obs, x_test = _points(x_obs), _points(x_test)
    n_obs, n_test, d = x_obs.shape[0], x_test.shape[0], x_obs.shape[1]
    y_obs = y_obs.contiguous()
    lib = _lib.load()
    nbytes = lib.gpvae_krig_workspace_bytes(_dt(x_obs), n_obs, n_test)
    work = torch.empty(nbytes, dtype=torch.uint8, device=x_obs.device)
    mean = torch.empty(n_test, dtype=x_obs.dtype, device=x_obs.device)
    samp = torch.empty_like(mean)
    vout = torch.empty_like(mean)
    info = torch.empty(1, dtype=torch.int32, device=x_obs.device)
    if xi is not None:
        xi = xi.contiguous()
    # named locals: a temporary's storage would be recycled for the next scalar before the launch
    var_t, ls_t, sig_t = scalar(var, x_obs), scalar(ls, x_obs), scalar(sigma, x_obs)
    check(lib.gpvae_krig_predict(_stream(), _dt(x_obs), d, n_obs, n_test, _p(x_obs), _p(x_test),
                                 _p(y_obs), _p(var_t), _p(ls_t), _p(sig_t), float(jitter), _p(xi),
                                 _p(mean), _p(samp), _p(vout), _p(info), _p(work)),
          "gpvae_krig_predict")
    return mean, samp, vout, info


def krig_predict_batched(x_obs, x_test, y_obs, var, ls, sigma, jitter=1.0e-5, xi=None):
    """krig_predict for a batch of hyper-parameter draws sharing points and observations
    (the vmap of inference.py:517-521).  var, ls, sigma: 1-D [S]; xi: [S, n_test] or None.
    Returns mean, sample, var: [S, n_test] and info int32[S]."""
    _dev(x_obs, x_test, y_obs, xi)
    x_obs, x_test = _points(x_obs), _points(x_test)
    n_obs, n_test, d = x_obs.shape[0], x_test.shape[0], x_obs.shape[1]
    dt, dv = x_obs.dtype, x_obs.device
    var_t = torch.as_tensor(var, dtype=dt, device=dv).reshape(-1).contiguous()
    ls_t = torch.as_tensor(ls, dtype=dt, device=dv).reshape(-1).contiguous()
    sig_t = torch.as_tensor(sigma, dtype=dt, device=dv).reshape(-1).contiguous()
    S = var_t.numel()
    if ls_t.numel() != S or sig_t.numel() != S:
        raise GpvaeError("krig_predict_batched: var, ls, sigma need the same length")
    if xi is not None:
        xi = xi.contiguous()
        if tuple(xi.shape) != (S, n_test):
            raise GpvaeError("krig_predict_batched: xi must be [S, n_test]")
    y_obs = y_obs.contiguous()
    lib = _lib.load()
    nbytes = lib.gpvae_krig_batched_workspace_bytes(_dt(x_obs), n_obs, n_test, S)
    work = torch.empty(nbytes, dtype=torch.uint8, device=dv)
    mean = torch.empty((S, n_test), dtype=dt, device=dv)
    samp = torch.empty_like(mean)
    vout = torch.empty_like(mean)
    info = torch.empty(S, dtype=torch.int32, device=dv)
    check(lib.gpvae_krig_predict_batched(_stream(), _dt(x_obs), d, n_obs, n_test, S, _p(x_obs),
                                         _p(x_test), _p(y_obs), _p(var_t), _p(ls_t), _p(sig_t),
                                         float(jitter), _p(xi), _p(mean), _p(samp), _p(vout),
                                         _p(info), _p(work)), "gpvae_krig_predict_batched")
    return mean, samp, vout, info


def krig_set_path(force_big):
    """True: gpvae_krig_predict_batched always takes the batched Gram + gpvae_potrf route; False (default): batches of
    small problems run one workgroup per problem with everything in LDS.  For tests and benchmarks."""
    _lib.load().gpvae_krig_set_path(1 if force_big else 0)


# ---- a10 / f1: MVN log-density and gradient ---------------------------------------------------------
def mvn_logp(x, f, var, ls, jitter=1.0e-5, grad=True):
    """log N(f_b; 0, K(x; var_b, ls_b) + jitter I) for a batch of chains, with d/df, d/dvar, d/dls.

    x: [n] or [n,d]; f: [S,n] (or [n]); var, ls: [S] (or scalars).
    Returns (logp [S], grad_f [S,n], grad_var [S], grad_ls [S], info int32[S]); gradients None if
    grad=False."""
    _dev(x, f)
    x = _points(x)
    n, d = x.shape
    dt, dv = x.dtype, x.device
    f = f.reshape(-1, n).contiguous()
    S = f.shape[0]
    var_t = torch.as_tensor(var, dtype=dt, device=dv).reshape(-1).contiguous()
    ls_t = torch.as_tensor(ls, dtype=dt, device=dv).reshape(-1).contiguous()
    if var_t.numel() != S or ls_t.numel() != S:
        raise GpvaeError("mvn_logp: var and ls need one entry per row of f")
    lib = _lib.load()
    work = torch.empty(lib.gpvae_mvn_workspace_bytes(_dt(x), n, S), dtype=torch.uint8, device=dv)
    logp = torch.empty(S, dtype=dt, device=dv)
    info = torch.empty(S, dtype=torch.int32, device=dv)
    gf = torch.empty((S, n), dtype=dt, device=dv) if grad else None
    gv = torch.empty(S, dtype=dt, device=dv) if grad else None
    gl = torch.empty(S, dtype=dt, device=dv) if grad else None
    check(lib.gpvae_mvn_logp(_stream(), _dt(x), d, n, S, _p(x), _p(f), _p(var_t), _p(ls_t),
                             float(jitter), _p(logp), _p(gf), _p(gv), _p(gl), _p(info), _p(work)),
          "gpvae_mvn_logp")
    return logp, gf, gv, gl, info


def check_index(idx, n, what="index"):
    """Raise unless every entry of the int64 device tensor idx lies in [0, n).  One host synchronisation: call it
    where an index set is BUILT (Inference.__init__ / regression), not per potential evaluation.  Repeated
    entries are legal: f[obs_idx] counts such a point once per occurrence, as the reference's gather does."""
    if idx.dtype != torch.long:
        raise GpvaeError("%s must be an int64 tensor" % what)
    if idx.numel() and (int(idx.min().item()) < 0 or int(idx.max().item()) >= int(n)):
        raise GpvaeError("%s out of range: entries must lie in [0, %d) (got min %d, max %d)"
                         % (what, int(n), int(idx.min().item()), int(idx.max().item())))
    return idx


def gp_potential(x, q, y_lik, lik_idx=None, jitter=1.0e-5, checked=False):
    """(U [chains], dU/dq [chains, n+3], info) of the latent-f GP regression model for rows
    q = (f [n], log var, log ls, log sigma): gpvae_gp_potential (inference.py:359-424).
    lik_idx (int64, repeats allowed) is range-checked here unless the caller did so already (checked=True:
    samplers call this thousands of times with one index set)."""
    _dev(x, q, y_lik, lik_idx)
    x = _points(x)
    n, d = x.shape
    if q.dim() != 2 or q.shape[1] != n + 3 or q.stride(1) != 1 or q.dtype != x.dtype:
        raise GpvaeError("gp_potential: q must be [chains, n+3], unit column stride, dtype of x")
    chains = q.shape[0]
    y_lik = y_lik.contiguous()
    n_lik = y_lik.numel()
    if lik_idx is not None:
        lik_idx = lik_idx.contiguous()
        if lik_idx.dtype != torch.long or lik_idx.numel() != n_lik:
            raise GpvaeError("gp_potential: lik_idx must be int64 with one entry per observation")
        if not checked:
            check_index(lik_idx, n, "gp_potential: lik_idx")
    lib = _lib.load()
    work = torch.empty(max(16, lib.gpvae_gp_potential_workspace_bytes(_dt(x), n, chains)), dtype=torch.uint8, device=x.device)
    U = torch.empty(chains, dtype=x.dtype, device=x.device)
    dU = torch.empty((chains, n + 3), dtype=x.dtype, device=x.device)
    info = torch.empty(chains, dtype=torch.int32, device=x.device)
    check(lib.gpvae_gp_potential(_stream(), _dt(x), d, n, chains, _p(x), _p(q), q.stride(0), _p(y_lik), _p(lik_idx),
                                 n_lik, float(jitter), _p(U), _p(dU), dU.stride(0), _p(info), _p(work)),
          "gpvae_gp_potential")
    return U, dU, info


# ---- a5-a7, a11: VAE -----------------------------------------------------------------------------------
def vae_param_count(n, h0, h1, z):
    return int(_lib.load().gpvae_vae_param_count(n, h0, h1, z))


class VaeWorkspace:
    """Caller-owned scratch for vae_elbo at a fixed (B, n, h0, h1, z, dtype)."""

    def __init__(self, B, n, h0, h1, z, dtype, device):
        lib = _lib.load()
        self.key = (B, n, h0, h1, z, dtype)
        nbytes = lib.gpvae_vae_workspace_bytes(_DT[dtype], B, n, h0, h1, z)
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)


def vae_elbo(params, Y, eps_z, dims, obs_sigma=0.1, grads=None, loss=None, work=None):
    """loss = -ELBO(Y) (device scalar); if `grads` (flat like params) is given it is overwritten."""
    _dev(params, Y, eps_z, grads)
    h0, h1, z = dims
    B, n = Y.shape
    if Y.stride(1) != 1 or not eps_z.is_contiguous() or eps_z.shape != (B, z):
        raise GpvaeError("vae_elbo: Y needs unit column stride, eps_z must be contiguous [B,z]")
    if params.numel() != vae_param_count(n, h0, h1, z) or not params.is_contiguous():
        raise GpvaeError("vae_elbo: params has the wrong length for (n,h0,h1,z)")
    if work is None or work.key != (B, n, h0, h1, z, Y.dtype):
        work = VaeWorkspace(B, n, h0, h1, z, Y.dtype, Y.device)
    if loss is None:
        loss = torch.empty(1, dtype=Y.dtype, device=Y.device)
    lib = _lib.load()
    check(lib.gpvae_vae_elbo(_stream(), _dt(Y), B, n, h0, h1, z, _p(params), _p(Y), Y.stride(0),
                             _p(eps_z), float(obs_sigma), _p(loss), _p(grads), _p(work.buf)),
          "gpvae_vae_elbo")
    return loss


def encoder_fwd(params, Y, dims, eps_z=None):
    """(mu, sd[, z]) of the encoder (vae_encoder's apply, vae.py:72-85; z = mu + sd*eps_z is the guide's
    sample, vae.py:135).  params: the flat vector (or its encoder head); Y [B,n]; eps_z [B,z] or None."""
    _dev(params, Y, eps_z)
    h0, h1, z = dims
    B, n = Y.shape
    if Y.stride(1) != 1:
        raise GpvaeError("encoder_fwd: Y needs unit column stride")
    head = n * h0 + h0 + h0 * h1 + h1 + 2 * (h1 * z + z)
    if params.numel() < head or not params.is_contiguous() or params.dtype != Y.dtype:
        raise GpvaeError("encoder_fwd: params too short for (n,h0,h1,z), not contiguous or wrong dtype")
    if eps_z is not None and (tuple(eps_z.shape) != (B, z) or not eps_z.is_contiguous()):
        raise GpvaeError("encoder_fwd: eps_z must be contiguous [B,z]")
    mu = torch.empty((B, z), dtype=Y.dtype, device=Y.device)
    sd = torch.empty_like(mu)
    zz = torch.empty_like(mu) if eps_z is not None else None
    lib = _lib.load()
    work = torch.empty(lib.gpvae_encoder_workspace_bytes(_dt(Y), B, n, h0, h1, z), dtype=torch.uint8,
                       device=Y.device)
    check(lib.gpvae_encoder_fwd(_stream(), _dt(Y), B, n, h0, h1, z, _p(params), _p(Y), Y.stride(0),
                                _p(eps_z), _p(mu), _p(sd), _p(zz), _p(work)), "gpvae_encoder_fwd")
    return (mu, sd) if eps_z is None else (mu, sd, zz)


def adam_(params, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    _dev(params, grads, m, v)
    lib = _lib.load()
    check(lib.gpvae_adam(_stream(), _dt(params), params.numel(), _p(params), _p(grads), _p(m), _p(v),
                         int(t), float(lr), float(b1), float(b2), float(eps)), "gpvae_adam")


def _dec_work(lib, Z, n, h0, h1, z):
    nbytes = lib.gpvae_decoder_workspace_bytes(_dt(Z), Z.shape[0], n, h0, h1, z)
    return torch.empty(nbytes, dtype=torch.uint8, device=Z.device)


def decoder_fwd(dec_params, Z, n, h0, h1):
    """F[c] = decoder(params, Z[c]); dec_params flat: V1 c1 V2 c2 V3 c3."""
    _dev(dec_params, Z)
    Z = Z.contiguous()
    C_, z = Z.shape
    F = torch.empty((C_, n), dtype=Z.dtype, device=Z.device)
    lib = _lib.load()
    check(lib.gpvae_decoder_fwd(_stream(), _dt(Z), C_, n, h0, h1, z, _p(dec_params), _p(Z), _p(F),
                                _p(_dec_work(lib, Z, n, h0, h1, z))), "gpvae_decoder_fwd")
    return F


def decoder_vjp(dec_params, Z, dF, h0, h1):
    _dev(dec_params, Z, dF)
    Z, dF = Z.contiguous(), dF.contiguous()
    C_, z = Z.shape
    n = dF.shape[1]
    dZ = torch.empty_like(Z)
    lib = _lib.load()
    check(lib.gpvae_decoder_vjp(_stream(), _dt(Z), C_, n, h0, h1, z, _p(dec_params), _p(Z), _p(dF),
                                _p(dZ), _p(_dec_work(lib, Z, n, h0, h1, z))), "gpvae_decoder_vjp")
    return dZ


def vae_potential_supported(dtype, h0, h1, z, n_obs):
    return bool(_lib.load().gpvae_vae_potential_supported(_DT[dtype], h0, h1, z, int(n_obs)))


def vae_potential(dec_params, q, y_obs, obs_idx, n, h0, h1, noise_scale=0.1, checked=False):
    """(U [chains], dU/dq [chains, z+1]) of VAEInference.regression for q = (z, log noise) rows: one launch
    (gpvae_vae_potential).  obs_idx: int64 device tensor (repeats allowed), range-checked here unless
    checked=True."""
    _dev(dec_params, q, y_obs, obs_idx)
    if q.stride(1) != 1 or obs_idx.dtype != torch.long or not obs_idx.is_contiguous() or not y_obs.is_contiguous():
        raise GpvaeError("vae_potential: q needs unit column stride; obs_idx int64 contiguous; y_obs contiguous")
    if y_obs.numel() != obs_idx.numel():
        raise GpvaeError("vae_potential: one observation per entry of obs_idx")
    if not checked:
        check_index(obs_idx, n, "vae_potential: obs_idx")
    chains, z = q.shape[0], q.shape[1] - 1
    U = torch.empty(chains, dtype=q.dtype, device=q.device)
    dU = torch.empty((chains, z + 1), dtype=q.dtype, device=q.device)
    check(_lib.load().gpvae_vae_potential(_stream(), _dt(q), chains, n, h0, h1, z, _p(dec_params), _p(q), q.stride(0),
                                          _p(y_obs), _p(obs_idx), obs_idx.numel(), float(noise_scale), _p(U), _p(dU),
                                          dU.stride(0)), "gpvae_vae_potential")
    return U, dU


def decoder_vjp_static(dec_params, Z, dF, h0, h1, out=None):
    """decoder_vjp through gpvae_decoder_vjp_static: no workspace, no allocation inside the library
    (its scratch is a 4 MiB static device array per GPU; calls that overlap on one GPU race).  Pass
    `out` to keep this wrapper allocation-free too."""
    _dev(dec_params, Z, dF, out)
    if not (Z.is_contiguous() and dF.is_contiguous()):
        raise GpvaeError("decoder_vjp_static: Z and dF must be contiguous")
    C_, z = Z.shape
    n = dF.shape[1]
    dZ = torch.empty_like(Z) if out is None else out
    check(_lib.load().gpvae_decoder_vjp_static(_stream(), _dt(Z), C_, n, h0, h1, z, _p(dec_params), _p(Z),
                                               _p(dF), _p(dZ)), "gpvae_decoder_vjp_static")
    return dZ
