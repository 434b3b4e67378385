"""Batched No-U-Turn sampler over a potential-and-gradient callable.

The reference hands its models to numpyro.infer.NUTS (src/model/inference.py:84-95, 477-487); NumPyro
is third-party code that is absent here, so this module restates the published algorithm it
implements — multinomial NUTS (Hoffman & Gelman 2014; Betancourt 2017) with the iterative tree
builder of Phan et al. 2019 (U-turn checks of every balanced sub-tree from at most `max_tree_depth`
momentum checkpoints), biased progressive sampling between doublings, the generalised U-turn
criterion on momentum sums, divergence threshold 1000, dual-averaging step sizes and a diagonal mass
matrix estimated over Stan's expanding warm-up windows.  It is not a copy of NumPyro's code and makes
no claim of reproducing its random stream.

All chains advance in lock step: one potential-and-gradient evaluation per leapfrog step for the whole
batch (that is how the HIP kernels behind it — gpvae_vae_potential, gpvae_gp_potential — want to be
called); chains whose tree has stopped ride along masked, so a chain's trajectory depends only on its
own numbers.  The tree depth of an iteration is the deepest tree of the batch.
"""
import ctypes as C
import math

import torch


def uniforms_per_iteration(max_tree_depth):
    """Columns of uniforms one NUTS iteration consumes: a direction and a transition draw per
    doubling, one multinomial draw per leaf."""
    return 2 * max_tree_depth + (1 << max_tree_depth)


def _ckpt_range(leaf):
    """Checkpoint slots (idx_min, idx_max) the leaf `leaf` of a sub-tree talks to: idx_max = number of
    set bits of leaf >> 1, and an odd leaf closes as many balanced sub-trees as it has trailing ones."""
    idx_max = bin(leaf >> 1).count("1")
    ones, m = 0, leaf
    while m & 1:
        ones, m = ones + 1, m >> 1
    return idx_max - ones + 1, idx_max


def _turning(minv, p_a, p_b, p_sum):
    """Generalised U-turn criterion between the two ends of a (sub-)trajectory."""
    adj = p_sum - 0.5 * (p_a + p_b)
    return ((minv * p_a * adj).sum(1) <= 0) | ((minv * p_b * adj).sum(1) <= 0)


def _sel(mask, a, b):
    return torch.where(mask[:, None] if a.dim() == 2 else mask, a, b)


class _Windows():
    """Stan's warm-up schedule: a fast initial buffer, doubling slow windows (mass matrix), a fast
    final buffer.  `ends` = last iteration of each slow window."""

    def __init__(self, num_warmup):
        self.ends = []
        self.start = num_warmup
        if num_warmup < 20:
            return
        init, term, base = 75, 50, 25
        if init + base + term > num_warmup:
            init, term = int(0.15 * num_warmup), int(0.1 * num_warmup)
            base = num_warmup - init - term
        self.start = init
        lo, size = init, base
        while lo < num_warmup - term:
            hi = lo + size
            if hi + 2 * size > num_warmup - term:
                hi = num_warmup - term
            self.ends.append(hi - 1)
            lo, size = hi, 2 * size


class _Adapter():
    """Warm-up adaptation per chain: dual-averaging step size (Hoffman & Gelman 2014, alg. 5) and a diagonal
    mass matrix estimated over Stan's windows (Welford), with the step-size search restarted at each window's
    end.  Shared by the torch and the device drivers."""

    def __init__(self, C, D, dt, dev, step_size, target_accept, num_warmup, adapt_mass_matrix):
        self.minv = torch.ones((C, D), dtype=dt, device=dev)  # inverse mass diagonal, per chain
        self.log_eps = torch.full((C,), math.log(step_size), dtype=dt, device=dev)
        self.log_eps_bar = self.log_eps.clone()
        self.h_bar = torch.zeros(C, dtype=dt, device=dev)
        self.mu = self.log_eps + math.log(10.0)
        self.target = target_accept
        self.m = 0
        self.win = _Windows(num_warmup if adapt_mass_matrix else 0)
        self.w_n, self.w_mean, self.w_m2 = 0, torch.zeros((C, D), dtype=dt, device=dev), torch.zeros((C, D), dtype=dt, device=dev)

    def step_size(self, warm):
        return torch.exp(self.log_eps if warm else self.log_eps_bar)

    def update(self, it, a, q):
        gamma, t0, kappa = 0.05, 10.0, 0.75
        self.m += 1
        m = self.m
        self.h_bar = (1 - 1 / (m + t0)) * self.h_bar + (self.target - a) / (m + t0)
        self.log_eps = self.mu - math.sqrt(m) / gamma * self.h_bar
        w = m ** (-kappa)
        self.log_eps_bar = w * self.log_eps + (1 - w) * self.log_eps_bar
        win = self.win
        if win.ends and it >= win.start:
            self.w_n += 1
            d1 = q - self.w_mean
            self.w_mean = self.w_mean + d1 / self.w_n
            self.w_m2 = self.w_m2 + d1 * (q - self.w_mean)
            if it in win.ends and self.w_n > 1:
                n = self.w_n
                var = self.w_m2 / (n - 1)
                self.minv = (n / (n + 5.0)) * var + 1.0e-3 * (5.0 / (n + 5.0))
                self.w_n, self.w_mean, self.w_m2 = 0, torch.zeros_like(q), torch.zeros_like(q)
                # restart the step-size search around the last averaged value
                self.log_eps = self.log_eps_bar.clone()
                self.mu = self.log_eps + math.log(10.0)
                self.h_bar = torch.zeros_like(self.h_bar)
                self.m = 0


def nuts(potential_and_grad, q0, num_warmup, num_samples, thinning=1, max_tree_depth=10, step_size=0.1,
         target_accept=0.8, seed=0, noise_fn=None, adapt_mass_matrix=True):
    """potential_and_grad(q [C, D]) -> (U [C], dU/dq [C, D]).  q0: [C, D] initial positions.

    Returns (samples [num_samples // thinning, C, D], info); info: mean acceptance statistic per chain
    over the sampling phase, adapted step sizes, inverse mass diagonal, divergent transitions and the
    mean number of leapfrog steps per iteration and chain.

    noise_fn(it) -> (normals [C, D], uniforms [C, uniforms_per_iteration(max_tree_depth)]) for
    iteration `it` (the inference drivers key its rows by the global chain id).  Default: one torch
    generator for the whole batch."""
    q = q0.clone()
    C, D = q.shape
    dev, dt = q.device, q.dtype
    K = uniforms_per_iteration(max_tree_depth)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    U, g = potential_and_grad(q)
    if not torch.isfinite(U).all():
        raise ValueError("nuts: the potential is not finite at the initial positions")
    ad = _Adapter(C, D, dt, dev, step_size, target_accept, num_warmup, adapt_mass_matrix)
    minv = ad.minv
    keep = []
    acc_stat = torch.zeros(C, dtype=dt, device=dev)
    n_div = torch.zeros((), dtype=torch.long, device=dev)
    n_leap = torch.zeros(C, dtype=dt, device=dev)
    inf = torch.full((C,), float("inf"), dtype=dt, device=dev)

    for it in range(num_warmup + num_samples):
        warm = it < num_warmup
        eps = ad.step_size(warm)
        minv = ad.minv
        if noise_fn is None:
            z = torch.randn((C, D), dtype=dt, device=dev, generator=gen)
            unif = torch.rand((C, K), dtype=dt, device=dev, generator=gen)
        else:
            z, unif = noise_fn(it)
        p0 = z / torch.sqrt(minv)
        H0 = U + 0.5 * (minv * p0 * p0).sum(1)
        ql, pl, gl = q, p0, g
        qr, pr, gr = q, p0, g
        q_prop, U_prop, g_prop = q, U, g
        logW = torch.zeros(C, dtype=dt, device=dev)
        p_sum = p0.clone()
        active = torch.ones(C, dtype=torch.bool, device=dev)
        sum_acc = torch.zeros(C, dtype=dt, device=dev)
        n_prop = torch.zeros(C, dtype=dt, device=dev)
        diverged = torch.zeros(C, dtype=torch.bool, device=dev)

        for depth in range(max_tree_depth):
            right = unif[:, depth] < 0.5
            qe, pe, ge = _sel(right, qr, ql), _sel(right, pr, pl), _sel(right, gr, gl)
            es = torch.where(right, eps, -eps)[:, None]
            sub_logW = -inf
            sub_psum = torch.zeros_like(q)
            sq, sU, sg = qe, U, ge
            sub_stop = torch.zeros(C, dtype=torch.bool, device=dev)
            building = active
            ck_p = [None] * (max_tree_depth + 1)
            ck_s = [None] * (max_tree_depth + 1)
            leaves = 1 << depth
            base = 2 * max_tree_depth + leaves - 1
            for leaf in range(leaves):
                ph = pe - 0.5 * es * ge
                qn = qe + es * minv * ph
                Un, gn = potential_and_grad(qn)
                pn = ph - 0.5 * es * gn
                delta = Un + 0.5 * (minv * pn * pn).sum(1) - H0
                delta = torch.where(torch.isfinite(delta), delta, inf)
                div = delta > 1000.0
                sum_acc = sum_acc + torch.where(building, torch.exp(-torch.clamp(delta, min=0.0)), torch.zeros_like(delta))
                n_prop = n_prop + building.to(dt)
                new_logW = torch.logaddexp(sub_logW, -delta)
                take = building & ~div & (unif[:, base + leaf] < torch.exp(-delta - new_logW))
                sq, sU, sg = _sel(take, qn, sq), _sel(take, Un, sU), _sel(take, gn, sg)
                sub_logW = torch.where(building, new_logW, sub_logW)
                sub_psum = _sel(building, sub_psum + pn, sub_psum)
                qe, pe, ge = _sel(building, qn, qe), _sel(building, pn, pe), _sel(building, gn, ge)
                lo, hi = _ckpt_range(leaf)
                turn = torch.zeros(C, dtype=torch.bool, device=dev)
                if leaf % 2 == 0:
                    ck_p[hi], ck_s[hi] = pe, sub_psum
                else:
                    for i in range(hi, lo - 1, -1):
                        turn = turn | _turning(minv, ck_p[i], pe, sub_psum - ck_s[i] + ck_p[i])
                diverged = diverged | (building & div)
                sub_stop = sub_stop | (building & (turn | div))
                building = building & ~sub_stop
                if leaves >= 8 and (leaf & 3) == 3 and leaf + 1 < leaves and not bool(building.any()):
                    break
            ok = active & ~sub_stop
            take = ok & (unif[:, max_tree_depth + depth] < torch.exp(torch.clamp(sub_logW - logW, max=0.0)))
            q_prop, U_prop, g_prop = _sel(take, sq, q_prop), _sel(take, sU, U_prop), _sel(take, sg, g_prop)
            logW = torch.where(ok, torch.logaddexp(logW, sub_logW), logW)
            p_sum = _sel(ok, p_sum + sub_psum, p_sum)
            okr, okl = ok & right, ok & ~right
            qr, pr, gr = _sel(okr, qe, qr), _sel(okr, pe, pr), _sel(okr, ge, gr)
            ql, pl, gl = _sel(okl, qe, ql), _sel(okl, pe, pl), _sel(okl, ge, gl)
            active = ok & ~_turning(minv, pl, pr, p_sum)
            if not bool(active.any()):
                break

        q, U, g = q_prop, U_prop, g_prop
        a = sum_acc / torch.clamp(n_prop, min=1.0)
        if warm:
            ad.update(it, a, q)
        else:
            acc_stat += a
            n_div += diverged.sum()
            n_leap += n_prop
            if (it - num_warmup) % thinning == 0:
                keep.append(q.clone())
    samples = torch.stack(keep) if keep else q.new_empty((0, C, D))
    info = {"accept_prob": acc_stat / max(num_samples, 1), "step_size": ad.step_size(False),
            "inverse_mass": ad.minv, "divergent": int(n_div.item()),
            "leapfrogs_per_iteration": n_leap / max(num_samples, 1)}
    return samples, info


class _Buffers(C.Structure):
    """gpvae_nuts_buffers of include/gpvae.h."""
    _vp, _ip = C.c_void_p, C.POINTER(C.c_int32)
    _fields_ = ([(n, C.c_void_p) for n in ("q", "U", "g", "minv", "eps", "z", "unif")] + [("ldu", C.c_int64)]
                + [(n, C.c_void_p) for n in ("ql", "pl", "gl", "qr", "pr", "gr", "q_prop", "U_prop", "g_prop", "logW",
                                             "p_sum", "H0", "sum_acc", "n_prop", "accept", "active", "diverged",
                                             "qe", "pe", "ge", "es", "sub_logW", "sub_psum", "sq", "sU", "sg",
                                             "sub_stop", "building", "right", "ck_p", "ck_s", "ph", "qn", "Un", "gn")]
                + [("ldg", C.c_int64), ("flags", C.c_void_p)])


def nuts_device(potential_and_grad, q0, num_warmup, num_samples, thinning=1, max_tree_depth=10, step_size=0.1,
                target_accept=0.8, seed=0, noise_fn=None, adapt_mass_matrix=True, fused=True):
    """The same sampler with the transition in libgpvae_hip (gpvae_nuts_op: one wave per chain, two launches
    per leapfrog step around the caller's potential instead of ~40 elementwise torch kernels).  Same
    arguments, results and random-number consumption as :func:`nuts`; q0 must be a device tensor.
    fused (default): a potential with a `nuts_iteration` hook (VAEInference.regression's, whose decoder head
    fits LDS) runs every iteration as one launch of gpvae_nuts_vae_iteration — same chains, to rounding;
    fused=False keeps the lock-step gpvae_nuts_op driver."""
    from . import _lib
    from ._lib import check
    from .ops import _dt
    if not q0.is_cuda:
        raise _lib.GpvaeError("nuts_device needs device (HIP) tensors; use nuts() on the CPU")
    lib = _lib.load()
    q = q0.clone().contiguous()
    Cn, D = q.shape
    dev, dt = q.device, q.dtype
    K = uniforms_per_iteration(max_tree_depth)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    U, g = potential_and_grad(q)
    if not torch.isfinite(U).all():
        raise ValueError("nuts: the potential is not finite at the initial positions")
    U, g = U.clone().contiguous(), g.clone().contiguous()
    ad = _Adapter(Cn, D, dt, dev, step_size, target_accept, num_warmup, adapt_mass_matrix)
    vec = lambda: torch.empty((Cn, D), dtype=dt, device=dev)
    sca = lambda: torch.empty(Cn, dtype=dt, device=dev)
    flg = lambda: torch.zeros(Cn, dtype=torch.int32, device=dev)
    t = {n: vec() for n in ("ql", "pl", "gl", "qr", "pr", "gr", "q_prop", "g_prop", "p_sum", "qe", "pe", "ge",
                            "sub_psum", "sq", "sg", "ph", "qn")}
    t.update({n: sca() for n in ("U_prop", "logW", "H0", "sum_acc", "n_prop", "accept", "es", "sub_logW", "sU")})
    t.update({n: flg() for n in ("active", "diverged", "sub_stop", "building", "right")})
    t["ck_p"] = torch.empty((max_tree_depth + 1, Cn, D), dtype=dt, device=dev)
    t["ck_s"] = torch.empty((max_tree_depth + 1, Cn, D), dtype=dt, device=dev)
    t["flags"] = torch.zeros(2, dtype=torch.int32, device=dev)
    t["z"], t["unif"], t["eps"] = vec(), torch.empty((Cn, K), dtype=dt, device=dev), sca()
    b = _Buffers()
    for n, ten in t.items():
        setattr(b, n, ten.data_ptr())
    b.q, b.U, b.g = q.data_ptr(), U.data_ptr(), g.data_ptr()
    b.ldu = K
    dtc = _dt(q)
    bp = C.byref(b)

    # the stream the caller is on when the run starts carries every launch of the run (looking the current
    # stream up per call cost more host time than the launch itself: 62 k look-ups in a 200-iteration run)
    run_stream = torch.cuda.current_stream(dev)
    h_stream = C.c_void_p(run_stream.cuda_stream)
    nuts_op = lib.gpvae_nuts_op

    def op(code, depth=0, leaf=0, lo=0, hi=0):
        rc = nuts_op(h_stream, dtc, code, Cn, D, max_tree_depth, depth, leaf, lo, hi, bp)
        if rc:
            check(rc, "gpvae_nuts_op")

    # potentials of this package write into the sampler's own buffers (no allocation per leapfrog step)
    eval_into = getattr(potential_and_grad, "eval_into", None)
    if eval_into is not None:
        Un_buf, gn_buf = sca(), vec()
        b.Un, b.gn, b.ldg = Un_buf.data_ptr(), gn_buf.data_ptr(), gn_buf.stride(0)

    # a potential that brings its own fused iteration (VAEInference.regression: gpvae_nuts_vae_iteration) runs the
    # whole transition — every doubling and leapfrog step of every chain — as ONE launch per iteration
    iterate = getattr(potential_and_grad, "nuts_iteration", None) if fused else None
    if iterate is not None and (eval_into is None):
        Un_buf, gn_buf = sca(), vec()  # the fused kernel's scratch
        b.Un, b.gn, b.ldg = Un_buf.data_ptr(), gn_buf.data_ptr(), gn_buf.stride(0)

    keep = []
    acc_stat = torch.zeros(Cn, dtype=dt, device=dev)
    n_div = torch.zeros((), dtype=torch.long, device=dev)
    n_leap = torch.zeros(Cn, dtype=dt, device=dev)
    for it in range(num_warmup + num_samples):
        # the torch-side operations of an iteration (noise and step-size copies, flag polls and clears, adapter
        # updates) must be ordered with the gpvae_nuts_op launches, which go to the stream captured above: one
        # look-up per iteration makes sure a potential or noise_fn has not left another stream current
        if torch.cuda.current_stream(dev).cuda_stream != run_stream.cuda_stream:
            raise _lib.GpvaeError("nuts_device: the current HIP stream changed during the run (potential_and_grad / "
                                  "noise_fn must restore the stream they were called on)")
        warm = it < num_warmup
        t["eps"].copy_(ad.step_size(warm))
        minv = ad.minv.contiguous()
        b.minv = minv.data_ptr()
        if noise_fn is None:
            t["z"].copy_(torch.randn((Cn, D), dtype=dt, device=dev, generator=gen))
            t["unif"].copy_(torch.rand((Cn, K), dtype=dt, device=dev, generator=gen))
        else:
            z, unif = noise_fn(it)
            t["z"].copy_(z)
            t["unif"].copy_(unif)
        if iterate is not None:
            iterate(h_stream, dtc, Cn, max_tree_depth, bp)
        else:
            op(0)
        for depth in range(max_tree_depth if iterate is None else 0):
            op(1, depth)
            leaves = 1 << depth
            for leaf in range(leaves):
                if leaf == 0:
                    op(2, depth)  # PRE; the later leaves' PRE rides in the previous leaf's POST_PRE launch
                if eval_into is not None:
                    eval_into(t["qn"], Un_buf, gn_buf, h_stream)
                else:
                    Un, gn = potential_and_grad(t["qn"])
                    if gn.stride(1) != 1 or not Un.is_contiguous():
                        Un, gn = Un.contiguous(), gn.contiguous()
                    b.Un, b.gn, b.ldg = Un.data_ptr(), gn.data_ptr(), gn.stride(0)
                lo, hi = _ckpt_range(leaf)
                op(3 if leaf + 1 == leaves else 6, depth, leaf, lo, hi)
                if leaves >= 8 and (leaf & 3) == 3 and leaf + 1 < leaves and int(t["flags"][0].item()) == 0:
                    break
                if leaves >= 8 and (leaf & 3) == 3:
                    t["flags"][:1].zero_()  # only chains that still build after the next leaves set it again
            op(4, depth)
            if int(t["flags"][1].item()) == 0:
                break
        if iterate is None:
            op(5)
        a = t["accept"]
        if warm:
            ad.update(it, a.clone(), q.clone())
        else:
            acc_stat += a
            n_div += t["diverged"].sum()
            n_leap += t["n_prop"]
            if (it - num_warmup) % thinning == 0:
                keep.append(q.clone())
    samples = torch.stack(keep) if keep else q.new_empty((0, Cn, D))
    info = {"accept_prob": acc_stat / max(num_samples, 1), "step_size": ad.step_size(False),
            "inverse_mass": ad.minv, "divergent": int(n_div.item()),
            "leapfrogs_per_iteration": n_leap / max(num_samples, 1)}
    return samples, info


class _Run(C.Structure):
    """gpvae_nuts_run of include/gpvae.h."""
    _fields_ = [("num_warmup", C.c_int32), ("num_samples", C.c_int32), ("thinning", C.c_int32), ("adapt_mass", C.c_int32),
                ("max_depth", C.c_int32), ("win_start", C.c_int32), ("n_ends", C.c_int32), ("ends", C.c_int32 * 24),
                ("seed", C.c_uint64), ("stream_base", C.c_uint64), ("row_len", C.c_int64), ("chain_ids", C.c_void_p),
                ("step_size", C.c_double), ("target_accept", C.c_double), ("q0", C.c_void_p), ("samples", C.c_void_p),
                ("accept_prob", C.c_void_p), ("leapfrogs", C.c_void_p), ("step_size_out", C.c_void_p),
                ("inverse_mass", C.c_void_p), ("divergent", C.c_void_p)]


def nuts_device_run(potential, q0, num_warmup, num_samples, thinning=1, max_tree_depth=10, step_size=0.1,
                    target_accept=0.8, noise_seed=0, noise_stream_base=0, chain_ids=None, adapt_mass_matrix=True):
    """The whole run — warm-up with its adaptation and sampling, every chain on its own — as ONE launch, for
    potentials that bring a `nuts_run` hook (VAEInference.regression with a decoder head that fits LDS:
    gpvae_nuts_vae_run).  Same algorithm and the same random numbers as :func:`nuts_device` driven with the Philox
    noise of Inference._mcmc: element (chain_ids[c], k) of the [chains, D + uniforms_per_iteration] normal matrix of
    stream `noise_stream_base | iteration`, uniforms = Phi(normal).  The adaptation arithmetic runs per chain in the
    chains' dtype inside the kernel; the per-iteration drivers do it as torch operations, so the two agree to
    rounding at first and like two runs of one sampler with slightly different step sizes after long warm-ups.
    Returns (samples [ceil(num_samples / thinning), C, D], info) as nuts_device does."""
    from . import _lib
    from ._lib import check
    from .ops import _dt
    run = getattr(potential, "nuts_run", None)
    if run is None:
        raise _lib.GpvaeError("nuts_device_run: this potential has no whole-run kernel (use nuts_device)")
    q0 = q0.contiguous()
    Cn, D = q0.shape
    dev, dt = q0.device, q0.dtype
    win = _Windows(num_warmup if adapt_mass_matrix else 0)
    if len(win.ends) > 24:
        raise _lib.GpvaeError("nuts_device_run: more than 24 mass-matrix windows")
    kept = (num_samples + thinning - 1) // thinning
    samples = torch.empty((kept, Cn, D), dtype=dt, device=dev)
    acc, leap, eps = (torch.empty(Cn, dtype=dt, device=dev) for _ in range(3))
    minv = torch.empty((Cn, D), dtype=dt, device=dev)
    ndiv = torch.zeros(Cn, dtype=torch.int32, device=dev)
    ids = None if chain_ids is None else chain_ids.to(device=dev, dtype=torch.long).contiguous()
    r = _Run()
    r.num_warmup, r.num_samples, r.thinning, r.adapt_mass = int(num_warmup), int(num_samples), int(thinning), int(bool(win.ends))
    r.max_depth, r.win_start, r.n_ends = int(max_tree_depth), int(win.start), len(win.ends)
    for i, e in enumerate(win.ends):
        r.ends[i] = int(e)
    r.seed, r.stream_base = int(noise_seed) & 0xFFFFFFFFFFFFFFFF, int(noise_stream_base) & 0xFFFFFFFFFFFFFFFF
    r.row_len = D + uniforms_per_iteration(max_tree_depth)
    r.chain_ids = None if ids is None else ids.data_ptr()
    r.step_size, r.target_accept = float(step_size), float(target_accept)
    r.q0, r.samples = q0.data_ptr(), samples.data_ptr()
    r.accept_prob, r.leapfrogs, r.step_size_out = acc.data_ptr(), leap.data_ptr(), eps.data_ptr()
    r.inverse_mass, r.divergent = minv.data_ptr(), ndiv.data_ptr()
    run(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), _dt(q0), Cn, C.byref(r))
    info = {"accept_prob": acc, "step_size": eps, "inverse_mass": minv, "divergent": int(ndiv.sum().item()),
            "leapfrogs_per_iteration": leap}
    return samples, info
