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
    minv = torch.ones((C, D), dtype=dt, device=dev)  # inverse mass diagonal, per chain
    log_eps = torch.full((C,), math.log(step_size), dtype=dt, device=dev)
    log_eps_bar = log_eps.clone()
    h_bar = torch.zeros(C, dtype=dt, device=dev)
    mu = log_eps + math.log(10.0)
    gamma, t0, kappa = 0.05, 10.0, 0.75
    m_da = 0
    win = _Windows(num_warmup if adapt_mass_matrix else 0)
    w_n, w_mean, w_m2 = 0, torch.zeros_like(q), torch.zeros_like(q)
    keep = []
    acc_stat = torch.zeros(C, dtype=dt, device=dev)
    n_div = torch.zeros((), dtype=torch.long, device=dev)
    n_leap = torch.zeros(C, dtype=dt, device=dev)
    inf = torch.full((C,), float("inf"), dtype=dt, device=dev)

    for it in range(num_warmup + num_samples):
        warm = it < num_warmup
        eps = torch.exp(log_eps if warm else log_eps_bar)
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
            m_da += 1
            h_bar = (1 - 1 / (m_da + t0)) * h_bar + (target_accept - a) / (m_da + t0)
            log_eps = mu - math.sqrt(m_da) / gamma * h_bar
            w = m_da ** (-kappa)
            log_eps_bar = w * log_eps + (1 - w) * log_eps_bar
            if win.ends and it >= win.start:
                w_n += 1
                d1 = q - w_mean
                w_mean = w_mean + d1 / w_n
                w_m2 = w_m2 + d1 * (q - w_mean)
                if it in win.ends and w_n > 1:
                    var = w_m2 / (w_n - 1)
                    minv = (w_n / (w_n + 5.0)) * var + 1.0e-3 * (5.0 / (w_n + 5.0))
                    w_n, w_mean, w_m2 = 0, torch.zeros_like(q), torch.zeros_like(q)
                    # restart the step-size search around the last averaged value
                    log_eps = log_eps_bar.clone()
                    mu = log_eps + math.log(10.0)
                    h_bar = torch.zeros_like(h_bar)
                    m_da = 0
        else:
            acc_stat += a
            n_div += diverged.sum()
            n_leap += n_prop
            if (it - num_warmup) % thinning == 0:
                keep.append(q.clone())
    samples = torch.stack(keep) if keep else q.new_empty((0, C, D))
    info = {"accept_prob": acc_stat / max(num_samples, 1), "step_size": torch.exp(log_eps_bar),
            "inverse_mass": minv, "divergent": int(n_div.item()),
            "leapfrogs_per_iteration": n_leap / max(num_samples, 1)}
    return samples, info
