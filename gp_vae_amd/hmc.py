"""Batched Hamiltonian Monte Carlo over a potential-and-gradient callable.

The reference runs numpyro.infer.NUTS (src/model/inference.py:84-99, 477-490); NumPyro is third-party
code that is absent here, so `Inference.run_mcmc` drives this sampler instead: plain HMC with an
identity mass matrix, a jittered number of leapfrog steps (uniform in [L/2, L] per iteration, which
breaks the resonances of fixed-length trajectories) and per-chain step-size adaptation by dual
averaging (Hoffman & Gelman 2014, alg. 5) during warm-up.  Same target density as NUTS (the model's
potential in NumPyro's unconstrained space), different trajectory rule — stated, not hidden.

All chains advance together: one potential-and-gradient evaluation per leapfrog step for the whole
batch, which is how the HIP kernels behind it (decoder fwd/VJP, gpvae_mvn_logp) want to be called.
"""
import math
import random

import torch


def hmc(potential_and_grad, q0, num_warmup, num_samples, thinning=1, num_leapfrog=16, step_size=0.05,
        target_accept=0.8, seed=0, noise_fn=None):
    """potential_and_grad(q [C, D]) -> (U [C], dU/dq [C, D]).  q0: [C, D] initial positions.

    Returns (samples [num_samples // thinning, C, D], info) where info holds the mean acceptance
    probability per chain over the sampling phase, the adapted step sizes and the number of
    divergent (non-finite energy) proposals.

    noise_fn (optional): noise_fn(it) -> (momenta [C, D], uniforms [C]) for iteration `it`.  The
    inference drivers pass one whose rows depend only on (seed, global chain id, it), so a chain's
    trajectory does not depend on which other chains share its batch — chains sharded over ranks
    reproduce the single-process run.  Default: one torch generator for the whole batch."""
    q = q0.clone()
    C, D = q.shape
    dev, dt = q.device, q.dtype
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    host_rng = random.Random(int(seed))
    U, g = potential_and_grad(q)
    if not torch.isfinite(U).all():
        raise ValueError("hmc: the potential is not finite at the initial positions")
    # dual averaging state (per chain)
    log_eps = torch.full((C,), math.log(step_size), dtype=dt, device=dev)
    log_eps_bar = log_eps.clone()
    h_bar = torch.zeros(C, dtype=dt, device=dev)
    mu = log_eps + math.log(10.0)
    gamma, t0, kappa = 0.05, 10.0, 0.75
    keep = []
    acc_sum = torch.zeros(C, dtype=dt, device=dev)
    divergent = 0
    for it in range(num_warmup + num_samples):
        warm = it < num_warmup
        eps = torch.exp(log_eps if warm else log_eps_bar)[:, None]
        if noise_fn is None:
            p = torch.randn((C, D), dtype=dt, device=dev, generator=gen)
            unif = torch.rand(C, dtype=dt, device=dev, generator=gen)
        else:
            p, unif = noise_fn(it)
        h0 = U + 0.5 * (p * p).sum(1)
        qn, gn = q, g
        pn = p - 0.5 * eps * gn
        Un = U
        nleap = host_rng.randint(max(1, num_leapfrog // 2), num_leapfrog)
        for s in range(nleap):
            qn = qn + eps * pn
            Un, gn = potential_and_grad(qn)
            pn = pn - (eps if s + 1 < nleap else 0.5 * eps) * gn
        h1 = Un + 0.5 * (pn * pn).sum(1)
        dh = h0 - h1
        bad = ~torch.isfinite(dh)
        divergent += int(bad.sum().item())
        a = torch.where(bad, torch.zeros_like(dh), torch.exp(torch.clamp(dh, max=0.0)))
        take = unif < a
        q = torch.where(take[:, None], qn, q)
        g = torch.where(take[:, None], gn, g)
        U = torch.where(take, Un, U)
        if warm:
            m = it + 1
            h_bar = (1 - 1 / (m + t0)) * h_bar + (target_accept - a) / (m + t0)
            log_eps = mu - math.sqrt(m) / gamma * h_bar
            w = m ** (-kappa)
            log_eps_bar = w * log_eps + (1 - w) * log_eps_bar
        else:
            acc_sum += a
            if (it - num_warmup) % thinning == 0:
                keep.append(q.clone())
    samples = torch.stack(keep) if keep else q.new_empty((0, C, D))
    info = {"accept_prob": acc_sum / max(num_samples, 1), "step_size": torch.exp(log_eps_bar),
            "divergent": divergent}
    return samples, info
