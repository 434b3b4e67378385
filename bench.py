#!/usr/bin/env python3
"""bench.py — GP-prior VAE training draws/s at the 64x64 grid (BASELINE.json metric, config C3).

One "step" = one full training step on one synthetic batch, nothing skipped:
  draw (ls, var, sigma) -> Gram 4096x4096 -> Cholesky -> B = 4096 MVN draws y = L eps + sigma eps'
  -> VAE ELBO forward + backward (2d reference layers 50/30/15) -> [all-reduce] -> Adam.
float32 throughout (the reference's dtype).  Data are synthetic by construction (GP draws).
N > 1: one process per GPU (torch.distributed "nccl" = RCCL), every rank draws its own GP and its
own B functions (weak scaling), one gradient all-reduce per step.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is for the draw kernel (the
largest single kernel of the step: n^2*B flop on f32 MFMA), timed with HIP events on the stream it
runs on inside the timed region.  `cpu_baseline` is the oracle's NumPy/LAPACK restatement of the
same step timed on this box's host cores (rank 0, N = 1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

# One hardware queue per stream in use (main, factor stream, RCCL's stream, two side streams of the
# Cholesky micro-benchmark): with HIP's default of 4 the collective's stream shares a queue with the
# factor stream, its packets wait behind a whole factor group and the overlap is lost.
# Read by the HIP runtime when it initialises, so it must be set before the first HIP call.
# (The GPU boxes export 4, HIP's default, so raise rather than setdefault.)
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_F64_MFMA_TFLOPS = 78.6   # dense f64 matrix peak
PEAK_HBM_GBS = 8000.0         # HBM3E, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak (no sparsity), MI355X_MICROARCH.md
# what v_mfma_f32_32x32x16_bf16 sustains on this chip with random operand bits and no memory traffic at all
# (tools/ub_mfma_power.hip; 2.45 PFLOP/s with constant operands): power management.  1.81 PFLOP/s on the round-2 box
# (profiles/r02_mfma_power.txt), 1.90 on the round-3 one (profiles/r03_mfma_shape_power.txt): devices differ
SUSTAINED_BF16_MFMA_RANDOM_TFLOPS = 1900.0
GRID = 64
N = GRID * GRID
B = 4096
DIMS = (50, 30, 15)  # 2d_examples.py hidden_dims / z_dim
JITTER = 1.0e-3      # 2d_examples.py:503
# HIP events bracket every 3rd launch of the draw in the timed region (roofline.avg_launch_ms): an event record
# costs the stream ~5.6 us (kernel trace), so timing every launch took up to 11 us out of every step (measured:
# 0.775-0.778 ms per step with every launch timed, 0.770 with every 3rd).  3 is coprime to the factor group of 16:
# every 4th launch would sample the group's first step — the one beside the factor stream's Gram build — four
# times as often as it occurs (0.505 against 0.444 ms in-loop).
DRAW_EVENTS_EVERY = int(os.environ.get("GPVAE_DRAW_EVENTS_EVERY", "3"))
LR = 1.0e-3


def grid_points():
    g = np.arange(0, 1, 1 / GRID)
    u, v = np.meshgrid(g, g)  # 2d_examples.py:562-564
    return np.stack([u.ravel(), v.ravel()], 1).astype(np.float32)


def bench_hyper(rng):
    """(ls, var, sigma) from the reference's priors, verbatim (vae.py:157-159): InverseGamma(1, .1),
    LogNormal(0, .1), HalfNormal(.1).  (An earlier revision clipped ls: the heavy tail of the
    InverseGamma produces nearly rank-one Gram matrices, which the factorisation now handles with
    LAPACK-level backward error — 0 failures in 256 unclipped draws at this shape.)"""
    from gp_vae_amd.gp import draw_hyper
    ls, var, sigma = draw_hyper(rng)
    return float(ls), float(var), float(sigma)


def cpu_baseline(steps=12, budget_s=12.0):
    """The oracle's train_step (NumPy/SciPy, multi-threaded LAPACK) at the same sizes."""
    from oracle import gpvae_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    rng = np.random.default_rng(0)
    x = grid_points()
    p = orc.init_params(rng, N, *DIMS, np.float32)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(a) for k, a in p.items()}
    t_used = 0.0
    done = 0
    for t in range(1, steps + 1):
        hyper = bench_hyper(rng)
        E = rng.standard_normal((B, N), dtype=np.float32)
        E2 = rng.standard_normal((B, N), dtype=np.float32)
        ez = rng.standard_normal((B, DIMS[2]), dtype=np.float32)
        t0 = time.perf_counter()
        _, p, m, v = orc.train_step(p, m, v, t, x, hyper, E, E2, ez, LR, JITTER)
        t_used += time.perf_counter() - t0
        done += 1
        if t_used > budget_s:  # a bounded sample: about 10-15 s of CPU work
            break
    return {"value": B * done / t_used, "unit": "draws/s", "cores": int(cores), "kind": "port",
            "sample": "%d full training steps (n=4096, B=4096, f32) of oracle.train_step, "
                      "normals pre-generated, %.1f s" % (done, t_used)}


def pmc_traffic(b3):
    """HBM bytes per draw launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of tools/prof_draw.py at this shape, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950): profiles/r03_draw_b3_pmc.json for the bf16 x 3 kernel, r01_draw_pmc.json for
    the f32-MFMA kernel.  bench.py cannot collect counters itself; null when no pass is committed."""
    name = "r03_draw_b3_pmc.json" if b3 else "r01_draw_pmc.json"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_launch"]), "profiles/" + name
    except (OSError, KeyError, ValueError):
        return None, None


def potrf_microbench(ops, x, batch=32, iters=5, dtype=torch.float32, parts=1):
    """Batched Cholesky TFLOP/s: `batch` Gram matrices with ls on the ladder linspace(0.05, 1, batch),
    var = 1 (batch 32).  parts > 1: the batch is issued as that many sub-batches
    on side streams (ops.potrf_overlapped_), timed fork to join on the current stream."""
    dev = x.device
    x = x.to(dtype)
    ls = torch.linspace(0.05, 1.0, batch, device=dev, dtype=dtype)
    var = torch.ones(batch, device=dev, dtype=dtype)
    Kb = torch.empty((batch, N, N), device=dev, dtype=dtype)
    info = None
    times = []
    for it in range(iters + 1):
        ops.gram_se(x, x, var, ls, JITTER, out=Kb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if parts > 1:
            info = ops.potrf_overlapped_(Kb, parts=parts, zero_upper=False, nan_on_fail=False)
        else:
            info = ops.potrf_(Kb, zero_upper=False, nan_on_fail=False)
        e1.record()
        torch.cuda.synchronize()
        if it > 0:
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(times))
    return {"batch": batch, "n": N, "streams": parts, "dtype": "f32" if dtype == torch.float32 else "f64", "ms": t * 1e3,
            "tflops": batch * N ** 3 / 3 / t / 1e12, "all_pd": bool((info == 0).all().item())}


def _event_time(f, reps):
    """Mean seconds per call of f() between HIP events on the current stream (one untimed call first)."""
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def extras(ops, x, dev, out=None):
    """Bounded side measurements under the driver's clock (about 5 s in all, rank 0, N = 1):
    the Gram builder's achieved HBM write rate (its bound), and one number each for
    the other BASELINE.json configurations C2, C4 and C5.  Their parity is the tests' business."""
    import math
    from gp_vae_amd.vae import StepEngine, SVIState, init_params, decoder_list_from_flat
    from gp_vae_amd.inference import VAEInference
    out = {} if out is None else out  # filled as it goes: what was measured before a failure stays in the line
    # a1: Gram build at the training factor group's shape (16 x 4096 x 4096), f32 and f64
    for dt, tag in ((torch.float32, "gram"), (torch.float64, "gram_f64")):
        xx = x.to(dt)
        ls = torch.linspace(0.05, 1.0, 16, device=dev, dtype=dt)
        var = torch.ones(16, device=dev, dtype=dt)
        Kb = torch.empty((16, N, N), device=dev, dtype=dt)
        t = _event_time(lambda: ops.gram_se(xx, xx, var, ls, JITTER, out=Kb), 10)
        nbytes = 16 * N * N * Kb.element_size()
        out[tag] = {"kernel": "gram_se_kernel", "batch": 16, "n": N, "bytes": nbytes, "ms": t * 1e3,
                    "GB_per_s": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 1e9 / PEAK_HBM_GBS,
                    "bound": "hbm write"}
        del Kb
    # a2: aggregated (cell-averaged) kernel, 32 x 32 cells of 16 sample points each: 2.7e8 kernel evaluations
    # for a 4 MiB matrix — bound by the vector ALU's sqrt / exp issue, not by memory
    pa = ops.randn((1024, 16, 2), torch.float32, dev, seed=3) * 0.01 + x[::4, None, :].to(torch.float32)
    t = _event_time(lambda: ops.gram_agg(pa, 1.0, 0.3, 1e-6), 20)
    out["gram_agg"] = {"kernel": "gram_agg_kernel", "n_cells": 1024, "m": 16, "ms": t * 1e3,
                       "kernel_evals_per_s": 1024 * 1024 * 256 / t, "bound": "valu (sqrt + exp per evaluation)"}
    # C2: 1D n = 1024, B = 4096, reference MLP 35/30/10, whole training step
    x1 = torch.arange(1024, dtype=torch.float32, device=dev) / 1024
    eng = StepEngine(x1, 1e-4, (35, 30, 10), 4096, LR, dtype=torch.float32, seed=1, group=16)
    st = SVIState(init_params(1024, 35, 30, 10, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(32)], st, train=True)
    torch.cuda.synchronize()
    time.sleep(2.0)  # right after the sustained MFMA load above the chip runs this light step at half speed
    t = None         # (power management; it recovers within 2 s of idling: tools/c2exp.py)
    for rep in (1, 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run([(rep, i) for i in range(256)], st, train=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 256
        t = dt if t is None else min(t, dt)
    out["c2"] = {"workload": "C2: 1D n=1024, B=4096, VAE 1024-35-30-10, f32, full training step",
                 "ms_per_step": t * 1e3, "draws_per_s": 4096 / t, "sample": "best of 2 runs of 256 steps"}
    del eng, st
    # C4: the n_obs = 32768 f64 factorisation (8 GiB in place), SE Gram of a 2D point set + nugget
    n4 = 32768
    g = torch.arange(192, dtype=torch.float64, device=dev) / 192
    u, v = torch.meshgrid(g, g, indexing="xy")
    pts = torch.stack([u.reshape(-1), v.reshape(-1)], 1)
    idx = torch.tensor(np.sort(np.random.default_rng(0).permutation(192 * 192)[:n4]), device=dev)
    A = ops.gram_se(pts[idx].contiguous(), pts[idx].contiguous(), 1.0, 0.2, 1e-5)
    A.diagonal().add_(0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    info = ops.potrf_(A, zero_upper=False, nan_on_fail=False)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    out["c4_potrf_f64"] = {"workload": "C4: Cholesky of the 32768 x 32768 f64 kriging matrix (192x192 grid subset)",
                           "n": n4, "s": t, "tflops": n4 ** 3 / 3 / t / 1e12,
                           "frac_of_mfma_peak": n4 ** 3 / 3 / t / 1e12 / PEAK_F64_MFMA_TFLOPS,
                           "pd": bool((info == 0).all().item())}
    del A
    # C5: VAE-prior regression potential + gradient, 1024 chains, decoder 10-30-35-300
    n5, dims5 = 300, (35, 30, 10)
    dec = decoder_list_from_flat(init_params(n5, *dims5, 1, torch.float32, dev), n5, *dims5)
    x5 = np.arange(0, 1, 1 / n5)
    inf = VAEInference(None, dec, dims5[2], 1, x5, np.sin(6 * x5), [20, 23, 100, 110, 117, 130, 133, 140, 170, 190],
                       {}, dtype=torch.float32)
    z = ops.randn((1024, dims5[2]), torch.float32, dev, seed=5)
    lsig = torch.full((1024,), math.log(0.1), device=dev)
    t = _event_time(lambda: inf.potential_and_grad(z, lsig), 200)
    out["c5"] = {"workload": "C5: VAEInference.regression potential + gradient, 1024 chains, decoder 10-30-35-300, f32",
                 "grad_evals_per_s": 1024 / t, "ms_per_batched_eval": t * 1e3}
    # the same model under the batched NUTS (gpvae_nuts_op transition + gpvae_vae_potential), end to end
    y5 = np.full(n5, np.nan)
    y5[[20, 23, 100, 110, 117, 130, 133, 140, 170, 190]] = np.sin(6 * x5[[20, 23, 100, 110, 117, 130, 133, 140, 170, 190]])
    inf = VAEInference(None, dec, dims5[2], 1, x5, y5, [20, 23, 100, 110, 117, 130, 133, 140, 170, 190],
                       {"num_warmup": 100, "num_samples": 100, "num_chains": 1024, "thinning": 1}, seed=1, dtype=torch.float32)
    walls = []
    for rep in range(2):  # the first call of a process also loads the kernels' code objects: both are reported
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inf.run_mcmc(3)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    t = walls[1]
    lf = float(inf.mcmc_info["leapfrogs_per_iteration"].mean().item())
    out["c5_nuts"] = {"workload": "C5 end to end: run_mcmc (warm-up with adaptation + sampling + unpacking the sites), batched NUTS, "
                                  "1024 chains, 100 warm-up + 100 sampling iterations; the whole run is one launch of "
                                  "gpvae_nuts_vae_run",
                      "wall_s": t, "wall_s_first_call_of_the_process": walls[0],
                      "chain_iterations_per_s": 1024 * 200 / t, "leapfrogs_per_iteration": lf,
                      "accept_prob": float(inf.mcmc_info["accept_prob"].mean().item()),
                      "divergent": int(inf.mcmc_info["divergent"])}
    # f2: the reference's vmap of gp_prediction over posterior draws (inference.py:511-521): S = 8000 draws on the 25 x 25
    # grid, n_obs = 10 (toy_examples.py:199) and 100 observed points; one workgroup per problem, everything in LDS
    g25 = np.arange(25) / 25.0
    u25, v25 = np.meshgrid(g25, g25)
    x25 = np.stack([u25.ravel(), v25.ravel()], 1).astype(np.float32)
    rng = np.random.default_rng(0)
    S = 8000
    hv, hl, hs = rng.lognormal(0, 0.1, S), rng.uniform(0.05, 0.5, S), np.abs(rng.normal(0, 0.1, S)) + 1e-3
    f2 = {}
    for n_obs in (10, 100):
        obs = np.sort(rng.permutation(625)[:n_obs])
        mask = np.ones(625, bool); mask[obs] = False
        yy = np.sin(4 * x25.sum(1)) + 0.1 * rng.standard_normal(625)
        xo, xt = torch.tensor(x25[obs], device=dev), torch.tensor(x25[mask], device=dev)
        yo = torch.tensor(yy[obs], dtype=torch.float32, device=dev)
        xi = ops.randn((S, int(mask.sum())), torch.float32, dev, seed=9)
        t = _event_time(lambda: ops.krig_predict_batched(xo, xt, yo, hv, hl, hs, 1e-5, xi), 3)
        info = ops.krig_predict_batched(xo, xt, yo, hv, hl, hs, 1e-5, xi)[3]
        f2["n_obs_%d" % n_obs] = {"problems_per_s": S / t, "ms": t * 1e3, "n_test": int(mask.sum()),
                                  "all_pd": bool((info == 0).all().item())}
    out["f2_kriging_batched"] = dict(f2, workload="f2: gpvae_krig_predict_batched, S = 8000 hyper-parameter draws, 25 x 25 grid, f32; "
                                                  "one workgroup per problem, Gram matrix / factor / inverse resident in LDS")
    # C3 again with the opt-in fp16 x 2 draw and factorisation (StepEngine(mfma="f16x2"): operands of the two
    # matrix products rounded to 22 bits, errors at the float32 level — tests/test_gpu_seams.py::test_h2_*,
    # test_potrf_h2_*): the same full training step as the headline, which stays on the operand-exact bf16 x 3
    eng = StepEngine(x, JITTER, DIMS, B, LR, dtype=torch.float32, seed=1, hyper_sampler=bench_hyper, group=16, mfma="f16x2")
    st = SVIState(init_params(N, *DIMS, 1, torch.float32, dev))
    eng.run([(0, i) for i in range(20)], st, train=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run([(1, i) for i in range(400)], st, train=True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 400
    out["c3_f16x2"] = {"workload": "C3 full training step with mfma='f16x2' (opt-in: fp16 x 2 draw + factorisation)",
                       "ms_per_step": t * 1e3, "draws_per_s": B / t, "sample": "400 steps after 20 warm-up steps",
                       "failed_factors": eng.failed_factors()}
    del eng, st
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, BEFORE this process has made any HIP call (a process
    that touched the GPU is never re-executed).  Rank 0's stdout carries the JSON line; the exit code
    is non-zero if any rank failed."""
    import subprocess
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": os.environ.get("MASTER_ADDR", "127.0.0.1"),
                "MASTER_PORT": os.environ.get("MASTER_PORT") or str(free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = None
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.time() + 30.0  # a rank died: the others would hang in a collective
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    bad = [(r, rc) for r, rc in enumerate(p.returncode for p in procs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        raise SystemExit(1)


def dry_run(args, world, rank):
    """--dry-run: the launcher, the rendezvous and the one collective of the path on gloo/CPU with a
    stand-in gradient; no HIP call, no throughput.  Exists so that `--gpus N` is testable without GPUs."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from gp_vae_amd import parallel
    g = torch.full((1024,), float(rank + 1))
    h = parallel.allreduce_sum_start(g)
    parallel.allreduce_wait(h)
    want = world * (world + 1) / 2
    assert float(g[0]) == want and float(g[-1]) == want, (float(g[0]), want)
    seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "GP-prior VAE training draws/sec at 64x64 grid", "value": None,
                          "unit": "draws/s", "n_gpus": world, "rccl_world": seen, "dry_run": True,
                          "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "allreduce_sum_checked": want}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--group", type=int, default=16, help="steps whose factors are built as one batch")
    ap.add_argument("--first-group", type=str, default="", help="sizes of the first factor groups, e.g. 4,4,8 (a ramp; then full groups); empty = the engine's default")
    ap.add_argument("--mfma", choices=["auto", "f32", "bf16x3", "f16x2"], default="auto",
                    help="matrix pipe of the float32 draws: bf16x3 (default where shapes allow) or the f32 MFMA")
    ap.add_argument("--hyper", choices=["host", "device"], default="device",
                    help="where the step's (ls, var, sigma) are drawn from the reference's priors (vae.py:157-159): "
                         "gpvae_hyper_sample on the factor stream (default), or host NumPy with 3 floats uploaded per "
                         "step (measured 3 %% slower at 1000 steps, 1-2 %% at 20: DESIGN.md §7)")
    ap.add_argument("--force-collective", action="store_true",
                    help="single GPU: create a one-rank RCCL group and run the per-step all-reduce anyway")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher + rendezvous + all-reduce on gloo/CPU only (no GPU work, no number)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the bounded side measurements (Gram HBM rate, C2, C4 potrf, C5)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])  # parent: no HIP call has been made
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d — refusing to label a "
                         "%d-rank run as %d GPUs" % (args.gpus, world, world, args.gpus))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from gp_vae_amd import ops, parallel
    parallel.FORCE_COLLECTIVE = bool(args.force_collective)
    from gp_vae_amd.vae import StepEngine, SVIState, init_params

    dev = torch.device("cuda", local)
    x = torch.tensor(grid_points(), device=dev)
    eng = StepEngine(x, JITTER, DIMS, B, LR, dtype=torch.float32, seed=1, overlap=not args.no_overlap,
                     hyper_sampler=bench_hyper if args.hyper == "host" else "device", group=args.group, mfma=args.mfma,
                     first_group=[int(v) for v in args.first_group.split(",")] if args.first_group else None)
    state = SVIState(init_params(N, *DIMS, 1, torch.float32, dev))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.run([(0, i) for i in range(args.warmup)], state, train=True)
    eng.draw_events = []
    eng.draw_events_every = DRAW_EVENTS_EVERY  # the draw's launches are timed live, every 3rd of them
    barrier()
    t0 = time.perf_counter()
    eng.run([(1, i) for i in range(args.steps)], state, train=True)
    host_enqueue = time.perf_counter() - t0  # the host's share: all launches of the timed steps issued
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3] * 2  # per-rank ms/step: min, max over ranks
    if world > 1:
        tt = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rank_ms = [-float(tt[1].item()) / args.steps * 1e3, float(tt[0].item()) / args.steps * 1e3]
        elapsed = float(tt[0].item())
    rccl_world = dist.get_world_size() if dist.is_initialized() else 1
    draw_ms = float(np.mean([a.elapsed_time(b) for a, b in eng.draw_events])) if eng.draw_events else None
    draw_timed = len(eng.draw_events) if eng.draw_events else 0
    eng.draw_events = None
    eng.draw_events_every = 1
    loss_last = float(eng.loss.item())
    draw_alone_ms = eng.time_draw_alone() if rank == 0 else None

    # batched Cholesky microbenchmark: every rank factors its own batch (weak), TFLOP/s summed
    chol = [potrf_microbench(ops, x, batch=32), potrf_microbench(ops, x, batch=args.group),
            potrf_microbench(ops, x, batch=16, dtype=torch.float64)]
    # the two-stream forms only without a process group: with RCCL's streams in the process the side
    # streams' hardware queues are no longer their own (measured: 68 instead of 80 TF)
    two_streams = world == 1 and not args.force_collective
    if two_streams:
        chol += [potrf_microbench(ops, x, batch=32, parts=2), potrf_microbench(ops, x, batch=64, parts=2),
                 potrf_microbench(ops, x, batch=16, dtype=torch.float64, parts=2)]
    if world > 1:
        for c in chol:
            tt = torch.tensor([c["tflops"], float(c["all_pd"])], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            c["tflops"] = float(tt[0].item())
            c["all_pd"] = bool(tt[1].item() == world)
    for c in chol:
        peak = PEAK_F32_MFMA_TFLOPS if c["dtype"] == "f32" else PEAK_F64_MFMA_TFLOPS
        c["frac_of_mfma_peak"] = c["tflops"] / (peak * world)
        c["n_gpus"] = world

    # The same W + K steps once more, now that the GPU has been busy for a while: the timed region above starts
    # ~10 ms after the process's first kernel, while the chip is still raising its clocks (a fixed rocBLAS GEMM runs
    # 4.30 -> 3.69 ms over the first ~60 ms of work: profiles/r03_clock_ramp.txt).  `value` stays the first
    # run's, as the contract asks; this one is reported beside it as `warm_repeat`.
    eng.run([(2, i) for i in range(args.warmup)], state, train=True)
    barrier()
    t0 = time.perf_counter()
    eng.run([(3, i) for i in range(args.steps)], state, train=True)
    barrier()
    warm = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([warm], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        warm = float(tt[0].item())

    if rank == 0:
        flop = float(N) * N * B  # algorithmic: n^2 * B  (draw TRMM)
        f32_equiv = flop / (draw_ms * 1e-3) / 1e12 if draw_ms else None
        if eng.b3:
            # bf16 x 3: six bf16 piece products per float product are ISSUED — the roofline counts those
            # against the dense bf16 peak; the f32-equivalent rate (n^2 B / t) is quoted beside it
            roof = {"kernel": "draw_b3_kernel (Y = E L^T + sigma E', bf16 x 3: f32-accurate, f32 accumulate)",
                    "mfma": "bf16x3", "bound": "mfma", "achieved": 6 * f32_equiv if f32_equiv else None,
                    "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6 * f32_equiv / PEAK_BF16_MFMA_TFLOPS if f32_equiv else None, "traffic": None,
                    "avg_launch_ms": draw_ms, "flop_per_launch": 6 * flop, "algorithmic_flop_per_launch": flop,
                    "f32_equivalent_tflops": f32_equiv,
                    "f32_equivalent_frac_of_f32_mfma_peak": f32_equiv / PEAK_F32_MFMA_TFLOPS if f32_equiv else None}
            roof["mfma_only_sustained_tflops_random_operands"] = SUSTAINED_BF16_MFMA_RANDOM_TFLOPS
            roof["frac_of_mfma_only_sustained"] = 6 * f32_equiv / SUSTAINED_BF16_MFMA_RANDOM_TFLOPS if f32_equiv else None
            roof["launches_timed"] = "%d of %d (every launch number divisible by %d in the timed region, HIP events on its stream)" % (
                draw_timed, args.steps, DRAW_EVENTS_EVERY)
            roof["note"] = ("avg_launch_ms / frac: HIP events around the draw's launches INSIDE the timed steps, where the "
                            "factor stream's and the normals stream's workgroups hold part of the CUs (a draw workgroup "
                            "fills a CU's LDS (145 KB of 160): nothing that needs LDS runs beside it on that CU); "
                            "avg_launch_ms_alone / frac_alone: the same launch with nothing beside it on the chip")
            if draw_alone_ms:  # the kernel's own time (what a kernel trace reports), nothing beside it on the chip
                roof["avg_launch_ms_alone"] = draw_alone_ms
                roof["frac_alone"] = 6 * flop / (draw_alone_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS
        elif getattr(eng, "h2", False):
            # fp16 x 2 (opt-in): three fp16 piece products per float product are issued
            roof = {"kernel": "draw_b3_kernel<NP=2> (Y = E L^T + sigma E', fp16 x 2: operands to 22 bits, f32 accumulate)",
                    "mfma": "f16x2", "bound": "mfma", "achieved": 3 * f32_equiv if f32_equiv else None,
                    "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": 3 * f32_equiv / PEAK_BF16_MFMA_TFLOPS if f32_equiv else None, "traffic": None,
                    "avg_launch_ms": draw_ms, "flop_per_launch": 3 * flop, "algorithmic_flop_per_launch": flop,
                    "f32_equivalent_tflops": f32_equiv}
            if draw_alone_ms:
                roof["avg_launch_ms_alone"] = draw_alone_ms
                roof["frac_alone"] = 3 * flop / (draw_alone_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS
        else:
            roof = {"kernel": "draw_kernel<float> (Y = E L^T + sigma E')", "mfma": "f32", "bound": "mfma",
                    "achieved": f32_equiv, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": f32_equiv / PEAK_F32_MFMA_TFLOPS if f32_equiv else None, "traffic": None,
                    "avg_launch_ms": draw_ms, "flop_per_launch": flop}
        out = {
            "metric": "GP-prior VAE training draws/sec at 64x64 grid",
            "value": B * args.steps * world / elapsed,
            "unit": "draws/s",
            "n_gpus": world,
            "rccl_world": rccl_world,
            "ms_per_step_rank_min_max": rank_ms,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3: 2D 64x64 grid (n=4096), exponential kernel, jitter 1e-3, "
                                   "B=4096 GP draws/step/GPU, VAE 4096-50-30-15, Adam lr 1e-3, hyper-priors of vae.py:157-159; "
                                   "full step gram+potrf+draw+ELBO fwd/bwd+Adam",
                       "n": N, "draws_per_step_per_gpu": B, "hidden_dims": list(DIMS[:2]),
                       "z_dim": DIMS[2], "parallelism": "dp%d" % world,
                       "draw_mfma": "bf16x3" if eng.b3 else ("f16x2" if getattr(eng, "h2", False) else "f32"),
                       "hyper_priors_drawn_on": args.hyper,
                       "factor_stream_overlap": not args.no_overlap, "factor_group": args.group,
                       "forced_single_rank_collective": bool(args.force_collective)},
            "roofline": roof,
            "final_loss": loss_last,
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
            "warm_repeat": {"ms_per_step": warm / args.steps * 1e3, "value": B * args.steps * world / warm,
                            "note": "the same warmup + steps again after the Cholesky microbenchmarks (clocks up); "
                                    "not the headline"},
            "cholesky": chol[0],
            "cholesky_factor_group": chol[1],
            "cholesky_f64": chol[2],
        }
        if two_streams:
            out.update({"cholesky_two_streams": chol[3], "cholesky_batch64_two_streams": chol[4],
                        "cholesky_f64_two_streams": chol[5]})
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = (None, None) if getattr(eng, "h2", False) else pmc_traffic(eng.b3)
        # the side measurements must not cost the headline its line: a failure in one of them is reported, not raised
        if world == 1 and not args.no_extras:
            try:
                extras(ops, x, dev, out)
            except Exception as e:  # noqa: BLE001
                out["extras_error"] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
