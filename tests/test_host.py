"""CPU tests of the host logic above the C ABI (parameter layouts, seeds, sharding) and of the
data-parallel step on world_size 2 (gloo): gradients of two ranks' own batches, all-reduced with
the product's one collective, equal the gradient of the concatenated batch, and both ranks end
with identical parameters.  The compute stand-in in these tests is the oracle (tests only)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gp_vae_amd import gp as gpmod
from gp_vae_amd import parallel, vae as vaemod
from oracle import gpvae_oracle as orc


def test_decoder_list_roundtrip_matches_reference_layout():
    n, h0, h1, z = 300, 35, 30, 10  # the shapes of the reference's pickled decoders
    P = n * h0 + h0 + h0 * h1 + h1 + 2 * (h1 * z + z) + z * h1 + h1 + h1 * h0 + h0 + h0 * n + n
    flat = torch.arange(P, dtype=torch.float64)
    lst = vaemod.decoder_list_from_flat(flat, n, h0, h1, z)
    assert [tuple(a.shape) for a in (lst[0][0], lst[2][0], lst[4][0])] == [(10, 30), (30, 35), (35, 300)]
    assert lst[1] == () and lst[3] == () and len(lst) == 5
    p = orc.unflatten(flat.numpy(), n, h0, h1, z)
    assert np.array_equal(lst[0][0].numpy(), p["V1"]) and np.array_equal(lst[4][1].numpy(), p["c3"])
    assert np.array_equal(lst[2][0].numpy(), p["V2"]) and np.array_equal(lst[2][1].numpy(), p["c2"])


def test_seed_and_stream_ids():
    assert gpmod._seed_of(5) == 5 and gpmod._seed_of(None) == 0
    assert gpmod._seed_of(np.array([1, 2], dtype=np.uint32)) != gpmod._seed_of(np.array([2, 1], dtype=np.uint32))
    ids = {parallel.philox_stream_id(r, (e, s), p) for r in range(8) for e in (1, 2) for s in range(50) for p in (1, 2, 3)}
    assert len(ids) == 8 * 2 * 50 * 3
    a = parallel.step_rng(3, 0, (7, 1)).random()
    assert a == parallel.step_rng(3, 0, (7, 1)).random() and a != parallel.step_rng(3, 1, (7, 1)).random()


def test_hyper_sampler_matches_oracle_priors():
    rng = np.random.default_rng(0)
    ls, var, sig = gpmod.draw_hyper(rng, 40000)
    assert abs(np.median(ls) - 0.1 / np.log(2)) < 0.01
    assert abs(np.std(np.log(var)) - 0.1) < 0.005 and abs(np.mean(sig) - 0.1 * np.sqrt(2 / np.pi)) < 0.002


def test_shard_covers_everything_once():
    got = sorted(i for r in range(8) for i in parallel.shard(1024, r, 8))
    assert got == list(range(1024)) and len(parallel.shard(1024, 3, 8)) == 128


def _dp_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, h0, h1, z, B = 24, 5, 4, 3, 6
        x = np.arange(0, 1, 1 / n)
        rng0 = np.random.default_rng(11)
        p = orc.init_params(rng0, n, h0, h1, z, np.float64)
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(a) for k, a in p.items()}
        batches = []
        for t in range(1, 4):
            key = (5, t)
            assert parallel.dist_info() == (rank, world)
            hyper = gpmod.draw_hyper(parallel.step_rng(0, rank, key))  # ranks draw different GPs
            r = np.random.default_rng([rank, t])
            E, E2, ez = r.standard_normal((B, n)), r.standard_normal((B, n)), r.standard_normal((B, z))
            K = orc.exp_kernel2(x, x, hyper[1], hyper[0], 1e-4)
            _, Y = orc.gp_draw(orc.cholesky_lower(K), E, hyper[2], E2)
            loss, g = orc.elbo_loss_and_grads(p, Y, ez)
            flat = torch.from_numpy(orc.flatten(g).copy())
            parallel.allreduce_sum_(flat)  # the product's one collective
            g = orc.unflatten(flat.numpy(), n, h0, h1, z)
            for k in orc.PARAM_NAMES:
                p[k], m[k], v[k] = orc.adam_step(p[k], g[k], m[k], v[k], t, 1e-3)
            batches.append((Y, ez, flat.numpy().copy()))
        np.savez(os.path.join(tmp, "rank%d.npz" % rank), params=orc.flatten(p),
                 **{"Y%d" % i: b[0] for i, b in enumerate(batches)},
                 **{"e%d" % i: b[1] for i, b in enumerate(batches)},
                 **{"g%d" % i: b[2] for i, b in enumerate(batches)})
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_world2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["params"], r1["params"])  # replicas stay identical
    assert np.array_equal(r0["g0"], r1["g0"])
    # summed gradient == gradient of the concatenated batch (the loss is a SUM, vae.py:165-166)
    n, h0, h1, z = 24, 5, 4, 3
    p = orc.init_params(np.random.default_rng(11), n, h0, h1, z, np.float64)
    Y = np.concatenate([r0["Y0"], r1["Y0"]])
    ez = np.concatenate([r0["e0"], r1["e0"]])
    _, g = orc.elbo_loss_and_grads(p, Y, ez)
    assert np.allclose(orc.flatten(g), r0["g0"], rtol=1e-10, atol=1e-10)
    assert not np.array_equal(r0["Y0"], r1["Y0"])  # the ranks really drew different batches


def test_hmc_recovers_a_gaussian_target():
    """The batched HMC driver (stand-in for the absent NUTS) on N(mu, diag(s^2)): adapted step sizes,
    acceptance near the target, sample moments within Monte-Carlo error."""
    from gp_vae_amd.hmc import hmc
    mu = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    sd = torch.tensor([0.5, 2.0, 1.0], dtype=torch.float64)

    def pot(q):
        r = (q - mu) / sd
        return 0.5 * (r * r).sum(1), r / sd

    smp, info = hmc(pot, torch.zeros((8, 3), dtype=torch.float64), 300, 600, num_leapfrog=12, seed=4)
    assert smp.shape == (600, 8, 3) and info["divergent"] == 0
    assert 0.6 < info["accept_prob"].mean().item() < 0.98
    flat = smp.reshape(-1, 3)
    assert torch.allclose(flat.mean(0), mu, atol=0.15)
    assert torch.allclose(flat.std(0), sd, rtol=0.15)
    # thinning and determinism
    a, _ = hmc(pot, torch.zeros((2, 3), dtype=torch.float64), 20, 40, thinning=4, seed=9)
    b, _ = hmc(pot, torch.zeros((2, 3), dtype=torch.float64), 20, 40, thinning=4, seed=9)
    assert a.shape == (10, 2, 3) and torch.equal(a, b)


def test_nuts_recovers_an_anisotropic_gaussian():
    """The batched NUTS (restated algorithm behind numpyro.infer.NUTS, inference.py:84-95) on
    N(mu, diag(s^2)) with scales over two decades: moments within Monte-Carlo error, the diagonal mass
    matrix finds the variances, no divergence, chains independent of their batch neighbours."""
    from gp_vae_amd.nuts import nuts, _ckpt_range, uniforms_per_iteration
    assert [_ckpt_range(k) for k in (0, 1, 6, 7, 13)] == [(1, 0), (0, 0), (3, 2), (0, 2), (2, 2)]
    assert uniforms_per_iteration(3) == 14
    mu = torch.tensor([1.0, -2.0, 0.0, 5.0], dtype=torch.float64)
    sd = torch.tensor([0.1, 1.0, 3.0, 10.0], dtype=torch.float64)

    def pot(q):
        r = (q - mu) / sd
        return 0.5 * (r * r).sum(1), r / sd

    smp, info = nuts(pot, torch.zeros((16, 4), dtype=torch.float64), 200, 200, max_tree_depth=7, seed=3)
    assert smp.shape == (200, 16, 4) and info["divergent"] == 0
    assert 0.6 < info["accept_prob"].mean().item() < 0.99
    flat = smp.reshape(-1, 4)
    assert torch.all((flat.mean(0) - mu).abs() < 0.1 * sd)
    assert torch.allclose(flat.std(0), sd, rtol=0.1)
    assert torch.allclose(info["inverse_mass"].mean(0), sd ** 2, rtol=0.5)
    assert 1.0 <= info["leapfrogs_per_iteration"].mean().item() < 2 ** 7

    # a chain's numbers do not depend on which chains share its batch (masked lock-step trees)
    def noise(rows):
        def fn(it):
            g = torch.Generator().manual_seed(1000 + it)
            z = torch.randn((6, 4), dtype=torch.float64, generator=g)
            u = torch.rand((6, uniforms_per_iteration(5)), dtype=torch.float64, generator=g)
            return z[rows], u[rows]
        return fn
    q0 = torch.linspace(-1, 1, 24, dtype=torch.float64).reshape(6, 4)
    full, _ = nuts(pot, q0, 30, 10, max_tree_depth=5, noise_fn=noise(torch.arange(6)))
    part, _ = nuts(pot, q0[[1, 4]], 30, 10, max_tree_depth=5, noise_fn=noise(torch.tensor([1, 4])))
    assert torch.equal(full[:, [1, 4]], part)


def test_decoder_param_files_round_trip(tmp_path):
    """Decoder-parameter I/O (the reference pickles VAE.fit()'s list, toy_examples.py:161-169): the npz form
    round-trips bit for bit, the pickle form is the reference's layout with plain numpy arrays, and the
    loader refuses anything that is not its own npz."""
    import pickle
    from gp_vae_amd._lib import GpvaeError
    from gp_vae_amd.vae import load_decoder_params, save_decoder_params
    rng = np.random.default_rng(0)
    shapes = [(10, 30), (30,), (30, 35), (35,), (35, 300), (300,)]
    a = [rng.standard_normal(sh).astype(np.float32) for sh in shapes]
    lst = [(torch.tensor(a[0]), torch.tensor(a[1])), (), (a[2], a[3]), (), (a[4], a[5])]
    f = str(tmp_path / "dec.npz")
    save_decoder_params(f, lst)
    back = load_decoder_params(f, device="cpu")
    assert [len(t) for t in back] == [2, 0, 2, 0, 2]
    flat = [back[0][0], back[0][1], back[2][0], back[2][1], back[4][0], back[4][1]]
    assert all(np.array_equal(t.numpy(), x) for t, x in zip(flat, a))
    g = str(tmp_path / "dec.pkl")
    save_decoder_params(g, lst, format="pickle")
    with open(g, "rb") as fh:  # a file this test wrote
        ref = pickle.load(fh)
    assert ref[1] == () and ref[3] == () and isinstance(ref[0][0], np.ndarray) and np.array_equal(ref[4][0], a[4])
    with pytest.raises(GpvaeError):
        load_decoder_params(g, device="cpu")
    np.savez(str(tmp_path / "bad.npz"), V1=a[0], c1=a[1], V2=a[4], c2=a[3], V3=a[4], c3=a[5])
    with pytest.raises(GpvaeError):
        load_decoder_params(str(tmp_path / "bad.npz"), device="cpu")


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher must start 2 ranks itself (the flag was
    parsed and ignored).  --dry-run: gloo, stand-in gradient, no GPU work — the launcher, rendezvous and
    the path's one collective are what is exercised."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world"] == 2 and out["dry_run"] and out["allreduce_sum_checked"] == 3.0
    one = _run_bench(["--gpus", "1", "--dry-run"])
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """Under a launcher (WORLD_SIZE set) a different --gpus is an error, never a mislabelled number."""
    r = _run_bench(["--gpus", "8", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_chains_shard_without_overlap():
    from gp_vae_amd.parallel import shard
    for world in (1, 2, 8):
        got = sorted(c for r in range(world) for c in shard(1024, r, world))
        assert got == list(range(1024))
