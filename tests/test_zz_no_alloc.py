"""The one entry point that allocates nothing (gpvae_decoder_vjp_static): through the C ABI, against
gpvae_decoder_vjp (the existing entry point with a caller workspace) and the oracle, on two calls in
a row — the second with a smaller input, so stale contents of the shared static scratch would show —
and rejection of sizes beyond the 4 MiB scratch.  Collected last (test_zz_*): the round end runs
`pytest -x`."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import gpvae_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
def test_decoder_vjp_static_matches_workspace_entry_point(tdt, tol):
    from gp_vae_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(5)
    # first call: 1024 chains of the reference decoder (10-30-35-300); second: fewer chains, other dims
    for chains, n, h0, h1, z in ((1024, 300, 35, 30, 10), (37, 64, 12, 9, 4), (300, 625, 50, 30, 15)):
        p = {k: 8 * v for k, v in orc.init_params(rng, n, h0, h1, z, np.float64).items()}
        dec = torch.tensor(np.concatenate([p[k].reshape(-1) for k in ("V1", "c1", "V2", "c2", "V3", "c3")]), dtype=tdt, device=DEV)
        Z = torch.tensor(rng.standard_normal((chains, z)), dtype=tdt, device=DEV)
        dF = torch.tensor(rng.standard_normal((chains, n)), dtype=tdt, device=DEV)
        want = ops.decoder_vjp(dec, Z, dF, h0, h1)          # existing entry point, caller workspace
        got = torch.full_like(Z, float("nan"))
        dt = 0 if tdt == torch.float32 else 1
        rc = lib.gpvae_decoder_vjp_static(C.c_void_p(torch.cuda.current_stream().cuda_stream), dt, chains, n, h0, h1, z,
                                          C.c_void_p(dec.data_ptr()), C.c_void_p(Z.data_ptr()), C.c_void_p(dF.data_ptr()),
                                          C.c_void_p(got.data_ptr()))
        assert rc == 0, lib.gpvae_last_error()
        torch.cuda.synchronize()
        assert rel(got.cpu().numpy(), want.cpu().numpy()) < tol, (chains, n)
        assert rel(got.cpu().numpy(), orc.decoder_vjp(p, Z.cpu().numpy().astype(np.float64), dF.cpu().numpy().astype(np.float64))) < max(tol, 1e-11)
        # the wrapper with a caller-provided output allocates nothing either
        out = torch.empty_like(Z)
        assert ops.decoder_vjp_static(dec, Z, dF, h0, h1, out=out) is out and rel(out.cpu().numpy(), want.cpu().numpy()) < tol


def test_decoder_vjp_static_rejects_sizes_beyond_the_scratch():
    from gp_vae_amd import GpvaeError, _lib, ops
    lib = _lib.load()
    null = C.c_void_p(0)
    one = C.c_void_p(16)  # non-NULL stand-ins: the size check comes before any access
    # 1 << 20 chains * (30 + 70) * 4 bytes = 400 MiB > 4 MiB
    assert lib.gpvae_decoder_vjp_static(null, 0, 1 << 20, 300, 35, 30, 10, one, one, one, one) == -3
    assert b"exceeds the static scratch" in lib.gpvae_last_error()
    assert lib.gpvae_decoder_vjp_static(null, 0, 0, 300, 35, 30, 10, null, null, null, null) == 0  # empty
    Z = torch.zeros((1 << 16, 10), device=DEV)
    with pytest.raises(GpvaeError):
        ops.decoder_vjp_static(torch.zeros(30 * 10 + 30 + 30 * 35 + 35 + 35 * 300 + 300, device=DEV), Z,
                               torch.zeros((1 << 16, 300), device=DEV), 35, 30)
