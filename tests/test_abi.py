"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/gpvae.h
declares, and validates arguments without touching a GPU (no compute call is made here)."""
import ctypes as C
import os
import re

import pytest

from gp_vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gpvae.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpvae_[a-z0-9_]+)\s*\(", src)))


def exported_symbols():
    """Dynamic symbols `gpvae_*` of the built library (nm -D; defined text symbols only)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    return sorted(set(re.findall(r"\bT (gpvae_[a-z0-9_]+)$", out, flags=re.M)))


def test_exported_symbols_equal_declared_symbols():
    """No undeclared export (developer probes live behind `make DEV=1`) and nothing declared is missing."""
    assert exported_symbols() == header_symbols()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) == 49
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, "ctypes signature missing for " + name
    assert sorted(_lib.SIGNATURES) == names
    assert lib.gpvae_abi_version() == 2


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "gpvae.h")).read()
    assert "torch" not in src.lower().replace("no torch", "") and "at::" not in src and 'extern "C"' in src


def test_argument_validation_without_a_gpu():
    lib = _lib.load()
    null = C.c_void_p(0)
    # bad dtype -> -(index of dtype argument) = -2, and a readable reason
    assert lib.gpvae_gram_se(null, 7, 1, 4, 4, null, null, 1, null, null, 0.0, null, 4, 16) == -2
    assert b"dtype" in lib.gpvae_last_error()
    assert lib.gpvae_gram_se(null, 0, 3, 4, 4, null, null, 1, null, null, 0.0, null, 4, 16) == -3
    assert lib.gpvae_gram_se(null, 0, 1, 4, 4, null, null, 1, null, null, 0.0, null, 4, 16) == -6  # x NULL
    assert lib.gpvae_potrf(null, 0, -1, 1, null, 0, 0, 0, null, 0, 0, null, 0, null) == -3
    assert lib.gpvae_potrf(null, 0, 8, 1, null, 8, 64, 0, null, 0, 0, null, 0, null) == -5  # A NULL
    assert lib.gpvae_draw(null, 0, 8, 4, null, 8, null, 8, null, 0, null, null, 0, null, 0) == -5
    assert lib.gpvae_adam(null, 0, 10, null, null, null, null, 0, 1e-3, 0.9, 0.999, 1e-8) == -8  # t >= 1
    assert lib.gpvae_vae_elbo(null, 0, 4, 8, 0, 2, 2, null, null, 8, null, 0.1, null, null, null) == -5
    assert lib.gpvae_krig_predict(null, 0, 3, 4, 4, null, null, null, null, null, null, 0.0, null,
                                  null, null, null, null, null) == -3
    # empty problems are accepted and do nothing
    assert lib.gpvae_gram_se(null, 0, 1, 0, 4, null, null, 1, null, null, 0.0, null, 4, 0) == 0
    assert lib.gpvae_potrf(null, 0, 0, 1, null, 0, 0, 0, null, 0, 0, null, 0, null) == 0
    assert lib.gpvae_draw(null, 0, 0, 4, null, 0, null, 0, null, 0, null, null, 0, null, 0) == 0
    assert lib.gpvae_trsm(null, 0, 2, 8, 4, null, 8, null, 8) == -3   # trans not 0/1
    assert lib.gpvae_trsm(null, 0, 1, 8, 4, null, 8, null, 8) == -6   # L is NULL
    assert lib.gpvae_trsm(null, 0, 1, 0, 4, null, 0, null, 0) == 0    # empty problem
    assert lib.gpvae_draw_batched(null, 0, 8, 4, null, 8, 64, null, 8, null, 0, null, null, 0, null, 0) == -5  # L NULL
    assert lib.gpvae_draw_batched(null, 0, 1 << 20, 4, null, 8, 64, null, 8, null, 0, null, null, 0, null, 0) == -3
    assert lib.gpvae_draw_batched(null, 0, 0, 4, null, 0, 0, null, 0, null, 0, null, null, 0, null, 0) == 0
    assert lib.gpvae_encoder_fwd(null, 0, 4, 8, 2, 2, 2, null, null, 8, null, null, null, null, null) == -8
    assert lib.gpvae_encoder_fwd(null, 0, 0, 8, 2, 2, 2, null, null, 8, null, null, null, null, null) == 0
    # the no-allocation VJP rejects what its 4 MiB static scratch cannot hold, before touching a GPU
    one = C.c_void_p(16)
    assert lib.gpvae_decoder_vjp_static(null, 0, 1 << 20, 300, 35, 30, 10, one, one, one, one) == -3
    assert b"static scratch" in lib.gpvae_last_error()
    assert lib.gpvae_decoder_vjp_static(null, 0, 0, 300, 35, 30, 10, null, null, null, null) == 0


def test_argument_validation_of_the_round_2_entry_points():
    """Bad arguments come back as negative codes with a reason, before any launch; empty problems are no-ops."""
    lib = _lib.load()
    null, one = C.c_void_p(0), C.c_void_p(16)
    assert lib.gpvae_gp_potential(null, 5, 1, 10, 4, one, one, 13, one, None, 10, 1e-5, one, one, 13, None, one) == -2
    assert lib.gpvae_gp_potential(null, 0, 3, 10, 4, one, one, 13, one, None, 10, 1e-5, one, one, 13, None, one) == -3
    assert lib.gpvae_gp_potential(null, 0, 1, 10, 4, one, one, 12, one, None, 10, 1e-5, one, one, 13, None, one) == -8
    assert b"ldq" in lib.gpvae_last_error()
    assert lib.gpvae_gp_potential(null, 0, 1, 10, 4, one, one, 13, one, None, 5, 1e-5, one, one, 13, None, one) == -11  # n_lik != n without lik_idx
    assert lib.gpvae_gp_potential(null, 0, 1, 10, 0, null, null, 13, null, None, 10, 1e-5, null, null, 13, None, null) == 0
    assert lib.gpvae_mvn_logp(null, 9, 1, 10, 4, one, one, one, one, 1e-5, one, null, null, null, None, one) == -2
    assert lib.gpvae_mvn_logp(null, 0, 1, 10, 70000, one, one, one, one, 1e-5, one, null, null, null, None, one) == -5
    assert lib.gpvae_mvn_logp(null, 0, 1, 10, 0, null, null, null, null, 1e-5, null, null, null, null, None, null) == 0
    # bf16 x 3 entry points: shapes the tiling does not take are refused, not mis-computed
    assert lib.gpvae_draw_b3_supported(4096, 4096) == 1 and lib.gpvae_draw_b3_supported(300, 100) == 0
    assert lib.gpvae_draw_b3_supported(4096, 100) == 0 and lib.gpvae_potrf_b3_supported(300) == 0
    assert lib.gpvae_potrf_b3_supported(4096) == 1 and lib.gpvae_potrf_b3_supported(256) == 0
    assert lib.gpvae_draw_b3(null, 300, 256, one, 304, 304 * 300, one, 304, 256 * 304, null, 0, null, null, 0, one, 304) < 0
    assert b"gpvae_draw_b3" in lib.gpvae_last_error()
    assert lib.gpvae_potrf_b3(null, 300, 1, one, 300, 90000, None, 0, one, one, 304, 304 * 300, 3 * 304 * 300) < 0
    assert lib.gpvae_randn_b3(null, 6, 1, 2, one, 8) < 0          # count % 4 != 0
    assert lib.gpvae_randn_b3(null, 0, 1, 2, null, 0) == 0
    assert lib.gpvae_split_b3(null, 8, 8, 1, null, 8, 64, 0, one, 8, 64, 192) < 0  # X NULL
    # fp16 x 2 entry points: the same shape rules, and every pointer is checked before a launch
    assert lib.gpvae_potrf_h2(null, 300, 1, one, 300, 90000, one, 0, one, one, 304, 304 * 300, 2 * 304 * 300, one) == -2
    assert b"gpvae_potrf_h2" in lib.gpvae_last_error()
    assert lib.gpvae_potrf_h2(null, 512, 1, one, 512, 512 * 512, one, 0, one, one, 512, 512 * 512, 2 * 512 * 512, None) == -14  # inv_scale
    assert lib.gpvae_potrf_h2(null, 512, 2, one, 512, 512 * 512, one, 0, one, one, 512, 512 * 512, 512 * 512, one) == -13    # stride2 < 2 planes
    assert lib.gpvae_potrf_h2(null, 0, 1, null, 0, 0, null, 0, null, null, 0, 0, 0, None) == 0
    assert lib.gpvae_potrf_h2_workspace_bytes(4096, 16) == 16 * 4096 * 64 * 4 + 256
    assert lib.gpvae_draw_h2(null, 300, 256, one, 304, 304 * 300, one, one, 304, 256 * 304, null, 0, null, null, 0, one, 304) < 0
    assert b"gpvae_draw_h2" in lib.gpvae_last_error()
    assert lib.gpvae_randn_h2(null, 6, 1, 2, one, 8) < 0          # count % 4 != 0
    assert lib.gpvae_split_h2(null, 8, 8, 1, null, 8, 64, 0, one, one, 8, 64, 128, one) < 0  # X NULL
    assert lib.gpvae_poisson_exp(null, 7, 4, 1, 2, 0, one, one, one) == -2                          # dtype
    assert lib.gpvae_poisson_exp(null, 0, 4, 1, 2, 2 ** 32, one, one, one) == -6                    # index range
    assert lib.gpvae_poisson_exp(null, 0, 4, 1, 2, 0, one, null, null) == -9                        # nothing to write
    assert lib.gpvae_poisson_exp(null, 0, 0, 1, 2, 0, null, null, null) == 0
    assert lib.gpvae_hyper_sample(null, 3, 4, 1, 2, 0, 0, 1.0, 0.1, 0.1, 0.1, one, one, one) < 0   # dtype
    assert lib.gpvae_hyper_sample(null, 0, 4, 1, 2, 0, 7, 1.0, 0.1, 0.1, 0.1, one, one, one) < 0   # prior id
    assert lib.gpvae_hyper_sample(null, 0, 0, 1, 2, 0, 0, 1.0, 0.1, 0.1, 0.1, null, null, null) == 0
    assert lib.gpvae_nuts_op(null, 0, 9, 4, 3, 5, 0, 0, 0, 0, null) == -3       # unknown op
    assert lib.gpvae_nuts_op(null, 0, 3, 4, 3, 5, 2, 4, 0, 0, null) == -8       # leaf 4 does not exist at depth 2
    assert lib.gpvae_nuts_op(null, 0, 0, 4, 3, 5, 0, 0, 0, 0, null) == -11 and b"NULL" in lib.gpvae_last_error()
    assert lib.gpvae_nuts_op(null, 0, 0, 0, 3, 5, 0, 0, 0, 0, null) == 0        # no chains: nothing to do
    assert lib.gpvae_vae_potential_supported(0, 35, 30, 10, 10) == 1 and lib.gpvae_vae_potential_supported(0, 1024, 1024, 64, 4096) == 0
    assert lib.gpvae_vae_potential(null, 0, 4, 300, 1024, 1024, 64, one, one, 65, one, None, 4096, 0.1, one, one, 65) < 0


def test_size_queries():
    lib = _lib.load()
    n, h0, h1, z = 300, 35, 30, 10
    want = n * h0 + h0 + h0 * h1 + h1 + 2 * (h1 * z + z) + z * h1 + h1 + h1 * h0 + h0 + h0 * n + n
    assert lib.gpvae_vae_param_count(n, h0, h1, z) == want
    assert lib.gpvae_vae_workspace_bytes(0, 100, n, h0, h1, z) >= 100 * n * 4
    assert lib.gpvae_vae_workspace_bytes(1, 100, n, h0, h1, z) >= 100 * n * 8
    assert lib.gpvae_krig_workspace_bytes(1, 100, 50) >= (100 * 100 + 51 * 100) * 8
    # f32, n % 256 == 0, no extra rows: parked diagonal blocks + the current group of outer panels (4 at
    # n >= 4096, 2 at n >= 1024, else 1) as three bf16 planes
    assert lib.gpvae_potrf_workspace_bytes(0, 4096, 0, 32) == 32 * 4096 * 64 * 4 + 32 * 3 * 4096 * 1024 * 2
    assert lib.gpvae_potrf_workspace_bytes(0, 2048, 0, 2) == 2 * 2048 * 64 * 4 + 2 * 3 * 2048 * 512 * 2
    assert lib.gpvae_potrf_workspace_bytes(0, 1024, 0, 2) == 2 * 1024 * 64 * 4 + 2 * 3 * 1024 * 512 * 2
    assert lib.gpvae_potrf_workspace_bytes(0, 768, 0, 2) == 2 * 768 * 64 * 4 + 2 * 3 * 768 * 256 * 2
    assert lib.gpvae_potrf_workspace_bytes(0, 4096, 5, 32) == 32 * 4096 * 64 * 4
    assert lib.gpvae_potrf_workspace_bytes(1, 4096, 0, 32) == 32 * 4096 * 64 * 8
    assert lib.gpvae_potrf_workspace_bytes(1, 100, 7, 2) == 2 * 128 * 64 * 8


def test_cpu_tensors_are_refused_loudly():
    import torch
    from gp_vae_amd import GpvaeError, ops
    x = torch.zeros(4)
    with pytest.raises(GpvaeError):
        ops.gram_se(x, x, 1.0, 1.0)
    with pytest.raises(GpvaeError):
        ops.potrf_(torch.eye(4))
