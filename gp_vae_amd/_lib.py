"""ctypes binding of libgpvae_hip.so (include/gpvae.h).

The library is the product: there is no CPU fallback.  Loading fails loudly if the shared object
is missing (run ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C
gp_vae_amd/csrc``), and every wrapper raises :class:`GpvaeError` on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpvae_hip.so")

F32, F64 = 0, 1
ABI_VERSION = 2  # include/gpvae.h GPVAE_ABI_VERSION
POTRF_ZERO_UPPER, POTRF_NAN_ON_FAIL, POTRF_SHARE_CHIP = 1, 2, 4


class GpvaeError(RuntimeError):
    pass


_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double

# name -> (restype, argtypes); must list every symbol include/gpvae.h declares
SIGNATURES = {
    "gpvae_abi_version": (_i, []),
    "gpvae_last_error": (C.c_char_p, []),
    "gpvae_gram_se": (_i, [_vp, _i, _i, _i64, _i64, _vp, _vp, _i, _vp, _vp, _d, _vp, _i64, _i64]),
    "gpvae_gram_agg": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _vp, _d, _vp, _i64]),
    "gpvae_potrf_workspace_bytes": (_i64, [_i, _i64, _i64, _i]),
    "gpvae_potrf": (_i, [_vp, _i, _i64, _i, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i, _vp]),
    "gpvae_trsm": (_i, [_vp, _i, _i, _i64, _i64, _vp, _i64, _vp, _i64]),
    "gpvae_draw": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64,
                        _vp, _i64]),
    "gpvae_draw_batched": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                _i64, _vp, _i64]),
    "gpvae_randn": (_i, [_vp, _i, _i64, _u64, _u64, _vp]),
    "gpvae_potrf_b3_supported": (_i, [_i64]),
    "gpvae_potrf_b3_workspace_bytes": (_i64, [_i64, _i]),
    "gpvae_potrf_b3": (_i, [_vp, _i64, _i, _vp, _i64, _i64, _vp, _i, _vp, _vp, _i64, _i64, _i64]),
    "gpvae_potrf_h2_workspace_bytes": (_i64, [_i64, _i]),
    "gpvae_potrf_h2": (_i, [_vp, _i64, _i, _vp, _i64, _i64, _vp, _i, _vp, _vp, _i64, _i64, _i64, _vp]),
    "gpvae_split_b3": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i64, _i, _vp, _i64, _i64, _i64]),
    "gpvae_randn_b3": (_i, [_vp, _i64, _u64, _u64, _vp, _i64]),
    "gpvae_draw_b3_supported": (_i, [_i64, _i64]),
    "gpvae_draw_b3": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64,
                           _vp, _i64]),
    "gpvae_hyper_sample": (_i, [_vp, _i, _i64, _u64, _u64, _i64, _i, _d, _d, _d, _d, _vp, _vp, _vp]),
    "gpvae_poisson_exp": (_i, [_vp, _i, _i64, _u64, _u64, _i64, _vp, _vp, _vp]),
    "gpvae_krig_workspace_bytes": (_i64, [_i, _i64, _i64]),
    "gpvae_krig_predict": (_i, [_vp, _i, _i, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp,
                                _vp, _vp, _vp, _vp, _vp]),
    "gpvae_krig_batched_workspace_bytes": (_i64, [_i, _i64, _i64, _i]),
    "gpvae_krig_predict_batched": (_i, [_vp, _i, _i, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gpvae_krig_set_path": (None, [_i]),
    "gpvae_mvn_workspace_bytes": (_i64, [_i, _i64, _i]),
    "gpvae_mvn_logp": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _vp,
                            _vp]),
    "gpvae_gp_potential_workspace_bytes": (_i64, [_i, _i64, _i]),
    "gpvae_gp_potential": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _i64, _vp, _vp, _i64, _d, _vp, _vp, _i64, _vp, _vp]),
    "gpvae_vae_param_count": (_i64, [_i64, _i, _i, _i]),
    "gpvae_vae_workspace_bytes": (_i64, [_i, _i64, _i64, _i, _i, _i]),
    "gpvae_vae_elbo": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _d, _vp, _vp,
                            _vp]),
    "gpvae_encoder_workspace_bytes": (_i64, [_i, _i64, _i64, _i, _i, _i]),
    "gpvae_encoder_fwd": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gpvae_adam": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d]),
    "gpvae_decoder_workspace_bytes": (_i64, [_i, _i64, _i64, _i, _i, _i]),
    "gpvae_decoder_fwd": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gpvae_decoder_vjp": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpvae_vae_potential_supported": (_i, [_i, _i, _i, _i, _i64]),
    "gpvae_vae_potential": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i64, _d, _vp, _vp, _i64]),
    "gpvae_split_h2": (_i, [_vp, _i64, _i64, _i, _vp, _i64, _i64, _i, _vp, _vp, _i64, _i64, _i64, _vp]),
    "gpvae_randn_h2": (_i, [_vp, _i64, _u64, _u64, _vp, _i64]),
    "gpvae_draw_h2": (_i, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64]),
    "gpvae_nuts_op": (_i, [_vp, _i, _i, _i64, _i64, _i, _i, _i, _i, _i, _vp]),
    "gpvae_nuts_vae_supported": (_i, [_i, _i, _i, _i, _i64]),
    "gpvae_nuts_vae_iteration": (_i, [_vp, _i, _i64, _i, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i64, _d]),
    "gpvae_nuts_vae_run": (_i, [_vp, _i, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i64, _d]),
    "gpvae_decoder_vjp_static": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises GpvaeError if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpvaeError(
            "libgpvae_hip.so not found at %s: the HIP library is required (no CPU fallback). "
            "Build it with `make -C gp_vae_amd/csrc` or __graft_entry__.build()." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gpvae_abi_version() != ABI_VERSION:
        raise GpvaeError("libgpvae_hip.so reports ABI version %d, this binding was written for %d (include/gpvae.h: "
                         "GPVAE_ABI_VERSION) — rebuild with `make -C gp_vae_amd/csrc`" % (lib.gpvae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gpvae_last_error()
        raise GpvaeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
