"""Re-serialise the reference's six decoder weight files (src/test/decoders/*) as one
plain .npz of float32 arrays: tests/golden/decoders.npz.

Runs in the BUILD container only (it reads /root/reference, which never travels to the GPU box).
The files are pickles of `[(W0,b0), (), (W1,b1), (), (W2,b2)]` numpy arrays; they are read with
`torch.load(..., weights_only=True)` — torch's restricted unpickler, which executes nothing the file
brings along and refuses every global that is not allow-listed — with exactly the three numpy
constructors an ndarray pickle names allow-listed (`numpy.core.multiarray._reconstruct`,
`numpy.ndarray`, `numpy.dtype`).  No `pickle.load`, no `allow_pickle=True`.  A file the loader
refuses is skipped and reported.

RESULT in this image (torch 2.10, numpy 2.2): every file is REFUSED — `Unsupported operand 67`: the
files are pickle protocol 3 and torch's weights-only unpickler does not implement its
SHORT_BINBYTES / BINBYTES opcodes; `numpy.load` (allow_pickle=False) refuses them as "pickled
(object) data".  No other loader is permitted for files shipped inside the reference, so the build
proceeds without them: nothing is written.

The .npz holds DATA only (weights = inputs of decoder known-answer tests; a z-grid; the oracle's
outputs on it).  The reference ships no expected outputs for these weights, so the outputs are
oracle-derived: parity stays unpinned.

usage: python tests/golden/make_decoder_fixtures.py
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SRC = "/root/reference/src/test/decoders"
NAMES = ["decoder_parmas_1d_n300_GP", "decoder_parmas_1d_n300_GP_fixls", "decoder_parmas_1d_n300_GP_fixls_500",
         "decoder_parmas_1d_n300_GP_fixls_ep70", "decoder_parmas_2d_n25_GP_fixls", "decoder_parmas_2d_n25_GP_fixns"]


def allow_list():
    # the pickles name numpy 1.x's module path; numpy 2 moved the function to numpy._core, so the
    # allow-list entry carries the name the file uses (torch accepts (callable, "qualified.name") pairs)
    from numpy._core.multiarray import _reconstruct
    return [(_reconstruct, "numpy.core.multiarray._reconstruct"), np.ndarray, np.dtype,
            type(np.dtype(np.float32))]


def load_weights_only(path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.serialization.safe_globals(allow_list()):
            return torch.load(path, weights_only=True)


def main():
    from oracle import gpvae_oracle as orc
    out = {}
    rng = np.random.default_rng(20261111)
    for name in NAMES:
        try:
            obj = load_weights_only(os.path.join(SRC, name))
        except Exception as e:  # the safe loader refused: say so, go on without the file
            print("SKIPPED %s: %s: %s" % (name, type(e).__name__, str(e).splitlines()[0][:200]))
            continue
        (V1, c1), e1, (V2, c2), e2, (V3, c3) = obj
        assert e1 == () and e2 == ()
        arrs = [np.asarray(a) for a in (V1, c1, V2, c2, V3, c3)]
        assert all(a.dtype == np.float32 for a in arrs)
        key = name.replace("decoder_parmas_", "")
        for tag, a in zip(("V1", "c1", "V2", "c2", "V3", "c3"), arrs):
            out["%s__%s" % (key, tag)] = a
        z_dim = arrs[0].shape[0]
        Z = rng.standard_normal((16, z_dim)).astype(np.float32)
        Z[0] = 0.0
        out["%s__Z" % key] = Z
        p64 = {t: a.astype(np.float64) for t, a in zip(("V1", "c1", "V2", "c2", "V3", "c3"), arrs)}
        with np.errstate(all="ignore"):
            out["%s__F_oracle_f64" % key] = orc.decoder(p64, Z.astype(np.float64))
        print("%s: shapes %s finite=%s" % (key, [a.shape for a in arrs], bool(all(np.isfinite(a).all() for a in arrs))))
    path = os.path.join(ROOT, "tests", "golden", "decoders.npz")
    if not out:
        print("no file was readable with the safe loaders: decoders.npz not written; the decoder tests use "
              "synthetic weights of the same shapes and spread (tests/test_gpu_parity.py)")
        return
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
