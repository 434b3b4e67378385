"""gp_vae_amd — MI355X (gfx950) hot path of edenx/gp-vae behind the reference's Python surface.

The directory is also reachable as ``gp-vae_amd`` (symlink) — the hyphenated name is not importable.
"""
import os as _os

# The training loop keeps three streams busy (draw/ELBO, factor stream, RCCL's stream).  HIP maps streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and on this stack the collective's stream then
# shares a queue with the factor stream: its event packets wait behind a whole group of factorisation
# launches and the overlap of the two streams is lost (1.17 instead of 1.04 ms per C3 step, DESIGN.md
# §6).  The runtime reads the variable when it initialises, i.e. at the first HIP call — effective as
# long as this package is imported before that; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import GpvaeError, LIB_PATH, load  # noqa: F401,E402
