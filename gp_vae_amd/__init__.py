"""gp_vae_amd — MI355X (gfx950) hot path of edenx/gp-vae behind the reference's Python surface.

The directory is also reachable as ``gp-vae_amd`` (symlink) — the hyphenated name is not importable.
"""
from ._lib import GpvaeError, LIB_PATH, load  # noqa: F401
