"""Developer helper: byte offsets of the workspace sections (now in the package: exavatar_release_amd/stats.py)."""
from exavatar_release_amd.stats import BIN_PARTS, CHUNK, HEADER_BYTES, SUBS, a256, bin_offsets, tile_offsets  # noqa: F401
