"""Developer helper: byte offsets of the tile workspace sections, mirroring carve_tile_ws() in csrc/common.h."""
HEADER_BYTES, SUBS, BIN_PARTS, CHUNK = 512, 64, 4, 1024


def a256(v):
    return (v + 255) & ~255


def tile_offsets(P, W, H):
    cells = ((W + 63) // 64) * ((H + 63) // 64)
    chunks = (P + CHUNK - 1) // CHUNK
    off, out = HEADER_BYTES, {}
    for name, size in (('chunk_cell', chunks * cells * 8), ('cell_cnt', cells * 8), ('cell_off', (cells + 1) * 8),
                       ('chunk_inst', (chunks + 1) * 4), ('chunk_vis', (chunks + 1) * 4), ('chunk_tiles', (chunks + 1) * 4),
                       ('chunk_off', (chunks + 1) * 4),
                       ('cell_desc', cells * 16), ('ranges', cells * SUBS * 8), ('slots', cells * SUBS * 16),
                       ('fwd_exit', cells * SUBS * 8), ('part_cnt', cells * BIN_PARTS * SUBS * 4), ('cell_long', cells * 4),
                       ('part_desc', cells * BIN_PARTS * 16)):
        out[name] = (off, size)
        off += a256(size)
    out['cells'], out['chunks'], out['total'] = cells, chunks, off
    return out
