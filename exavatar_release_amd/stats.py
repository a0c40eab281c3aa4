"""Work counters of one render, read back from its workspaces (developer / benchmark helper; synchronises).

Mirrors ``carve_tile_ws`` / ``carve_bin_ws`` of ``csrc/common.h``.  Used by ``bench.py`` for the secondary (VALU) roofline
of SURVEY.md 8(d) -- "fp32 vector + v_exp_f32 throughput for the pixel-Gaussian evaluations per pass" -- and for the
per-sub-tile list-length histogram that section asks for with every result.  Needs ``config.keep_debug = True`` during the
render (``rasterizer._debug_last`` then holds the workspaces of the most recent forward)."""
import torch

HEADER_BYTES, SUBS, BIN_PARTS, CHUNK, BATCH = 2560, 64, 4, 1024, 64


def a256(v):
    return (v + 255) & ~255


def tile_offsets(P, W, H):
    """Byte (offset, size) of every section of the tile workspace."""
    cells = ((W + 63) // 64) * ((H + 63) // 64)
    chunks = (P + CHUNK - 1) // CHUNK
    off, out = HEADER_BYTES, {}
    for name, size in (('chunk_cell', chunks * cells * 8), ('cell_cnt', cells * 8), ('cell_off', (cells + 1) * 8),
                       ('chunk_inst', (chunks + 1) * 4), ('chunk_vis', (chunks + 1) * 4), ('chunk_tiles', (chunks + 1) * 4),
                       ('chunk_off', (chunks + 1) * 4),
                       ('cell_desc', cells * 16), ('ranges', cells * SUBS * 8), ('slots', cells * SUBS * 16),
                       ('fwd_exit', cells * SUBS * 8), ('part_cnt', cells * BIN_PARTS * SUBS * 4), ('cell_long', cells * 4),
                       ('part_desc', cells * BIN_PARTS * 16), ('cls_code', cells * SUBS)):
        out[name] = (off, size)
        off += a256(size)
    out['cells'], out['chunks'], out['total'] = cells, chunks, off
    return out


def bin_offsets(cap):
    """Byte offsets of the sections of the bin workspace of ``cap`` instances."""
    off, out = 0, {}
    for name, size in (('keys', cap * 8), ('sorted', cap * 4), ('bucket', cap * 16), ('owner', (cap // BATCH + 1) * 16),
                       ('bmask', (cap // BATCH + 1) * 8), ('touched', cap), ('ckpt', (cap // BATCH + 1) * 5 * 64 * 4)):
        out[name] = (off, size)
        off += a256(size)
    return out


LIST_BINS = (0, 1, 17, 33, 65, 129, 257, 513, 1025, 2049)      # lower edges of the list-length histogram


def render_stats(tile_ws, bin_ws, P, W, H, capacity):
    """Counters of the training render whose workspaces these are (uint8 tensors):

    ``lists``          non-empty 8x8 sub-tile lists;  ``list_hist``: their lengths binned at :data:`LIST_BINS`;  ``list_max``
    ``instances``      sum of the list lengths (exact-footprint sub-tile instances)
    ``fwd_walked``     entries of the batches the forward blend ENTERED (it leaves a list when all 64 pixels have stopped)
    ``fwd_pairs``      64 x fwd_walked: pixel-Gaussian evaluations of the forward blend (upper bound: it may leave a batch early)
    ``bwd_blended``    entries the forward blended into at least one pixel = what the backward blend replays
    ``bwd_pairs``      64 x bwd_blended
    """
    lay = tile_offsets(P, W, H)
    nsub = lay['cells'] * SUBS
    o = lay['fwd_exit'][0]
    ex = tile_ws[o:o + nsub * 8].view(torch.int32).view(-1, 2).to(torch.int64)
    n, entered = ex[:, 0], ex[:, 1]
    o = lay['ranges'][0]
    rg = tile_ws[o:o + nsub * 8].view(torch.int32).view(-1, 2).to(torch.int64)
    length = rg[:, 1] - rg[:, 0]
    walked = torch.where(length > 0, torch.minimum(n, entered * BATCH), torch.zeros_like(n))    # (padding sub-tiles: unwritten)
    b = bin_offsets(capacity)['bmask'][0]
    nslots = capacity // BATCH
    bm = bin_ws[b:b + nslots * 8].view(torch.int64)
    # popcount of the 64-bit masks (slots without work hold zero)
    x = bm.clone()
    cnt = torch.zeros_like(x)
    for _ in range(64):
        cnt += x & 1
        x = (x >> 1) & 0x7fffffffffffffff
    edges = torch.tensor(LIST_BINS + (1 << 40,), device=length.device)
    hist = torch.histc(torch.bucketize(length, edges, right=True).float() - 1, bins=len(LIST_BINS), min=0, max=len(LIST_BINS))
    return {'lists': int((length > 0).sum()), 'list_hist': [int(v) for v in hist.tolist()], 'list_max': int(length.max()),
            'instances': int(length.sum()), 'fwd_walked': int(walked.sum()), 'fwd_pairs': 64 * int(walked.sum()),
            'bwd_blended': int(cnt.sum()), 'bwd_pairs': 64 * int(cnt.sum())}
