/*
 * quad_stream_model.c -- CPU model of a candidate forward-blend schedule, NOT a product path and
 * not an oracle: test / design infrastructure like the rest of oracle/.
 *
 * Today's render_fwd_kernel (exavatar_release_amd/csrc/render_fwd.hip) gives one wave an 8x8-pixel sub-tile, stages the
 * sub-tile's depth-sorted list 64 entries at a time and lets all 64 lanes evaluate every staged entry, four entries per
 * loop trip, until every pixel of the sub-tile is dead.  Only ~20 % of those (pixel, splat) pairs contribute.  The
 * candidate keeps lists, batches and the per-pixel rule, but lets the four 4x4-pixel QUADS of the sub-tile (16 lanes
 * each) walk a staged batch as four independent streams: every staged entry gets a 4-bit mask of the quads its
 * {alpha >= 1/255} bounding box reaches -- the same conservative box preprocess_fwd.hip uses for the sub-tile rect --
 * each quad pulls only its own entries (in list order, four per trip, padded with null entries), and a quad stops as
 * soon as ITS 16 pixels are dead.  A batch costs as many trips as its longest quad stream.
 *
 * This file executes exactly that control flow on the CPU, with the blend arithmetic of csrc/blend.h (partial products
 * of four, stop rule on T P_j), so that (a) the images can be held against the oracle -- a quad mask that dropped a
 * contributing entry, a padding or stop-rule mistake would show -- and (b) loop trips are counted exactly, including
 * the group-of-four granularity and the per-batch restarts that a per-entry count ignores.
 *
 * Sub-tile lists: the entries of the 16x16 tile's list that pass the per-pixel alpha test at >= 1 pixel of the sub-tile
 * (the product's exact-footprint lists: 709 k vs 710 k entries on C3).
 */
#include "raster_oracle.c"

#define SUB 8
#define BATCH 64

typedef struct { float T, live, C[3], D; } Pix;

/* alpha of entry g at pixel (fx, fy): the oracle's rule (skip -> 0) */
static inline float entry_alpha(const Geo* g, float fx, float fy) {
    float dx, dy;
    const float power = gauss_power(g, fx, fy, &dx, &dy);
    if (power > 0.0f) return 0.0f;
    const float a = fminf(ALPHA_MAX, g->opacity * expf(power));
    return a < ALPHA_MIN ? 0.0f : a;
}

/* csrc/blend.h::blend_group4 for one pixel: four entries through partial products; the entry that would push
 * T (1 - a) under 1e-4 is not blended and kills the pixel */
static inline void blend_group4_px(Pix* p, const float alpha[4], const Geo* const g[4]) {
    float a[4], Tb[4], w[4];
    for (int j = 0; j < 4; ++j) a[j] = alpha[j] * p->live;
    const float P1 = 1.0f - a[0], P2 = P1 * (1.0f - a[1]), P3 = P2 * (1.0f - a[2]), P4 = P3 * (1.0f - a[3]);
    Tb[0] = p->T; Tb[1] = p->T * P1; Tb[2] = p->T * P2; Tb[3] = p->T * P3;
    const float T4 = p->T * P4;
    for (int j = 0; j < 4; ++j) w[j] = a[j] * Tb[j];
    if (T4 < T_EPS) {
        const float after[4] = {Tb[1], Tb[2], Tb[3], T4};
        float Tn = p->T;
        for (int j = 0; j < 4; ++j) {
            const int stop = after[j] < T_EPS;
            if (stop) w[j] = 0.0f; else Tn = after[j];
        }
        p->live = 0.0f;
        p->T = Tn;
    } else {
        p->T = T4;
    }
    for (int j = 0; j < 4; ++j)
        if (g[j]) {
            p->C[0] += g[j]->col[0] * w[j]; p->C[1] += g[j]->col[1] * w[j]; p->C[2] += g[j]->col[2] * w[j];
            p->D += g[j]->depth * w[j];
        }
}

/* 4-bit mask of the quads of sub-tile (ox, oy) that the conservative {alpha >= 1/255} box of g reaches
 * (preprocess_fwd.hip: half extents sqrt((2 ln(255 o) + 1e-3) cov_xx) * 1.001 + 0.01 px); cov from the conic */
static inline unsigned quad_mask_box(const Geo* g, int ox, int oy) {
    const float o255 = 255.0f * g->opacity;
    if (!(o255 >= 1.0f)) return 0u;
    const float idet = 1.0f / (g->A * g->C - g->B * g->B);
    const float sxx = g->C * idet, syy = g->A * idet;
    const float tau2 = 2.0f * logf(o255) + 1e-3f;
    const float ex = sqrtf(tau2 * sxx) * 1.001f + 0.01f, ey = sqrtf(tau2 * syy) * 1.001f + 0.01f;
    unsigned m = 0u;
    for (int q = 0; q < 4; ++q) {
        const float x0 = (float)(ox + 4 * (q & 1)), y0 = (float)(oy + 4 * (q >> 1));      /* pixel centres x0 .. x0 + 3 */
        if (g->px + ex >= x0 && g->px - ex <= x0 + 3.0f && g->py + ey >= y0 && g->py - ey <= y0 + 3.0f) m |= 1u << q;
    }
    return m;
}

/*
 * stats[0] sub-tile list entries   [1] entries walked today (until all 64 pixels dead)   [2] loop trips today (groups of 4)
 * stats[3] loop trips of the quad-stream schedule with box masks   [4] ... with exact masks (an entry is in a quad's
 * stream iff it passes the alpha test at one of the quad's pixels: the bound)   [5] unused
 * [6] non-empty sub-tiles   [7] entries whose box mask covers all four quads
 * mode: 0 = today's schedule, 1 = quad streams with box masks, 2 = quad streams with exact masks (which one renders the
 * image; the trip counts of all three are always produced).
 */
long exa_model_quad_stream_forward(const OrcSettings* s, int32_t P, const float* means3D, const float* colors_precomp,
                                   const float* opacities, const float* scales, const float* rotations, int32_t mode,
                                   float* out_color, float* out_depth, float* out_alpha, long* stats) {
    if (!s || P < 0 || !stats || mode < 0 || mode > 2) return -1;
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, n_tiles = gx * gy;
    Geo* geo = (Geo*)malloc(sizeof(Geo) * (size_t)(P > 0 ? P : 1));
    long* tile_off = (long*)calloc((size_t)n_tiles + 1, sizeof(long));
    if (!geo || !tile_off) return -2;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        preprocess_one(s, i, means3D, opacities, scales, rotations, NULL, gx, gy, &geo[i]);
        geo[i].col[0] = colors_precomp[i * 3]; geo[i].col[1] = colors_precomp[i * 3 + 1]; geo[i].col[2] = colors_precomp[i * 3 + 2];
    }
    for (int i = 0; i < P; ++i)
        if (geo[i].radius > 0)
            for (int ty = geo[i].y0; ty < geo[i].y1; ++ty)
                for (int tx = geo[i].x0; tx < geo[i].x1; ++tx) ++tile_off[ty * gx + tx + 1];
    for (int t = 0; t < n_tiles; ++t) tile_off[t + 1] += tile_off[t];
    const long D = tile_off[n_tiles];
    Key* keys = (Key*)malloc(sizeof(Key) * (size_t)(D > 0 ? D : 1));
    long* cur = (long*)malloc(sizeof(long) * (size_t)(n_tiles > 0 ? n_tiles : 1));
    if (!keys || !cur) return -2;
    memcpy(cur, tile_off, sizeof(long) * (size_t)n_tiles);
    for (int i = 0; i < P; ++i)
        if (geo[i].radius > 0) {
            uint32_t bits;
            memcpy(&bits, &geo[i].depth, 4);
            for (int ty = geo[i].y0; ty < geo[i].y1; ++ty)
                for (int tx = geo[i].x0; tx < geo[i].x1; ++tx) {
                    Key* k = &keys[cur[ty * gx + tx]++];
                    k->depth_bits = bits; k->id = i;
                }
        }
    free(cur);
    long max_list = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const long n = tile_off[t + 1] - tile_off[t];
        if (n > 1) qsort(keys + tile_off[t], (size_t)n, sizeof(Key), key_cmp);
        if (n > max_list) max_list = n;
    }
    long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t HW = (size_t)W * H;

#pragma omp parallel
    {
        long lst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const Geo** list = (const Geo**)malloc(sizeof(Geo*) * (size_t)(max_list > 0 ? max_list : 1));
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < n_tiles; ++t) {
            const int tx = t % gx, ty = t / gx;
            const Key* tl = keys + tile_off[t];
            const long tn = tile_off[t + 1] - tile_off[t];
            for (int sub = 0; sub < 4; ++sub) {
                const int ox = tx * TILE + SUB * (sub & 1), oy = ty * TILE + SUB * (sub >> 1);
                if (ox >= W || oy >= H) continue;
                /* exact-footprint list of this sub-tile */
                long n = 0;
                for (long k = 0; k < tn; ++k) {
                    const Geo* g = &geo[tl[k].id];
                    int hit = 0;
                    for (int j = 0; j < SUB && !hit; ++j)
                        for (int i = 0; i < SUB && !hit; ++i)
                            if (ox + i < W && oy + j < H && entry_alpha(g, (float)(ox + i), (float)(oy + j)) > 0.0f) hit = 1;
                    if (hit) list[n++] = g;
                }
                lst[0] += n;
                if (n) ++lst[6];
                Pix px_now[64], px_box[64], px_exact[64];
                for (int l = 0; l < 64; ++l) {
                    const int inside = ox + (l & 7) < W && oy + (l >> 3) < H;
                    const Pix init = {1.0f, inside ? 1.0f : 0.0f, {0.f, 0.f, 0.f}, 0.f};
                    px_now[l] = init; px_box[l] = init; px_exact[l] = init;
                }
                for (int variant = 0; variant < 3; ++variant) {
                    Pix* px = variant == 0 ? px_now : (variant == 1 ? px_box : px_exact);
                    for (long base = 0; base < n; base += BATCH) {
                        int any_live = 0;
                        for (int l = 0; l < 64; ++l) any_live |= px[l].live != 0.0f;
                        if (!any_live) break;
                        const int cnt = (int)(n - base < BATCH ? n - base : BATCH);
                        if (variant == 0) {
                            /* today: all 64 lanes walk the batch four entries per trip until every pixel is dead */
                            for (int k = 0; k < cnt; k += 4) {
                                const Geo* g4[4];
                                for (int u = 0; u < 4; ++u) g4[u] = k + u < cnt ? list[base + k + u] : NULL;
                                ++lst[2];
                                lst[1] += (k + 4 <= cnt ? 4 : cnt - k);
                                int live_after = 0;
                                for (int l = 0; l < 64; ++l) {
                                    float al[4];
                                    for (int u = 0; u < 4; ++u)
                                        al[u] = g4[u] ? entry_alpha(g4[u], (float)(ox + (l & 7)), (float)(oy + (l >> 3))) : 0.0f;
                                    blend_group4_px(&px[l], al, g4);
                                    live_after |= px[l].live != 0.0f;
                                }
                                if (!live_after) break;
                            }
                        } else {
                            /* quad streams: per staged entry a 4-bit quad mask, per quad its own compacted stream */
                            unsigned qm[BATCH];
                            for (int e = 0; e < cnt; ++e) {
                                const Geo* g = list[base + e];
                                if (variant == 1) {
                                    qm[e] = quad_mask_box(g, ox, oy);
                                    if (qm[e] == 15u) ++lst[7];
                                } else {
                                    qm[e] = 0u;
                                    for (int l = 0; l < 64; ++l)
                                        if (ox + (l & 7) < W && oy + (l >> 3) < H &&
                                            entry_alpha(g, (float)(ox + (l & 7)), (float)(oy + (l >> 3))) > 0.0f)
                                            qm[e] |= 1u << (((l >> 2) & 1) | (((l >> 5) & 1) << 1));
                                }
                            }
                            int trips = 0;
                            for (int q = 0; q < 4; ++q) {
                                int stream[BATCH], m = 0, qtrips = 0;
                                for (int e = 0; e < cnt; ++e) if (qm[e] & (1u << q)) stream[m++] = e;
                                for (int k = 0; k < m; k += 4) {
                                    int alive = 0;
                                    for (int l = 0; l < 64; ++l)
                                        if ((((l >> 2) & 1) | (((l >> 5) & 1) << 1)) == q) alive |= px[l].live != 0.0f;
                                    if (!alive) break;                       /* this quad's 16 pixels are dead */
                                    ++qtrips;
                                    const Geo* g4[4];
                                    for (int u = 0; u < 4; ++u) g4[u] = k + u < m ? list[base + stream[k + u]] : NULL;
                                    for (int l = 0; l < 64; ++l) {
                                        if ((((l >> 2) & 1) | (((l >> 5) & 1) << 1)) != q) continue;
                                        float al[4];
                                        for (int u = 0; u < 4; ++u)
                                            al[u] = g4[u] ? entry_alpha(g4[u], (float)(ox + (l & 7)), (float)(oy + (l >> 3))) : 0.0f;
                                        blend_group4_px(&px[l], al, g4);
                                    }
                                }
                                if (qtrips > trips) trips = qtrips;
                            }
                            lst[variant == 1 ? 3 : 4] += trips;
                        }
                    }
                }
                const Pix* out = mode == 0 ? px_now : (mode == 1 ? px_box : px_exact);
                for (int l = 0; l < 64; ++l) {
                    const int i = ox + (l & 7), j = oy + (l >> 3);
                    if (i >= W || j >= H) continue;
                    const size_t pix = (size_t)j * W + i;
                    out_color[pix] = out[l].C[0] + out[l].T * s->bg[0];
                    out_color[HW + pix] = out[l].C[1] + out[l].T * s->bg[1];
                    out_color[2 * HW + pix] = out[l].C[2] + out[l].T * s->bg[2];
                    out_depth[pix] = out[l].D;
                    out_alpha[pix] = 1.0f - out[l].T;
                }
            }
        }
#pragma omp critical
        for (int i = 0; i < 8; ++i) st[i] += lst[i];
        free(list);
    }
    /* pixels of tiles without any list: background */
    (void)D;
    for (int i = 0; i < 8; ++i) stats[i] = st[i];
    free(keys); free(tile_off); free(geo);
    return 0;
}
