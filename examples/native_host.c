/*
 * A native host of the C ABI (include/exa_raster.h): plain C99 + the HIP runtime API, no torch, no Python.
 * What a C / C++ trainer's inner loop does with the library: device buffers it owns, workspaces sized by
 * exa_raster_workspace_sizes, upstream's two-stage protocol (stage 1, read the instance count, stage 2) for the first
 * render of a configuration, then the fused call with that capacity, then the backward -- on a stream it owns.
 *
 *   gcc -std=c99 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/native_host.c \
 *       exavatar_release_amd/libexa_raster.so -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o native_host
 *   ./native_host scene.bin out.bin
 *
 * scene.bin (little endian): int32 P, H, W; float tanfovx, tanfovy; float bg[3], view[16], proj[16], campos[3];
 * float means3D[P*3], scales[P*3], rotations[P*4], opacities[P], colors[P*3], dL_dcolor[3*H*W].
 * out.bin: float color[3*H*W], depth[H*W], alpha[H*W]; int32 radii[P]; float dL_dmeans3D[P*3], dL_dmeans2D[P*3],
 * dL_dcolors[P*3], dL_dopacity[P], dL_dscales[P*3], dL_drotations[P*4]; then the same images again from the fused call.
 * tests/test_gpu_native_host.py holds out.bin against the Python surface bit for bit.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "exa_raster.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_EXA(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "exa_raster status %d: %s (%s:%d)\n", r_, exa_raster_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static void* upload(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
    if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[3];
    float cam[2 + 3 + 16 + 16 + 3];
    if (fread(hdr, 4, 3, f) != 3 || fread(cam, 4, 40, f) != 40) return 1;
    const int32_t P = hdr[0], H = hdr[1], W = hdr[2];
    const size_t HW = (size_t)H * W, n_in = (size_t)P * (3 + 3 + 4 + 1 + 3) + 3 * HW;
    float* in = (float*)malloc(n_in * 4);
    if (!in || fread(in, 4, n_in, f) != n_in) return 1;
    fclose(f);
    const float *h_m3 = in, *h_sc = h_m3 + 3 * P, *h_rot = h_sc + 3 * P, *h_op = h_rot + 4 * P, *h_col = h_op + P, *h_g = h_col + 3 * P;

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    float* d_cam = (float*)upload(cam + 2, 38 * 4);                   /* bg 3 | view 16 | proj 16 | campos 3 */
    float *m3 = (float*)upload(h_m3, 12u * P), *sc = (float*)upload(h_sc, 12u * P), *rot = (float*)upload(h_rot, 16u * P);
    float *op = (float*)upload(h_op, 4u * P), *col = (float*)upload(h_col, 12u * P), *g = (float*)upload(h_g, 12 * HW);
    if (!d_cam || !m3 || !sc || !rot || !op || !col || !g) return 2;

    ExaRasterSettings s;
    s.image_height = H; s.image_width = W; s.tanfovx = cam[0]; s.tanfovy = cam[1];
    s.bg = d_cam; s.scale_modifier = 1.0f; s.viewmatrix = d_cam + 3; s.projmatrix = d_cam + 19; s.sh_degree = 0;
    s.campos = d_cam + 35; s.prefiltered = 0; s.debug = 0;

    /* outputs and workspaces: owned by the caller */
    float *color, *depth, *alpha, *grads;
    int32_t* radii;
    CHECK_HIP(hipMalloc((void**)&color, 12 * HW)); CHECK_HIP(hipMalloc((void**)&depth, 4 * HW)); CHECK_HIP(hipMalloc((void**)&alpha, 4 * HW));
    CHECK_HIP(hipMalloc((void**)&radii, 4u * P + 4)); CHECK_HIP(hipMalloc((void**)&grads, 4u * 17 * P + 4));
    ExaRasterWorkspaceSizes z;
    CHECK_EXA(exa_raster_workspace_sizes(P, W, H, 0, &z));
    void *geom, *tile, *bin, *gws;
    CHECK_HIP(hipMalloc(&geom, z.geom_bytes + 4)); CHECK_HIP(hipMalloc(&tile, z.tile_bytes + 4));

    /* render 1: upstream's protocol -- stage 1, read the instance count (16-byte D2H copy), allocate exactly, stage 2 */
    CHECK_EXA(exa_raster_forward_bin(&s, P, 0, m3, NULL, col, op, sc, rot, NULL, radii, geom, tile, st));
    ExaRasterHeader hd;
    CHECK_HIP(hipMemcpyAsync(&hd, tile, sizeof hd, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_EXA(exa_raster_header_status(&hd));
    const uint64_t capacity = hd.num_rendered > 64 ? hd.num_rendered : 64;
    CHECK_EXA(exa_raster_workspace_sizes(P, W, H, capacity, &z));
    CHECK_HIP(hipMalloc(&bin, z.bin_bytes + 4)); CHECK_HIP(hipMalloc(&gws, z.grad_bytes + 4));
    CHECK_EXA(exa_raster_forward_render(&s, P, geom, tile, bin, capacity, color, depth, alpha, 1, st));
    float *d3 = grads, *d2 = d3 + 3 * P, *dc = d2 + 3 * P, *dop = dc + 3 * P, *dsc = dop + P, *drot = dsc + 3 * P;
    CHECK_EXA(exa_raster_backward(&s, P, 0, m3, NULL, col, op, sc, rot, NULL, radii, geom, tile, bin, capacity, g, NULL, NULL, gws,
                                  d2, d3, dc, dop, dsc, drot, NULL, NULL, st));
    CHECK_HIP(hipStreamSynchronize(st));

    const size_t n_out = 5 * HW + P + 17u * P + 5 * HW;
    float* out = (float*)malloc(n_out * 4);
    float* o = out;
    CHECK_HIP(hipMemcpy(o, color, 12 * HW, hipMemcpyDeviceToHost)); o += 3 * HW;
    CHECK_HIP(hipMemcpy(o, depth, 4 * HW, hipMemcpyDeviceToHost)); o += HW;
    CHECK_HIP(hipMemcpy(o, alpha, 4 * HW, hipMemcpyDeviceToHost)); o += HW;
    CHECK_HIP(hipMemcpy(o, radii, 4u * P, hipMemcpyDeviceToHost)); o += P;
    CHECK_HIP(hipMemcpy(o, grads, 4u * 17 * P, hipMemcpyDeviceToHost)); o += 17u * P;

    /* render 2: the fused call with the capacity the first render measured (no host round trip; the header says whether it fit) */
    CHECK_HIP(hipMemsetAsync(color, 0xff, 12 * HW, st));
    CHECK_EXA(exa_raster_forward(&s, P, 0, m3, NULL, col, op, sc, rot, NULL, radii, geom, tile, bin, capacity, color, depth, alpha, 0, st));
    CHECK_HIP(hipMemcpyAsync(&hd, tile, sizeof hd, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_EXA(exa_raster_header_status(&hd));
    CHECK_HIP(hipMemcpy(o, color, 12 * HW, hipMemcpyDeviceToHost)); o += 3 * HW;
    CHECK_HIP(hipMemcpy(o, depth, 4 * HW, hipMemcpyDeviceToHost)); o += HW;
    CHECK_HIP(hipMemcpy(o, alpha, 4 * HW, hipMemcpyDeviceToHost)); o += HW;

    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 4, n_out, f) != n_out) { perror(argv[2]); return 1; }
    fclose(f);
    printf("native_host: ABI %d, P %d, %dx%d, %u instances (capacity %llu), visible %u\n", exa_raster_version(), P, W, H,
           hd.num_rendered, (unsigned long long)capacity, hd.num_visible);
    return 0;
}
