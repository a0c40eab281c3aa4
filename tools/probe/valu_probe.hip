// Developer probe: sustained VALU issue rate of one gfx950 SIMD versus waves per SIMD, for plain fp32 fma,
// packed fp32 fma, v_exp_f32 and broadcast ds_read_b128 -- the numbers the render kernels are budgeted with.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000, CH = 16;
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
    __shared__ float4 s[64];
    s[threadIdx.x] = make_float4(seed, seed, seed, seed);
    float x[CH]; v2f y[CH / 2];
    for (int i = 0; i < CH; ++i) x[i] = seed + i + threadIdx.x;
    for (int i = 0; i < CH / 2; ++i) y[i] = v2f{seed + i, seed - i};
    __syncthreads();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < CH / 2; ++i) y[i] = __builtin_elementwise_fma(y[i], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f});
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < CH; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { const float4 v = s[(it + i) & 63]; x[i] += v.x; asm volatile("" : "+v"(x[i])); }
        } else if (MODE == 4) {      // v_cmp + v_cndmask pairs
#pragma unroll
            for (int i = 0; i < CH; ++i) x[i] = x[i] < 3.0f ? x[i] + 1.0f : seed;
        }
    }
    float acc = 0.f;
    for (int i = 0; i < CH; ++i) acc += x[i];
    for (int i = 0; i < CH / 2; ++i) acc += y[i].x + y[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, int per_inst_results, int insts_per_iter) {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * 4 * w;
        k<MODE><<<blocks, 64>>>(d, 1.0f); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<blocks, 64>>>(d, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)w * ITERS * insts_per_iter;
        printf("%-22s waves/SIMD %d: %.1f us -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, w, ms * 1e3,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 1, CH);
    run<1>("v_pk_fma_f32", 2, CH / 2);
    run<2>("v_exp_f32", 1, CH);
    run<3>("ds_read_b128 bcast+add", 1, CH);
    run<4>("v_cmp+v_cndmask+v_add", 1, CH);
    return 0;
}
