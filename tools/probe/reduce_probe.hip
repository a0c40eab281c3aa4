// Developer probe: checks the packed wave reduction of render_bwd.hip lane by lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define main_kernel_include
namespace exa_probe {
__device__ __forceinline__ float dpp_quad_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ float shr4(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true)); }
__device__ __forceinline__ float shr8(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true)); }
}
using namespace exa_probe;
__global__ void probe(const float* in, float* out) {
    const int lane = threadIdx.x;
    float x = in[lane];
    out[0 * 64 + lane] = dpp_quad_xor1(x);
    out[1 * 64 + lane] = dpp_quad_xor2(x);
    out[2 * 64 + lane] = shr4(x);
    out[3 * 64 + lane] = shr8(x);
    {
        float a = x, b = x + 100.0f;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        out[4 * 64 + lane] = a; out[5 * 64 + lane] = b;
    }
    {
        float a = x, b = x + 100.0f;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        out[6 * 64 + lane] = a; out[7 * 64 + lane] = b;
    }
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = (float)i;
    float *din, *dout; hipMalloc(&din, 256); hipMalloc(&dout, 8 * 256);
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, dout);
    std::vector<float> o(8 * 64); hipMemcpy(o.data(), dout, 8 * 256, hipMemcpyDeviceToHost);
    const char* names[8] = {"quad_xor1", "quad_xor2", "row_shr4", "row_shr8", "pl16swap[0]", "pl16swap[1]", "pl32swap[0]", "pl32swap[1]"};
    for (int r = 0; r < 8; ++r) { printf("%-12s", names[r]); for (int i = 0; i < 64; ++i) printf(" %3d", (int)o[r * 64 + i]); printf("\n"); }
    return 0;
}
