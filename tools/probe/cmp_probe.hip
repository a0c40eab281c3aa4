// Developer probe: cost of rank-counting idioms on gfx950 (cycles per compare, one wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_u64(const unsigned long long* keys, unsigned* out, long long* cyc, int n) {
    unsigned long long mine[4]; unsigned rank[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) mine[r] = keys[threadIdx.x + 64 * r];
    __shared__ unsigned long long s[1024];
    for (int i = threadIdx.x; i < n; i += 64) s[i] = keys[i];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int j = 0; j < n; ++j) { unsigned long long k = s[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) rank[r] += (k < mine[r]) ? 1u : 0u; }
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 4; ++r) out[threadIdx.x + 64 * r] = rank[r];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_u32cmp(const unsigned long long* keys, unsigned* out, long long* cyc, int n) {
    unsigned mine[4]; unsigned rank[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) mine[r] = (unsigned)(keys[threadIdx.x + 64 * r] >> 32);
    __shared__ unsigned s[1024];
    for (int i = threadIdx.x; i < n; i += 64) s[i] = (unsigned)(keys[i] >> 32);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int j = 0; j < n; ++j) { unsigned k = s[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) rank[r] += (k < mine[r]) ? 1u : 0u; }
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 4; ++r) out[threadIdx.x + 64 * r] = rank[r];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_u32arith(const unsigned long long* keys, unsigned* out, long long* cyc, int n) {
    unsigned mine[4]; unsigned rank[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) mine[r] = (unsigned)(keys[threadIdx.x + 64 * r] >> 32);
    __shared__ unsigned s[1024];
    for (int i = threadIdx.x; i < n; i += 64) s[i] = (unsigned)(keys[i] >> 32);
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int j = 0; j < n; ++j) { unsigned k = s[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) rank[r] += (k - mine[r]) >> 31; }      // keys < 2^31
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 4; ++r) out[threadIdx.x + 64 * r] = rank[r];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    const int n = 1024;
    std::vector<unsigned long long> h(n);
    for (int i = 0; i < n; ++i) h[i] = ((unsigned long long)((i * 2654435761u) & 0x7fffffffu) << 32) | i;
    unsigned long long* dk; unsigned* dout; long long* dc;
    hipMalloc(&dk, n * 8); hipMalloc(&dout, 256 * 4); hipMalloc(&dc, 8);
    hipMemcpy(dk, h.data(), n * 8, hipMemcpyHostToDevice);
    long long c;
    for (int rep = 0; rep < 2; ++rep) {
        k_u64<<<1, 64>>>(dk, dout, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("u64 compare     : %lld cycles for %d x 4 compares -> %.1f cycles/compare\n", c, n, c / (4.0 * n));
        k_u32cmp<<<1, 64>>>(dk, dout, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("u32 compare     : %lld -> %.1f cycles/compare\n", c, c / (4.0 * n));
        k_u32arith<<<1, 64>>>(dk, dout, dc, n); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("u32 sub+shift   : %lld -> %.1f cycles/compare\n", c, c / (4.0 * n));
    }
    return 0;
}
