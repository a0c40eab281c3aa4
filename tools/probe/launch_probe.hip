// Developer probe: how long does it take just to launch N tiny workgroups on MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS> __global__ void k_empty(int* p) {
    __shared__ int s[LDS > 0 ? LDS : 1];
    if (LDS > 0) { s[threadIdx.x % LDS] = threadIdx.x; }
    if (p && blockIdx.x == 0x7fffffff) p[0] = s[0];
}
template <int LDS> float run(int grid, int block) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_empty<LDS><<<grid, block>>>(nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) k_empty<LDS><<<grid, block>>>(nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}
int main() {
    int grids[] = {1024, 4096, 16384, 65536};
    for (int g : grids) {
        printf("grid %6d: 64thr/noLDS %.1f us | 64thr/3KB %.1f | 64thr/8KB %.1f | 256thr/noLDS %.1f | 256thr/12KB %.1f\n", g,
               run<0>(g, 64), run<768>(g, 64), run<2048>(g, 64), run<0>(g, 256), run<3072>(g, 256));
    }
    return 0;
}
