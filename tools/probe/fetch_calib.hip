// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS pipeline's access patterns
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports exactly 1/2 of a wide coalesced 16 B/lane stream; other widths and
// WRITE_SIZE are uncalibrated -- "calibrate on a known byte count in your own access pattern").  Three kernels over a
// 1 GiB buffer (larger than L2 + Infinity Cache), each with a known byte count:
//   stream_read   every lane reads 16 B, coalesced           (the per-Gaussian / per-pixel streams)
//   gather64      every lane reads 48 B of a random 64-B record (the splat-record gathers of the render kernels)
//   stream_read4  every lane reads 4 B, coalesced            (sorted ids, checkpoints, pixel gradients)
//   stream_read12 every lane reads 3 x 4 B at stride 12 B    (means3D / colours in the per-Gaussian kernels)
//   stream_write  every lane writes 16 B, coalesced
// Build: hipcc -w --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   and   rocprofv3 --kernel-trace --pmc WRITE_SIZE ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void stream_read(const float4* p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *sink = acc;
}
__global__ void stream_read4(const float* p, size_t n, float* sink) {          // 4 B per lane, coalesced (ids, checkpoints)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) *sink = acc;
}
__global__ void stream_read12(const float* p, size_t n3, float* sink) {        // 3 x 4 B per lane at stride 12 B (means3D, colours)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n3; i += (size_t)gridDim.x * blockDim.x) acc += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    if (acc == 123.456f) *sink = acc;
}
__global__ void gather64(const float4* p, size_t nrec, size_t reads, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < reads; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = (i * 2654435761ull + 12345ull) % nrec;          // pseudo-random record
        const float4* q = p + r * 4;
        float4 a = q[0], b = q[1], c = q[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void stream_write(float4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
    const size_t bytes = 1ull << 30, n16 = bytes / 16, nrec = bytes / 64, reads = 4u << 20;
    float4* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        stream_read<<<4096, 256>>>(buf, n16, sink);
        gather64<<<4096, 256>>>(buf, nrec, reads, sink);
        stream_read4<<<4096, 256>>>((const float*)buf, bytes / 4, sink);
        stream_read12<<<4096, 256>>>((const float*)buf, bytes / 12, sink);
        stream_write<<<4096, 256>>>(buf, n16);
    }
    hipDeviceSynchronize();
    printf("stream_read bytes %zu  gather64 useful bytes %zu (64-B lines touched %zu)  stream_write bytes %zu\n", bytes,
           reads * 48, reads * 64, bytes);
    return 0;
}
