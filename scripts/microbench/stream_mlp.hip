// How fast can a kernel shaped like the settled linearisation stream its 72 B per point?  One thread reads 13 dwords from 13 row arrays
// ([row][N], as the neighbour state) + one float4, at the occupancy of k_lin (4 waves per SIMD: 39.5 KB of LDS per 256-thread block),
// handling P = 1, 2 or 4 points per thread with all loads of its P points issued before the first use.
// build: hipcc -O2 --offload-arch=gfx950 scripts/microbench/stream_mlp.hip -o /tmp/stream_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int P, int LDS_BYTES>
__global__ __launch_bounds__(256) void k_stream(const uint32_t *__restrict__ rows, const float4 *__restrict__ pts, uint32_t n, size_t stride, float *__restrict__ out) {
    __shared__ char pad[LDS_BYTES];
    if (threadIdx.x == 1000) pad[threadIdx.x] = 1;             // keep the allocation
    const uint32_t base = (blockIdx.x * 256u * P) + threadIdx.x;
    uint32_t w[P][13];
    float4 p[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const uint32_t i = base + 256u * k;
        const bool ok = i < n;
        p[k] = ok ? pts[i] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 13; ++r) w[k][r] = ok ? rows[(size_t)(6 + r) * stride + i] : 0u;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        float a = p[k].x + p[k].y * p[k].z + p[k].w;
#pragma unroll
        for (int r = 0; r < 13; ++r) a = a * 1.0001f + __uint_as_float(w[k][r] & 0x3FFFFFFFu);
        acc += a;
    }
    // wave reduction + one store per wave (stands for the Gram reduction)
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256u + threadIdx.x) >> 6] = acc + (float)pad[0];
}
template <int P, int LDS>
static void run(const char *name, const uint32_t *rows, const float4 *pts, uint32_t n, size_t stride, float *out) {
    const uint32_t blocks = (n + 256 * P - 1) / (256 * P);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((k_stream<P, LDS>), dim3(blocks), dim3(256), 0, 0, rows, pts, n, stride, out);
    hipEventRecord(e0);
    const int reps = 50;
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((k_stream<P, LDS>), dim3(blocks), dim3(256), 0, 0, rows, pts, n, stride, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps;
    printf("%-40s %6.2f us per launch = %.2f TB/s of 68 B per point\n", name, us, 68.0 * n / us * 1e-6);
}
int main() {
    const uint32_t n = 1000000; const size_t stride = (n + 63) & ~63u;
    uint32_t *rows; float4 *pts; float *out;
    hipMalloc(&rows, 19 * stride * 4); hipMalloc(&pts, n * 16 + 64); hipMalloc(&out, n);
    hipMemset(rows, 1, 19 * stride * 4); hipMemset(pts, 0, n * 16);
    run<1, 39000>("1 point / thread, 4 waves / SIMD (k_lin)", rows, pts, n, stride, out);
    run<2, 39000>("2 points / thread, 4 waves / SIMD", rows, pts, n, stride, out);
    run<4, 39000>("4 points / thread, 4 waves / SIMD", rows, pts, n, stride, out);
    run<1, 19000>("1 point / thread, 8 waves / SIMD", rows, pts, n, stride, out);
    run<2, 19000>("2 points / thread, 8 waves / SIMD", rows, pts, n, stride, out);
    run<1, 1024>("1 point / thread, LDS-unlimited", rows, pts, n, stride, out);
    return 0;
}
