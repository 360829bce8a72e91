// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS path's access patterns: kernels that move a known
// number of bytes the way k_lin does, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`; scripts/microbench/fetch_calib.sh
// divides the counters by the known bytes.  Patterns:
//   rows4   13 row arrays [row][N] read 4 B per lane (the settled launch's state rows: certificate, q0, fit word, plane)
//   pts16   one float4 per lane, coalesced (the source points)
//   settled rows4 + pts16 together = the 68 B per point of a settled k_lin launch
//   gather16 one float4 per lane at a random position of a large array (candidate / neighbour gathers)
//   write4  19 row arrays written 4 B per lane (the state rows a searched point writes)
// Two sizes each: N = 1 M points (the C4 launch: everything fits the 256 MB Infinity Cache) and N = 12 M (it does not).
// build: hipcc -O2 --offload-arch=gfx950 scripts/microbench/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(256) void calib_rows4(const uint32_t *__restrict__ rows, uint32_t n, size_t stride, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    if (i < n) {
#pragma unroll
        for (int r = 0; r < 13; ++r) acc += rows[(size_t)(6 + r) * stride + i];
    }
    if (acc == 0x12345u) out[i] = 1.f;
}
__global__ __launch_bounds__(256) void calib_pts16(const float4 *__restrict__ pts, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float4 p = make_float4(0, 0, 0, 0);
    if (i < n) p = pts[i];
    if (p.x + p.y + p.z + p.w == 12345.f) out[i] = 1.f;
}
__global__ __launch_bounds__(256) void calib_settled(const uint32_t *__restrict__ rows, const float4 *__restrict__ pts, uint32_t n, size_t stride, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    float4 p = make_float4(0, 0, 0, 0);
    if (i < n) {
        p = pts[i];
#pragma unroll
        for (int r = 0; r < 13; ++r) acc += rows[(size_t)(6 + r) * stride + i];
    }
    if (acc == 0x12345u && p.x == 1.f) out[i] = 1.f;
}
__global__ __launch_bounds__(256) void calib_gather16(const float4 *__restrict__ pts, uint32_t n, uint32_t m, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float4 p = make_float4(0, 0, 0, 0);
    if (i < n) {
        uint32_t hsh = i * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        p = pts[hsh % m];
    }
    if (p.x + p.y + p.z + p.w == 12345.f) out[i] = 1.f;
}
__global__ __launch_bounds__(256) void calib_write4(uint32_t *__restrict__ rows, uint32_t n, size_t stride, uint32_t v) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
#pragma unroll
        for (int r = 0; r < 19; ++r) rows[(size_t)r * stride + i] = v + r;
    }
}

int main(int argc, char **argv) {
    const uint32_t sizes[2] = {1000000u, 12000000u};
    const uint32_t nmax = sizes[1];
    const size_t stride = ((size_t)nmax + 63) & ~(size_t)63;
    uint32_t *rows; float4 *pts; float *out;
    if (hipMalloc(&rows, 19 * stride * 4) != hipSuccess || hipMalloc(&pts, (size_t)nmax * 16 + 64) != hipSuccess || hipMalloc(&out, (size_t)nmax * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(rows, 1, 19 * stride * 4); hipMemset(pts, 0, (size_t)nmax * 16); hipMemset(out, 0, (size_t)nmax * 4);
    hipDeviceSynchronize();
    printf("{\"kernels\": {\n");
    for (int s = 0; s < 2; ++s) {
        const uint32_t n = sizes[s];
        const size_t st = s == 0 ? (((size_t)n + 63) & ~(size_t)63) : stride;     // (the 1 M case with the stride k_lin's state has)
        const dim3 grid((n + 255) / 256), blk(256);
        const int reps = 3;
        for (int k = 0; k < reps; ++k) {
            hipLaunchKernelGGL(calib_rows4, grid, blk, 0, 0, rows, n, st, out);
            hipLaunchKernelGGL(calib_pts16, grid, blk, 0, 0, pts, n, out);
            hipLaunchKernelGGL(calib_settled, grid, blk, 0, 0, rows, pts, n, st, out);
            hipLaunchKernelGGL(calib_gather16, grid, blk, 0, 0, pts, n, nmax, out);
            hipLaunchKernelGGL(calib_write4, grid, blk, 0, 0, rows, n, st, (uint32_t)k);
            hipDeviceSynchronize();
        }
        printf("  \"n_%u\": {\"rows4_read_bytes\": %zu, \"pts16_read_bytes\": %zu, \"settled_read_bytes\": %zu, \"gather16_read_bytes_min\": %zu, \"gather16_read_bytes_lines128\": %zu, \"write4_bytes\": %zu, \"grid_x\": %u}%s\n",
               n, (size_t)13 * 4 * n, (size_t)16 * n, (size_t)68 * n, (size_t)16 * n, (size_t)128 * n, (size_t)19 * 4 * n, grid.x, s == 0 ? "," : "");
    }
    printf("}}\n");
    return 0;
}
