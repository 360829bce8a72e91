// Issue cost of wave64 VALU instruction classes on gfx950, measured: every wave runs a long stream of ONE instruction over 8
// independent registers (no dependency stalls: what is left is the issue rate) and reads the shader clock (s_memtime) around it.
// W waves per SIMD run the same stream; SIMD cycles per wave-instruction = elapsed / (W x instructions) is reported at W = 1, 2, 4 -
// W = 1 shows how fast ONE wave can issue, the value at saturation is what one instruction costs its SIMD.  MI355X_MICROARCH.md gives 2 cycles for v_fma_f32 (SIMD-32, two
// passes); fp64 and the transcendental pipe are not in its table.
// build + run (GPU box):  hipcc -O2 --offload-arch=gfx950 scripts/microbench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA_F32, ADD_U32, CNDMASK, MED3_F32, CMP_F32, PK_MUL_F32, FMA_F64, MUL_F64, ADD_F64, RCP_F64, RSQ_F64, RCP_F32, SQRT_F32, CVT_F64_F32, CVT_F32_F64, MAX_F64, N_OPS };
static const char *kNames[N_OPS] = {"v_fma_f32", "v_add_u32", "v_cndmask_b32", "v_med3_f32", "v_cmp_lt_f32 (e64, sgpr pair)", "v_pk_mul_f32", "v_fma_f64", "v_mul_f64", "v_add_f64",
                                   "v_rcp_f64", "v_rsq_f64", "v_rcp_f32", "v_sqrt_f32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_max_f64"};

template <int OP>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long *cyc, float *sink, int iters) {
    float f[8]; double d[8]; unsigned u[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int k = 0; k < 8; ++k) { f[k] = 1.0f + threadIdx.x * 1e-3f + k; d[k] = 1.0 + threadIdx.x * 1e-3 + k; u[k] = threadIdx.x + k; p[k] = f2{f[k], f[k] + 1.f}; }
    const float fb = 1.0000001f, fc = 1e-9f;
    const double db = 1.0000000001, dc = 1e-12;
    unsigned long long m;
    const unsigned long long msk = 0x5555AAAA3333CCCCull + blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define X(k)                                                                                                                          \
    if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));                                     \
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));                                       \
    if (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "s"(msk));                 \
    if (OP == MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));                                    \
    if (OP == CMP_F32) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f[k]), "v"(fb));                                   \
    if (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(p[(k + 1) & 7]));                                 \
    if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(db), "v"(dc));                                      \
    if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(db));                                                   \
    if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));                                                   \
    if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));                                                                 \
    if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));                                                                 \
    if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));                                                                 \
    if (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));                                                               \
    if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));                                            \
    if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));                                            \
    if (OP == MAX_F64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[k]) : "v"(db));
            REP8(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int k = 0; k < 8; ++k) acc += f[k] + (float)d[k] + (float)u[k] + p[k].x + p[k].y;
    if (OP == CMP_F32) acc += (float)(m & 1ull);
    if (acc == 123.456f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
static double run(int waves_per_simd, int iters, unsigned long long *d_cyc, float *d_sink, int n_cu) {
    // one block of 64 * 4 * W threads per CU: W waves on each of the CU's 4 SIMDs (W = 8: two blocks of 1024 threads per CU)
    const int blocks_per_cu = waves_per_simd > 4 ? 2 : 1;
    const int threads = 64 * 4 * waves_per_simd / blocks_per_cu;
    n_cu *= blocks_per_cu;
    hipLaunchKernelGGL(k_issue<OP>, dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink, iters);
    hipLaunchKernelGGL(k_issue<OP>, dim3(n_cu), dim3(threads), 0, 0, d_cyc, d_sink, iters);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { fprintf(stderr, "launch failed\n"); exit(1); }
    const int nw = n_cu * threads / 64;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), d_cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= nw;
    return mean / ((double)iters * 32.0 * waves_per_simd);          // cycles of the SIMD per wave-instruction
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *d_cyc; float *d_sink;
    hipMalloc(&d_cyc, sizeof(unsigned long long) * n_cu * 64);
    hipMalloc(&d_sink, 64);
    const int iters = 2000;
    // __builtin_readcyclecounter = s_memtime: a 100 MHz constant clock on gfx950?  calibrate against the shader clock with v_fma_f32
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"unit\": \"shader-clock cycles (s_memtime) of one SIMD per wave64 instruction, W waves per SIMD issuing\", \"ops\": {\n", prop.name, n_cu, prop.clockRate / 1000);
#define RUN(OP)                                                                                                                        \
    {                                                                                                                                  \
        double c1 = run<OP>(1, iters, d_cyc, d_sink, n_cu), c2 = run<OP>(2, iters, d_cyc, d_sink, n_cu), c4 = run<OP>(4, iters, d_cyc, d_sink, n_cu), c8 = run<OP>(8, iters, d_cyc, d_sink, n_cu); \
        printf("  \"%s\": {\"w1\": %.3f, \"w2\": %.3f, \"w4\": %.3f, \"w8\": %.3f}%s\n", kNames[OP], c1, c2, c4, c8, OP == N_OPS - 1 ? "" : ",");       \
    }
    RUN(FMA_F32) RUN(ADD_U32) RUN(CNDMASK) RUN(MED3_F32) RUN(CMP_F32) RUN(PK_MUL_F32) RUN(FMA_F64) RUN(MUL_F64) RUN(ADD_F64) RUN(RCP_F64) RUN(RSQ_F64)
    RUN(RCP_F32) RUN(SQRT_F32) RUN(CVT_F64_F32) RUN(CVT_F32_F64) RUN(MAX_F64)
    printf("}}\n");
    return 0;
}
