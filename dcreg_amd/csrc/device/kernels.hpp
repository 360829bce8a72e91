// gfx950 kernels of the DCReg hot path: fused exact 5-NN + plane fit + point-to-plane row + J^T J / J^T r
// reduction (DCReg/src/icp_test_runner.cpp:1714-1915), plus the index-build and k-NN utility kernels.
// The per-thread search / plane-fit / row functions live in search.hpp.
//   partial: double[pose][block][32]  (21 H + 6 g + sum r^2 + sum b^2 + n_eff + n_pt + pad)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "search.hpp"

namespace dcreg {

// ---------------------------------------------------------------- wave64 sum in lane 63 via DPP
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    int lo = (int)(uint32_t)b, hi = (int)(uint32_t)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    // for the row-masked broadcasts the disabled rows keep "old" = 0 -> they add 0
    const double o = __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
    return v + o;
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of each row = row total
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 = wave total
    return v;
}

// Transposed wave reduction of 32 per-lane values: a halving butterfly.  At step `bit` every lane keeps half of its
// values (those whose index has that bit equal to the lane's bit) and adds the partner lane's copies, so the
// number of live values halves while the number of lanes summed doubles: 16+8+4+2+1 exchanges instead of 32 x 6.
// After the five halving steps lane l holds value (l & 31) summed over its 32-lane half; one more exchange adds
// the halves.  Result: EVERY lane l returns the wave total of value (l & 31).  Fixed order -> deterministic.
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), m);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <int N, int BIT>
__device__ __forceinline__ void halve_step(double (&v)[32], int lane) {
    const bool up = (lane >> BIT) & 1;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const double keep = up ? v[2 * i + 1] : v[2 * i];
        const double send = up ? v[2 * i] : v[2 * i + 1];
        v[i] = keep + shfl_xor_f64(send, 1 << BIT);
    }
}
__device__ __forceinline__ double wave_transpose_reduce32(double (&v)[32], int lane) {
    // index bit b of the value ends up selected by lane bit b: process index bit 0 with lane bit 0 first
    halve_step<32, 0>(v, lane);   // v[i] now = value 2i + b0
    halve_step<16, 1>(v, lane);   // value 4i + 2 b1 + b0
    halve_step<8, 2>(v, lane);
    halve_step<4, 3>(v, lane);
    halve_step<2, 4>(v, lane);    // v[0] = value (lane & 31) over the lanes sharing bit 5
    return v[0] + shfl_xor_f64(v[0], 32);
}

// ---------------------------------------------------------------- wave reduction of the rows on the matrix cores
// The 29 sums H = sum A A^T, g = sum A b, sum b^2, sum r^2 are the entries of the 8x8 Gram matrix X^T X of the wave's
// 64 rows X[p] = [A0..A5, b, r] - a GEMM, so it runs as v_mfma_f64_16x16x4_f64 instead of 29 products per lane and a
// 31-value cross-lane reduction on the VALU (which is what bounds this kernel).  Two points share one 16-wide operand row
// (even point in columns 0-7, odd point in 8-15): the 16x16 result then holds the even points' Gram matrix in its upper
// left 8x8 block and the odd points' in the lower right one, K = 32 instead of 64 and no zero padding; the two cross
// blocks are discarded.  Staging: every lane stores its row to the wave's private LDS area (stride 9 doubles: the 64-bit
// stores and the operand loads are bank-conflict free), the operand of step kb is ONE ds_read_b64 per lane and serves as
// both A and B.  Operand layout: A[i][k] / B[k][j] in lane i + 16 k; result D[row][col]: col = lane & 15, row = (lane >> 4)
// + 4 reg.  Returns u0 = M[lane >> 4][lane & 7], u1 = M[(lane >> 4) + 4][lane & 7] of the wave's Gram matrix M (all
// lanes).  The summation order is fixed by the instruction sequence: deterministic.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wave_gram_mfma(const double (&row)[8], double *stage, int lane, double &u0, double &u1) {
#pragma unroll
    for (int c = 0; c < 8; ++c) stage[lane * kRowStride + c] = row[c];
    const int c16 = lane & 15, k = lane >> 4;
    const double *op = stage + (2 * k + (c16 >> 3)) * kRowStride + (c16 & 7);
    mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const double x = op[kb * 8 * kRowStride];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    const bool even = c16 < 8;
    u0 = even ? acc[0] : acc[2];
    u1 = even ? acc[1] : acc[3];
    u0 = dpp_add<0x128, 0xF>(u0);      // row_ror:8 - the lane holding the other half's block entry
    u1 = dpp_add<0x128, 0xF>(u1);
}
// slot of the 32-double partial row -> entry (a, b) of the Gram matrix: 21 H (upper triangle, row-major), 6 g = (j, 6),
// sum r^2 = (7, 7), sum b^2 = (6, 6)
__device__ __forceinline__ int gram_entry_of_slot(int slot) {
    constexpr uint8_t tab[29] = {0 * 8 + 0, 0 * 8 + 1, 0 * 8 + 2, 0 * 8 + 3, 0 * 8 + 4, 0 * 8 + 5, 1 * 8 + 1, 1 * 8 + 2, 1 * 8 + 3, 1 * 8 + 4,
                                 1 * 8 + 5, 2 * 8 + 2, 2 * 8 + 3, 2 * 8 + 4, 2 * 8 + 5, 3 * 8 + 3, 3 * 8 + 4, 3 * 8 + 5, 4 * 8 + 4, 4 * 8 + 5,
                                 5 * 8 + 5, 0 * 8 + 6, 1 * 8 + 6, 2 * 8 + 6, 3 * 8 + 6, 4 * 8 + 6, 5 * 8 + 6, 7 * 8 + 7, 6 * 8 + 6};
    return tab[slot];
}

// the searches of k_lin sweep the occupied rows of their ball (search.hpp knn_shells<.., true>); the ring walk is what dcreg_knn runs
constexpr bool kLinSweep = true;

// XCD-aware block remap: hardware places block b on XCD b % 8 (as that XCD's (b / 8)-th block).
//   chunk == 0: every XCD gets ONE contiguous run of query blocks, so spatially adjacent (Hilbert-ordered) queries share
//               that XCD's L2 - best when the work per query is uniform;
//   chunk == c: runs of c consecutive query blocks are dealt round-robin to the XCDs - keeps c * 256 neighbouring queries
//               on one L2 but spreads spatially clustered heavy queries (a misaligned corridor end) over all eight XCDs.
// Bijective for any n.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n, uint32_t chunk) {
    const uint32_t nx = 8u;
    const uint32_t xcd = b % nx, k = b / nx;
    if (chunk == 0) {
        const uint32_t q = n / nx, r = n % nx;
        // XCD x owns q + (x < r) blocks, laid out back to back
        const uint32_t base = xcd * q + (xcd < r ? xcd : r);
        return base + k;
    }
    const uint32_t span = nx * chunk;                 // query blocks per round
    const uint32_t full = (n / span) * span;          // complete rounds; the ragged tail maps to itself
    if (b >= full) return b;
    return (k / chunk) * span + xcd * chunk + (k % chunk);
}

// ---------------------------------------------------------------- the linearisation kernel
// One linearisation (steps 1-5 of an iteration, icp_test_runner.cpp:1714-1915) of a pose is ONE launch of k_lin, one thread per source
// point: transform, then as little as the point's state (search.hpp kStateRows) allows -
//   level 1  no certificate, or the point has left its certificate's radius: exact 6-NN search (bounded by the old neighbours when
//            there are some), new certificate;
//   level 2  the set certificate holds but the neighbour ORDER may have changed: gather the known neighbours, order them at the new
//            position, radius gate, plane fit, neighbour-only gates, new fit certificate;
//   level 3  the fit certificate holds too: the stored plane;
// and for every point alike residual, weight, weight gate, row from the plane; reduction.  A wave runs a level only if one of its 64
// points needs it: on a settled trajectory almost none needs more than level 3, i.e. 72 B of state and point per source point and
// no gather at all.  Every level reproduces bitwise what a fresh search + fit at this pose gives, the rows are built by the same code
// in the same lane order, and the block / chunk / pose sums take them in index order: the sums do not depend on the state's history.
// (Measured alternative, round 3: certificates tested by a lean search-free kernel that puts the points to search on compacted work
// lists for two follow-up kernels - 38 us instead of 43.5 us for the settled 1 M launch, but the follow-up kernels cost 10 us of
// stream time even when the lists are empty and a whole-run bench of 9.8 k instead of 11.4 k it/s: profiles/r03_ablation.md.)
// MODE 0: reduction only.  MODE 1: also dump per-point results (parity tests).
struct DebugDev {
    int32_t *nn_idx; float *nn_d2; uint8_t *flag; double *normal; double *r; double *s;
    uint32_t *stats;   // per point: candidates evaluated | outermost shell << 16
    unsigned long long *stamps;   // MODE 2: per wave (query block x 4 + wave) eight words - shader-clock stamps at the phase boundaries of
                                  // k_lin (start, state loaded + tests, search done, fit done, row done, reduction done) and the wave's
                                  // searched / refitted lane counts (scripts/wave_phases.py)
};

// Single-pose launches finish inside the kernel (no second launch): blocks are grouped in chunks of kChunk consecutive
// partial rows; the last block to finish in a chunk (ticket counter) sums that chunk's rows in a fixed order and writes
// the chunk row, with a check word that carries the launch's sequence number (publish_row), straight into pinned host-coherent
// memory.  The host spins on the rows and adds the few chunk rows in index order.  WHO sums is timing dependent, WHAT is summed in which order is not: the
// result is deterministic.  Batched launches (many poses, few blocks
// each) use k_finalize instead: a ticket per block costs more there than the extra launch (measured, profiles/r01_search_ablation.md
// addendum 5).
constexpr int kChunk = 64;
// counters that many blocks hit with atomics live one per 128-byte line: atomics on ONE line are served one after the other (~11 ns
// each), whatever word they address - 3907 ticket arrivals on two lines were 10 us of a 46 us kernel
constexpr int kCounterStride = 32;      // uint32 words
struct FinArgs {
    unsigned int *tickets;         // [n_chunks * kCounterStride], zero between launches (the last arrival resets its ticket)
    double *out;                   // pinned, device-mapped: [n_chunks][kSlots]; slot 31 = check word (publish_row)
    unsigned long long seq;
    unsigned int direct;           // 1: a launch of at most kChunk blocks - every block publishes its own row to out[block], the host adds
                                   //    them in block_sum_rows' order (context.hip linearize_end): no ticket, no cross-XCD read of the rows
    unsigned int chunks_per_pose;  // tickets and chunk rows of pose p start at p * chunks_per_pose (single-pose launches: pose 0).  Batched
                                   //    launches whose poses are ONE chunk each (<= kChunk blocks: the Monte-Carlo batches) finish in the kernel
                                   //    like single-pose ones - the chunk row IS the pose's result row - instead of a k_finalize launch
};

// Cross-block traffic of the tree uses agent-scope (sc1, write-through / L2-coherent) relaxed atomics plus an explicit
// s_waitcnt instead of __threadfence(): a release fence on gfx950 is a full L2 write-back (buffer_wbl2), measured at
// +10 us per launch when every block executes one.
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_system(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A result row in pinned host memory: 31 doubles + a check word, all 32 stored at once by the first 32 lanes of a wave.  The check word
// = launch number x K + sum of (bit pattern of slot i) x K_i (distinct odd multipliers: equal changes in two slots do not cancel): the
// host takes the row when the word fits the 31 values it sees next to it - whatever order the link delivered the stores in.  No
// "data, wait for the acknowledgement, then flag": one PCIe round trip less on every result.  Called by threads 0 .. 31 of the block.
__host__ __device__ inline unsigned long long row_check_mult(int i) { return 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * i + 3); }
__device__ __forceinline__ void publish_row(double *orow, double value, unsigned long long seq) {
    const int l = threadIdx.x;                      // 0 .. 31; lane 31 carries no value
    const unsigned long long bits = l < 31 ? (unsigned long long)__double_as_longlong(value) : 0ull;
    unsigned long long c = l < 31 ? bits * row_check_mult(l) : seq * row_check_mult(31);
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)c, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(c >> 32), m);
        c += ((unsigned long long)hi << 32) | lo;
    }
    __hip_atomic_store((unsigned long long *)(orow + l), l < 31 ? bits : c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum of `count` (<= kChunk) rows of kSlots doubles, fixed order: lane group g (of G = kLinBlock / 32) adds rows g, g + G, ... then the
// G group sums are added in order.  All kLinBlock threads call; threads < 31 return the total of their slot.
__device__ __forceinline__ double block_sum_rows(const double *rows, uint32_t count, double (*sm)[kSlots]) {
    constexpr int G = kLinBlock / 32;               // lane groups of 32: one slot each
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double v[kChunk / G];
#pragma unroll
    for (int u = 0; u < kChunk / G; ++u) {
        const uint32_t r = (uint32_t)grp + (uint32_t)G * u;
        v[u] = r < count ? ld_agent(rows + (size_t)r * kSlots + j) : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < kChunk / G; ++u) t += v[u];
    __syncthreads();
    sm[grp][j] = t;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < 31) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) tot += sm[gi][threadIdx.x];
    }
    return tot;
}

// k_lin's one-wave instantiation leaves a row per 64-point TILE (four per query block) and no ticket: k_sum_tiles, queued behind it, adds
// them - one block per chunk of kChunk query blocks.  Block row r = ((0 + tile 4r) + tile 4r+1) + ... exactly as block_publish adds the
// four waves' values, then the rows as block_sum_rows adds them: the same additions in the same order, bit for bit.  (In the kernel
// itself the sum would be the last arrival's, ONE wave with a kilobyte per lane to load: four dependent rounds of loads, 3 us each -
// measured; here it is one round behind a kernel boundary.)
// Grid: (chunks of a pose, poses) - batched launches of one-chunk poses (the Monte-Carlo batches) get their pose rows this way.
static __global__ __launch_bounds__(kLinBlock) void k_sum_tiles(const double *__restrict__ tile_rows, uint32_t n_blocks_x, double *__restrict__ out,
                                                                unsigned long long seq, const uint32_t *__restrict__ abort_flag) {
    if (abort_flag && *abort_flag != 0u) return;       // the launch was called off: k_lin wrote no rows, nothing is published
    __shared__ double sm[kLinBlock / 32][kSlots];
    constexpr int G = kLinBlock / 32, T = kLinBlock / 64;
    const uint32_t chunk = blockIdx.x;
    const uint32_t count = min((uint32_t)kChunk, n_blocks_x - chunk * kChunk);
    const double *rows = tile_rows + ((size_t)blockIdx.y * n_blocks_x + (size_t)chunk * kChunk) * T * kSlots;
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double x[kChunk / G][T];
#pragma unroll
    for (int u = 0; u < kChunk / G; ++u) {
        const uint32_t r = (uint32_t)grp + (uint32_t)G * u;
#pragma unroll
        for (int w = 0; w < T; ++w) x[u][w] = r < count ? rows[((size_t)r * T + w) * kSlots + j] : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < kChunk / G; ++u) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < T; ++w) v += x[u][w];
        t += v;
    }
    sm[grp][j] = t;
    __syncthreads();
    if (threadIdx.x < 32) {
        double tot = 0.0;
        if (threadIdx.x < 31) {
#pragma unroll
            for (int gi = 0; gi < G; ++gi) tot += sm[gi][threadIdx.x];
        }
        publish_row(out + ((size_t)blockIdx.y * gridDim.x + chunk) * kSlots, tot, seq);
    }
}

// ---------------------------------------------------------------- the gate of a pipelined launch
// An ICP loop alternates one linearisation and a 3 us host step, and every launch costs ~4 us of host time plus ~2 us until the
// device starts: idle time for a device that has nothing else queued.  A GATED linearisation is queued while its predecessor still
// runs, before its pose exists: k_gate (one wave) sits in the stream in front of it and polls a small record in pinned host
// memory; when the host publishes the pose there, the gate copies it into the device-resident PoseArg the linearisation reads,
// and retires.  The host can also call the launch off (abort bit): the kernels behind the gate then return at once.
// A gate that waits longer than kGateTimeoutTicks (wall clock, 100 MHz; far longer than the host ever waits for a result) aborts by
// itself, so a vanished host cannot leave the queue spinning.
// The gate record: 14 words of pinned, host-coherent memory.  w[0] = (launch number << 1) | abort, w[1..9] = R, w[10..12] = t (bit
// patterns of doubles), w[13] = kGateSalt ^ w[0] ^ ... ^ w[12].  The gate reads all words with ONE load
// per lane and accepts them only if the number is the awaited one AND the checksum holds: the loads of one poll may be served in
// any order relative to the host's stores, a torn snapshot fails the checksum and is simply polled again - one PCIe round trip
// between "pose published" and "pose on the device", whatever the read granularity of the link.
constexpr int kGateWords = 14;
constexpr unsigned long long kGateTimeoutTicks = 12000000000ull;     // 120 s
struct alignas(128) GateHost { unsigned long long w[32]; };
constexpr unsigned long long kGateSalt = 0x9E3779B97F4A7C15ull;
static __global__ __launch_bounds__(64) void k_gate(const GateHost *__restrict__ hg, unsigned long long want, PoseArg *__restrict__ dst,
                                                   uint32_t fresh, uint32_t *__restrict__ abort_flag) {
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned long long v = 0, seq = 0;
    for (;;) {
        v = lane < kGateWords ? __hip_atomic_load(&hg->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
        // x = xor of all the record's words (lanes beyond it contribute 0): the salt when the record is whole
        unsigned long long x = v;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m);
            x ^= ((unsigned long long)hi << 32) | lo;
        }
        const uint32_t slo = __builtin_amdgcn_readfirstlane((uint32_t)v), shi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        seq = ((unsigned long long)shi << 32) | slo;                        // lane 0's word
        const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)x), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
        const bool whole = (((unsigned long long)xhi << 32) | xlo) == kGateSalt;
        if ((seq >> 1) == want && whole) break;
        // a whole record with a LATER number: the host has moved on, i.e. it called this launch off before this gate ever ran
        if ((seq >> 1) > want && whole) { seq = (want << 1) | 1ull; break; }
        if (wall_clock64() - t0 > kGateTimeoutTicks) { seq = (want << 1) | 1ull; break; }     // nobody opens - give up
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane >= 1 && lane <= 12) {
        const double d = __longlong_as_double((long long)v);
        if (lane <= 9) dst->R[lane - 1] = d; else dst->t[lane - 10] = d;
    }
    if (lane == 0) { dst->state = 0; dst->fresh = fresh; *abort_flag = (uint32_t)(seq & 1ull); }
}

// ---- the gate INSIDE the first kernel of a small launch (round 5; round 4 measured both forms: profiles/r04_gate_in_kernel.txt).  A
// launch of a few dozen blocks lasts 4-10 us, and the boundary between the one-wave gate kernel and it is 1.5 us of that; a launch of
// thousands of blocks pays 2 us for the instructions below in every wave instead.  So: launches the device holds at once (at most
// kChunk query blocks: the fixture, a LiDAR frame) wait for their pose themselves - every wave requests the DEVICE copy of the gate
// record together with its first loads (a wave dispatched after the pose has reached the device finds it whole and its own: no extra
// round trip), the first wave of the launch polls the host record and fills the device copy (and, for the kernels queued behind this
// one, the device-resident PoseArg and abort word k_gate would have filled) - and large launches keep k_gate.
struct alignas(128) GateDev { unsigned long long w[16]; };   // w[0] = (launch number << 1) | abort, w[1..12] = R, t, w[13] = checksum (as GateHost)
struct GateArgs {
    const GateHost *host; GateDev *dev;
    unsigned long long want;           // this launch's number
    PoseArg *pose_out; uint32_t *abort_out;     // what the kernels behind this one read (null: nothing is queued behind the gated kernel)
    uint32_t fresh;
};
// one lane-parallel read of a 14-word gate record (host or device copy): the words (lane l < 14 holds w[l]) and whether they form a
// whole record (checksum) - uniform results in `seq` (w[0]) and the return value.  All 64 lanes of the wave call.
template <int SCOPE>
__device__ __forceinline__ bool gate_read(const unsigned long long *rec, unsigned long long &v, unsigned long long &seq) {
    const int lane = threadIdx.x & 63;
    v = lane < kGateWords ? __hip_atomic_load(rec + lane, __ATOMIC_RELAXED, SCOPE) : 0ull;
    unsigned long long x = v;          // xor of all the record's words (lanes beyond it contribute 0): the salt when the record is whole
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m);
        x ^= ((unsigned long long)hi << 32) | lo;
    }
    const uint32_t slo = __builtin_amdgcn_readfirstlane((uint32_t)v), shi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    seq = ((unsigned long long)shi << 32) | slo;
    const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)x), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return (((unsigned long long)xhi << 32) | xlo) == kGateSalt;
}
// Called by every wave of a launch gated in the kernel, convergently, once its pose-independent loads are in flight; v / seq / whole = the
// wave's read of the DEVICE record (gate_read<agent>, requested with those loads).  Returns false when the launch was called off.
__device__ __forceinline__ bool gate_wait(const GateArgs &gt, unsigned long long v, unsigned long long seq, bool whole, PoseArg &P) {
    const int lane = threadIdx.x & 63;
    GateDev *gd = gt.dev;
    if (!(whole && (seq >> 1) == gt.want)) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
            // the polling wave: host record -> device record
            const unsigned long long t0 = wall_clock64();
            unsigned long long hv = 0, hseq = 0;
            bool call_off = false;
            for (;;) {
                const bool hw = gate_read<__HIP_MEMORY_SCOPE_SYSTEM>(gt.host->w, hv, hseq);
                if ((hseq >> 1) == gt.want && hw) { call_off = (hseq & 1ull) != 0ull; break; }
                // a whole record with a LATER number: the host has moved on, i.e. it called this launch off before this wave ever ran
                if (((hseq >> 1) > gt.want && hw) || wall_clock64() - t0 > kGateTimeoutTicks) { call_off = true; break; }     // (or nobody opens)
                __builtin_amdgcn_s_sleep(2);
            }
            if (call_off) {             // an abort record of this launch's number (the pose words are whatever they were: nobody reads them)
                hv = lane == 0 ? ((gt.want << 1) | 1ull) : (lane < kGateWords - 1 ? hv : 0ull);
                unsigned long long x = lane < kGateWords - 1 ? hv : 0ull;
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m);
                    x ^= ((unsigned long long)hi << 32) | lo;
                }
                if (lane == kGateWords - 1) hv = x ^ kGateSalt;
            }
            if (gt.pose_out) {          // what k_gate would have left for the kernels queued behind this one
                if (lane >= 1 && lane <= 12) {
                    const double d = __longlong_as_double((long long)hv);
                    if (lane <= 9) gt.pose_out->R[lane - 1] = d; else gt.pose_out->t[lane - 10] = d;
                }
                if (lane == 0) { gt.pose_out->state = 0; gt.pose_out->fresh = gt.fresh; *gt.abort_out = call_off ? 1u : 0u; }
            }
            if (lane < kGateWords) __hip_atomic_store(&gd->w[lane], hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every wave that came too early (the polling one included): until the device record is whole and this launch's
        for (;;) {
            whole = gate_read<__HIP_MEMORY_SCOPE_AGENT>(gd->w, v, seq);
            if (whole && (seq >> 1) == gt.want) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if ((seq & 1ull) != 0ull) return false;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, k + 1), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), k + 1);
        const double d = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        if (k < 9) P.R[k] = d; else P.t[k - 9] = d;
    }
    return true;
}

// ---- the common tail of a block: the wave's rows -> its Gram matrix and counts in LDS (the wave's LDS staging area must be free) ...
// gm: where this wave's 8x8 Gram matrix goes (64 doubles; may be the head of its own staging area: the operand reads are over by then)
__device__ __forceinline__ void wave_rows_to_lds(const double (&row)[8], uint8_t flag, double *stage, double *gm, double (*cnt)[2],
                                                 double extra0 = 0.0, double extra1 = 0.0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double u0, u1;
    wave_gram_mfma(row, stage, lane, u0, u1);
    // (the wave's Gram matrix, M[a][b] at a * 8 + b)
    if ((lane & 15) < 8) {
        gm[(lane >> 4) * 8 + (lane & 7)] = u0;
        gm[32 + (lane >> 4) * 8 + (lane & 7)] = u1;
    }
    const unsigned long long eff = __builtin_amdgcn_ballot_w64(flag == 1), inr = __builtin_amdgcn_ballot_w64(flag != 0);
    // (extra0 / extra1: the wave's searched / refitted lanes x LinArgs::count_scale, riding above the counts)
    if (lane == 0) { cnt[wave][0] = (double)__builtin_popcountll(eff) + extra0; cnt[wave][1] = (double)__builtin_popcountll(inr) + extra1; }
}
// ... and, after a block barrier: the block partial (fixed order, no float atomics) and, for single-pose launches, the arrival at
// the chunk's ticket: the last of its blocks sums the chunk and publishes the row to the host.  All 256 threads call.
// gm0 / gm_stride: the waves' Gram matrices (wave w at gm0 + w * gm_stride); red: kLinBlock / 32 x kSlots doubles of scratch for the chunk
// sum (may overlap the Gram matrices: they are consumed before)
template <bool FUSED>
__device__ __forceinline__ void block_publish(const double *gm0, int gm_stride, double (*red)[kSlots], double (*cnt)[2], int *s_role, double *my_rows, uint32_t vb,
                                              uint32_t n_blocks_x, const FinArgs &fin, uint32_t pose_id) {
    if (threadIdx.x < kSlots) {
        double t = 0.0;
        if (threadIdx.x < 29) {
            const int e = gram_entry_of_slot(threadIdx.x);
#pragma unroll
            for (int w = 0; w < kLinBlock / 64; ++w) t += gm0[w * gm_stride + e];
        } else if (threadIdx.x < 31) {
#pragma unroll
            for (int w = 0; w < kLinBlock / 64; ++w) t += cnt[w][threadIdx.x - 29];
        }
        if (FUSED && fin.direct) {
            publish_row(fin.out + (size_t)vb * kSlots, t, fin.seq);
        } else if (FUSED) {
            st_agent(my_rows + (size_t)vb * kSlots + threadIdx.x, t);
            wait_stores();                                 // the row is at the coherence point before the ticket is taken
        } else {
            my_rows[(size_t)vb * kSlots + threadIdx.x] = t;
        }
    }
    if (FUSED && !fin.direct) {
        const uint32_t chunk = vb / kChunk;
        const uint32_t csize = min((uint32_t)kChunk, n_blocks_x - chunk * kChunk);
        const size_t gchunk = (size_t)pose_id * fin.chunks_per_pose + chunk;         // this pose's chunk among all of the launch
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int prev = __hip_atomic_fetch_add(&fin.tickets[gchunk * kCounterStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_role = (prev == csize - 1) ? 1 : 0;
            if (prev == csize - 1) __hip_atomic_store(&fin.tickets[gchunk * kCounterStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*s_role == 1) {                                  // last block of this chunk: sum its rows, publish to the host
            const double t = block_sum_rows(my_rows + (size_t)chunk * kChunk * kSlots, csize, red);
            double *orow = fin.out + gchunk * kSlots;
            if (threadIdx.x < 32) publish_row(orow, t, fin.seq);
        }
    }
}

constexpr int kLinDepth = 2;                      // register sets of k_lin's candidate pipeline (search.hpp knn_search DEPTH)
// points per block of the advance pass (k_advance below): kAdvTile / kLinBlock query blocks of k_lin
constexpr int kAdvTile = 1536;
static_assert(kAdvTile % kLinBlock == 0 && kAdvTile <= 65536, "a tile is a whole number of query blocks; list entries are 16-bit offsets");

// ---------------------------------------------------------------- k_lin
// ONE: blocks of one wave (64 points; four of them cover a query block).  A four-wave block holds its wave slots until its slowest wave
// is through; one-wave blocks give a slot back when its wave ends - worth 6-9 % of a launch whose searches are long (queries a cell and
// more from the surface), and a loss where they are short (four times the blocks to dispatch: profiles/r06_ablation.md section 5).  Each
// block leaves the row of its TILE; k_sum_tiles, queued behind, adds the tile rows in the order block_publish and block_sum_rows add the
// waves' values and the block rows: the 31 sums are bitwise those of the four-wave launch.
template <int MODE, bool FUSED, bool FAST, bool GATE = false, bool ONE = false>
static __global__ __launch_bounds__(ONE ? kWave : kLinBlock, kLinOcc) void k_lin(const float4 *__restrict__ src, uint32_t n_src, GridDev g,
                                                          PoseArg pose1, const PoseArg *__restrict__ poses, LinArgs a,
                                                          double *__restrict__ partials, uint32_t n_blocks_x, FinArgs fin,
                                                          DebugDev dbg, const uint32_t *__restrict__ abort_flag, GateArgs gt) {
    if (abort_flag && *abort_flag != 0u) return;       // a gated launch the host called off (uniform: every block returns)
    static_assert(!ONE || (MODE == 0 && FUSED && !GATE), "the one-wave instantiation is the fused product launch behind k_gate");
    constexpr int kWavesHere = ONE ? 1 : kLinBlock / kWave;
    __shared__ double red[ONE ? 1 : kLinBlock / 32][kSlots];      // (ONE: the Gram matrix lives in the wave's RunList, behind the staging area)
    __shared__ double cnt[kWavesHere][2];
    __shared__ int s_role;
    __shared__ RunList runs[kWavesHere];
    static_assert(sizeof(RunList) >= sizeof(double) * (kWave * kRowStride + 64), "the one-wave block's reduction lives in its RunList");
    const double *const gm0 = &red[0][0];            // the waves' 8x8 Gram matrices (wave w at gm0 + 64 w); later the scratch of the chunk sum
    constexpr int gm_stride = 64;
    const int wave = threadIdx.x >> 6;
    const uint32_t pose_id = blockIdx.y;
    // (ONE: the blocks of the grid are tiles, four per query block; XCD runs and group order as for the query blocks they belong to)
    const uint32_t vbt = ONE ? xcd_remap(blockIdx.x, n_blocks_x * (kLinBlock / kWave), a.xcd_chunk * (kLinBlock / kWave)) : 0u;
    uint32_t vb = ONE ? vbt / (kLinBlock / kWave) : xcd_remap(blockIdx.x, n_blocks_x, a.xcd_chunk);
    const uint32_t tile = ONE ? vbt % (kLinBlock / kWave) : (uint32_t)wave;    // the wave's tile of its query block
    if (a.n_groups) {       // heavy groups first (k_group_cost).  The groups cover the END of the block range: the blocks no group covers
                            // are the first ones - dispatched first, whatever they cost
        const uint32_t lead = n_blocks_x - a.n_groups * a.group_blocks;
        if (vb >= lead) {
            const uint32_t slot = (vb - lead) / a.group_blocks;
            vb = lead + (uint32_t)a.group_order[slot] * a.group_blocks + (vb - lead) % a.group_blocks;
        }
    }
    const uint32_t i = ONE ? vb * kLinBlock + tile * kWave + threadIdx.x : vb * kLinBlock + threadIdx.x;
    auto stamp = [&](int k, unsigned long long v) {       // MODE 2 only (timing probe): one store by lane 0, nothing kept in registers
        if constexpr (MODE == 2) {
            if ((threadIdx.x & 63) == 0 && dbg.stamps) dbg.stamps[((size_t)vb * (kLinBlock / 64) + wave) * 8 + k] = v;
        }
    };
    // (the counter reads sit behind the same MODE test as the stores: s_memtime has side effects the compiler does not remove, and in
    //  the product instantiations it would also be a scheduling barrier)
    if constexpr (MODE == 2) stamp(0, __builtin_readcyclecounter());
    PoseArg P;
    if (poses) P = poses[pose_id]; else P = pose1;
    uint8_t flag = 0;
    const bool have_q = i < n_src;
    const bool keep = a.state != nullptr && P.state != kNoIdx;              // the pose owns a state
    const bool old = keep && P.fresh == 0u;                                 // ... that holds the results of earlier launches
    const bool CERT = old && a.use_cert != 0;                               // ... whose certificates are to be used (uniform)
    // the groups of this pose's state (search.hpp kStateRows: V0, V1, V2, W3 = what the fast path reads; X, Y = the six positions)
    uint32_t *const sbase = keep ? a.state + (size_t)P.state * kStateRows * a.state_stride : nullptr;
    const size_t ss = a.state_stride;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    uint4 *const SV0 = reinterpret_cast<uint4 *>(sbase + kStV0 * ss), *const SX = reinterpret_cast<uint4 *>(sbase + kStX * ss);
    dbl2 *const SV1 = reinterpret_cast<dbl2 *>(sbase + kStV1 * ss), *const SV2 = reinterpret_cast<dbl2 *>(sbase + kStV2 * ss);
    uint32_t *const SW3 = sbase + kStW3 * ss;
    uint2 *const SY = reinterpret_cast<uint2 *>(sbase + kStY * ss);
    const float4 s4 = have_q ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // what the fast path needs, in one batch of loads: certificate, reference position, fit word, plane (56 B + the 16 B of the point)
    uint32_t cert = kCertSearch, fitw = kFitNone, q0[3] = {0u, 0u, 0u};
    dbl2 p01 = {0.0, 0.0}, p23 = {0.0, 0.0};
    if (CERT && have_q) {
        const uint4 v0 = SV0[i];
        p01 = SV1[i]; p23 = SV2[i];
        cert = v0.x; fitw = v0.y; q0[0] = v0.z; q0[1] = v0.w; q0[2] = SW3[i];
    }
    if constexpr (GATE) {      // a small launch gated in the kernel: the pose arrives now, the loads above are in flight meanwhile (pose1: state / fresh)
        unsigned long long gate_v = 0ull, gate_seq = 0ull;
        const bool gate_whole = gate_read<__HIP_MEMORY_SCOPE_AGENT>(gt.dev->w, gate_v, gate_seq);
        if (!gate_wait(gt, gate_v, gate_seq, gate_whole, P)) return;       // called off: every wave returns here, before any barrier
    }
    float qx, qy, qz;
    body_to_global(P, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
    const float q0x = __uint_as_float(q0[0]), q0y = __uint_as_float(q0[1]), q0z = __uint_as_float(q0[2]);
    // three levels: (1) the set certificate fails -> search; (2) it holds but the fit certificate does not -> gather the known
    // neighbours, order, fit; (3) both hold -> the stored plane.  OUT certificates: nothing to do at all.
    const bool need = have_q && !(CERT && cert_holds(cert, q0x, q0y, q0z, qx, qy, qz));
    bool refit = have_q && !need && !cert_is_out(cert) && !fit_holds(fitw, q0x, q0y, q0z, qx, qy, qz);
    uint32_t stats = 0;
    uint32_t w_search = 0, w_refit = 0;             // lanes of this wave that were searched / only refitted (uniform)
    uint32_t adv_s = 0, adv_r = 0;                  // ... and what the advance pass in front of this launch did for this block's points
    if (a.adv_counts && tile == 0u) {                // (taken and zeroed again: the passes only ever add to zeroes)
        adv_s = a.adv_counts[(size_t)vb * kCounterStride]; adv_r = a.adv_counts[(size_t)vb * kCounterStride + 1];
        if ((threadIdx.x & 63u) == 0u) { a.adv_counts[(size_t)vb * kCounterStride] = 0u; a.adv_counts[(size_t)vb * kCounterStride + 1] = 0u; }
    }
    KnnResult<5> nn;
    Fit fit;
    uint8_t gate = 255;                             // 0: plane usable; 2 / 3: neighbour-only gate failed; 255: radius gate failed / OUT
    const bool level3 = have_q && !need && !refit && !cert_is_out(cert);
    auto stored_plane = [&](const dbl2 &a01, const dbl2 &a23) {
        gate = (uint8_t)(fitw & 3u);
        fit.plane[0] = a01.x; fit.plane[1] = a01.y; fit.plane[2] = a23.x; fit.plane[3] = a23.y;
    };
    if constexpr (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(1, __builtin_readcyclecounter()); }
    if (!wave_any(need || refit)) {
        // the whole wave is at level 3 (or OUT): the plane words loaded up front are all it needs.  (They are consumed HERE and not
        // below: kept alive across the search they would cost a dozen registers at its peak, i.e. scratch spills; the waves that do
        // search load them a second time afterwards - from the cache.)
        if (level3) stored_plane(p01, p23);
    } else {
        uint32_t pos6[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) pos6[j] = kNoIdx;
        // the index the state WRITES of this branch are addressed with: the same number behind a compiler barrier, so that their
        // addresses are formed where they are used (kept from the top of the kernel, five 64-bit addresses ride through the search)
        uint32_t iw = i;
        asm volatile("" : "+v"(iw));
        const unsigned long long need_mask = __builtin_amdgcn_ballot_w64(need);
        w_search = (uint32_t)__builtin_popcountll(need_mask);
        if (need_mask != 0ull) {
            if (a.search_count && (threadIdx.x & 63) == 0)       // 64 counters on lines of their own (kCounterStride): see there
                atomicAdd(a.search_count + (size_t)(blockIdx.x & 63u) * (kCounterStride / 2), (unsigned long long)__builtin_popcountll(need_mask));
            const bool warm = old && a.warm != 0;
            if (warm && need) {
                const uint4 x = SX[i];
                const uint2 y = SY[i];
                pos6[0] = x.x; pos6[1] = x.y; pos6[2] = x.z; pos6[3] = x.w; pos6[4] = y.x; pos6[5] = y.y;
            }
            bool by_team = false;                   // uniform: the team served every lane that had to be searched
            // a wave with a few lanes to search, each of them near its old neighbours: the 64 lanes serve one query at a time
            if (warm && w_search <= (uint32_t)a.team_max) {
                float tb = 0.f;
                bool tight = false;
                if (need && pos6[5] != kNoIdx) tb = team_bound(g, a, pos6, qx, qy, qz, tight);
                if (!wave_any(need && !tight)) {
                    uint32_t tpos[6], tcert;
                    by_team = team_search6(g, runs[wave].team, a, need_mask, qx, qy, qz, tb, tpos, tcert) == need_mask;
                    if (a.search_count && by_team && (threadIdx.x & 63) == 0)      // (statistics: the word next to the search counter)
                        atomicAdd(a.search_count + (size_t)(blockIdx.x & 63u) * (kCounterStride / 2) + 1, (unsigned long long)w_search);
                    if (by_team && need) {
                        cert = tcert;
#pragma unroll
                        for (int j = 0; j < 6; ++j) pos6[j] = tpos[j];
                    }
                }
            }
            // (all or nothing: a team result kept alive across the lock-step search would cost that search registers)
            if (!by_team) {
                Set6 s6;
                uint32_t c2;
                lin_search6<kLinSweep, kLinDepth>(g, runs[wave], a, need, warm, pos6, qx, qy, qz, s6, c2);
                if (need) {
                    cert = c2;
#pragma unroll
                    for (int j = 0; j < 6; ++j) pos6[j] = s6.pos[j];
                    stats = (s6.n_eval & 0xFFFFu) | ((s6.n_shell & 0x7FFFu) << 16);
                }
            }
            if (need && keep) {
                SX[iw] = make_uint4(pos6[0], pos6[1], pos6[2], pos6[3]);
                SY[iw] = make_uint2(pos6[4], pos6[5]);
            }
        }
        if constexpr (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(2, __builtin_readcyclecounter()); stamp(6, (unsigned long long)w_search); }
        // level 2 for the lanes that were searched and the lanes whose order may have changed
        const bool set = have_q && !cert_is_out(cert);
        const bool fitnow = set && (need || refit);
        w_refit = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(fitnow && !need));
        if (wave_any(fitnow)) {
            // a SET6 certificate says nothing about which five of the six are nearest: the fit certificate must then cover the gap
            // between the 5th and the 6th itself, i.e. the fit works on all six
            const bool use6 = fitnow && cert_is_set6(cert);
            const bool six = wave_any(use6);
            const bool presorted = !wave_any(fitnow && !need);     // every lane that fits was searched just now: its six are in order
            if (fitnow && !need) {
                const uint4 x = SX[iw];
                const uint2 y = SY[iw];
                pos6[0] = x.x; pos6[1] = x.y; pos6[2] = x.z; pos6[3] = x.w; pos6[4] = y.x; pos6[5] = y.y;
            }
            if (!use6) pos6[5] = kNoIdx;
            if (fitnow) {
                const uint8_t in_r = fit_from_set<FAST>(g, a, qx, qy, qz, pos6, six, nn, fit, presorted);
                gate = in_r ? (uint8_t)(fit.word & 3u) : (uint8_t)255;
                if (keep) {                         // the new reference position, the certificate as seen from there, the fit
                    if (!need) {                    // (the old reference position is read again: three registers less across the search)
                        const uint4 o0 = SV0[iw];
                        const uint32_t o1 = SW3[iw];
                        cert = cert_rebased(cert, __uint_as_float(o0.z), __uint_as_float(o0.w), __uint_as_float(o1), qx, qy, qz);
                    }
                    SV0[iw] = make_uint4(cert, fit.word, __float_as_uint(qx), __float_as_uint(qy));
                    SV1[iw] = dbl2{fit.plane[0], fit.plane[1]};
                    SV2[iw] = dbl2{fit.plane[2], fit.plane[3]};
                    SW3[iw] = __float_as_uint(qz);
                }
            }
        }
        if constexpr (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3, __builtin_readcyclecounter()); stamp(7, (unsigned long long)w_refit); }
        if (level3) {                               // the lanes of this wave that needed neither: their stored plane, loaded again
            const dbl2 b01 = SV1[iw], b23 = SV2[iw];
            fitw = reinterpret_cast<const uint32_t *>(SV0 + iw)[1];
            stored_plane(b01, b23);
        }
        if (!set && need && keep) {                 // searched and found OUT: certificate and reference position, no fit
            SV0[iw] = make_uint4(cert, kFitNone, __float_as_uint(qx), __float_as_uint(qy));
            SW3[iw] = __float_as_uint(qz);
        }
    }
    double nrm[3] = {0.0, 0.0, 0.0}, r_pt = 0.0, s_pt = 0.0;
    double row[8];                                  // (zero unless the point is effective; initialised here, not at the top: sixteen
#pragma unroll                                      //  registers of zeros are not carried through the search)
    for (int k = 0; k < 8; ++k) row[k] = 0.0;
    if (have_q) {
        if (gate == 0) flag = row_of_plane<FAST>(P, a, s4, qx, qy, qz, fit.plane, row, nrm, r_pt, s_pt);
        else flag = gate == 255 ? (uint8_t)0 : gate;
    }
    if constexpr (MODE == 2) stamp(4, __builtin_readcyclecounter());
    if (MODE == 1 && have_q) {                      // (debug launches search and fit every point: nn is this launch's list)
        const uint32_t oi = __float_as_uint(s4.w);
#pragma unroll
        for (int j = 0; j < 5; ++j) {        // the neighbour list is defined for queries that pass the radius gate (:1726)
            if (dbg.nn_idx) dbg.nn_idx[5 * (size_t)oi + j] = flag != 0 ? (int32_t)nn.idx[j] : -1;
            if (dbg.nn_d2) dbg.nn_d2[5 * (size_t)oi + j] = flag != 0 ? nn.d2[j] : __builtin_inff();
        }
        if (flag == 1 || flag == 4) {
            if (dbg.normal) { dbg.normal[3 * (size_t)oi] = nrm[0]; dbg.normal[3 * (size_t)oi + 1] = nrm[1]; dbg.normal[3 * (size_t)oi + 2] = nrm[2]; }
            if (dbg.r) dbg.r[oi] = r_pt;
            if (dbg.s) dbg.s[oi] = s_pt;
        }
        if (dbg.flag) dbg.flag[oi] = flag;
        if (dbg.stats) dbg.stats[oi] = stats;
    }
    // (the wave's RunList is free now: it stages the rows)
    if constexpr (ONE) {
        double *const gm = runs[0].stage + kWave * kRowStride;                              // behind the staging area
        wave_rows_to_lds(row, flag, runs[0].stage, gm, cnt, a.count_scale * (double)(w_search + adv_s), a.count_scale * (double)(w_refit + adv_r));
        __syncthreads();
        constexpr uint32_t T = kLinBlock / kWave;
        if (threadIdx.x < kSlots) {                        // the tile's row; k_sum_tiles (behind this kernel) adds the rows
            double t = 0.0;
            if (threadIdx.x < 29) t = gm[gram_entry_of_slot(threadIdx.x)];
            else if (threadIdx.x < 31) t = cnt[0][threadIdx.x - 29];
            partials[(((size_t)pose_id * n_blocks_x + vb) * T + tile) * kSlots + threadIdx.x] = t;
        }
    } else {
    wave_rows_to_lds(row, flag, runs[wave].stage, &red[0][0] + wave * gm_stride, cnt, a.count_scale * (double)(w_search + adv_s),
                     a.count_scale * (double)(w_refit + adv_r));
    if constexpr (MODE == 2) stamp(5, __builtin_readcyclecounter());
    __syncthreads();
    block_publish<FUSED>(gm0, gm_stride, red, cnt, &s_role, partials + (size_t)pose_id * n_blocks_x * kSlots, vb, n_blocks_x, fin, pose_id);
    }
}

// ---------------------------------------------------------------- the advance pass: searches and refits in dense waves
// In the launches of a converging run's TRANSITION (a few per cent of the points have left their certificates) k_lin pays the full
// price of a search for a few lanes: a block stays as long as ONE lock-step search lasts whichever of its waves runs it, and nearly
// every block has a lane that needs one (profiles/r04_ablation.md sections 1, 18).  k_advance takes those lanes out of k_lin: it runs
// IN FRONT of it on the same stream, tests the certificates of a TILE of kAdvTile points per block (36 B per point: the point and the
// two state groups the tests read), collects the points that fail in a per-block list - searches from the front, refit-only points
// from the back - and works the list off in DENSE waves of 64: search (lin_search6, bounded by the old neighbours), plane fit, new
// certificate, fit word, plane and reference position written to the state exactly as k_lin's levels 1 and 2 write them.  k_lin then
// finds every certificate fresh (its reference position is the point's position at this very pose) and every wave takes the
// stored-plane path.  Nothing else changes hands: the rows, their order and the 31 sums are k_lin's own - what the state holds never
// changes a result (history independence), so the sums are bitwise those of a launch without the pass.  The host decides per launch
// (context.hip: the fraction of points the last completed launch searched; clouds whose query blocks exceed what the device holds).
// A point whose new certificate has no slack at all (exact distance ties) is searched again by k_lin - correct, merely slower.
constexpr int kAdvDepth = 4;                      // trips in flight in the pass's searches (search.hpp knn_search DEPTH); > 2: the kernel takes
                                                  // the registers of two waves per SIMD - its dense waves are few and each a chain of round trips
template <bool FAST>
static __global__ __launch_bounds__(kLinBlock, (kAdvDepth > 2 ? 2 : kLinOcc)) void k_advance(const float4 *__restrict__ src, uint32_t n_src, GridDev g, PoseArg pose1,
                                                                              const PoseArg *__restrict__ poses, LinArgs a,
                                                                              uint32_t *__restrict__ counts, const uint32_t *__restrict__ abort_flag) {
    if (abort_flag && *abort_flag != 0u) return;       // a gated launch the host called off
    __shared__ RunList runs[kLinBlock / kWave];
    __shared__ uint16_t list[kAdvTile];              // offsets into the tile: points to search from the front, refit-only points from the back
    __shared__ uint32_t n_list[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PoseArg P;
    if (poses) P = poses[0]; else P = pose1;
    const size_t ss = a.state_stride;
    uint32_t *const sbase = a.state + (size_t)P.state * kStateRows * ss;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    uint4 *const SV0 = reinterpret_cast<uint4 *>(sbase + kStV0 * ss), *const SX = reinterpret_cast<uint4 *>(sbase + kStX * ss);
    dbl2 *const SV1 = reinterpret_cast<dbl2 *>(sbase + kStV1 * ss), *const SV2 = reinterpret_cast<dbl2 *>(sbase + kStV2 * ss);
    uint32_t *const SW3 = sbase + kStW3 * ss;
    uint2 *const SY = reinterpret_cast<uint2 *>(sbase + kStY * ss);
    if (threadIdx.x < 2) n_list[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kAdvTile;
    // ---- the tests: every thread takes kAdvTile / kLinBlock points, all their loads in flight together
    constexpr int U = kAdvTile / kLinBlock;
    {
        float4 s4[U];
        uint4 v0[U];
        uint32_t w3[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = base + (uint32_t)(u * kLinBlock) + threadIdx.x;
            const bool have = i < n_src;
            s4[u] = have ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            v0[u] = have ? SV0[i] : make_uint4(0x80000000u | 0x7F000000u, kFitNone, 0u, 0u);      // (padding: an OUT certificate with a huge slack)
            w3[u] = have ? SW3[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = base + (uint32_t)(u * kLinBlock) + threadIdx.x;
            const bool have = i < n_src;
            float qx, qy, qz;
            body_to_global(P, (double)s4[u].x, (double)s4[u].y, (double)s4[u].z, qx, qy, qz);
            const float q0x = __uint_as_float(v0[u].z), q0y = __uint_as_float(v0[u].w), q0z = __uint_as_float(w3[u]);
            const bool need = have && !cert_holds(v0[u].x, q0x, q0y, q0z, qx, qy, qz);
            const bool refit = have && !need && !cert_is_out(v0[u].x) && !fit_holds(v0[u].y, q0x, q0y, q0z, qx, qy, qz);
            const unsigned long long ms = __builtin_amdgcn_ballot_w64(need), mr = __builtin_amdgcn_ballot_w64(refit);
            uint32_t bs = 0, br = 0;
            if (lane == 0) {
                if (ms) bs = atomicAdd(&n_list[0], (uint32_t)__builtin_popcountll(ms));
                if (mr) br = atomicAdd(&n_list[1], (uint32_t)__builtin_popcountll(mr));
            }
            bs = (uint32_t)__builtin_amdgcn_readfirstlane((int)bs); br = (uint32_t)__builtin_amdgcn_readfirstlane((int)br);
            const uint32_t ps = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
            const uint32_t pr = __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u));
            const uint16_t off = (uint16_t)(u * kLinBlock + (int)threadIdx.x);
            if (need) list[bs + ps] = off;
            if (refit) list[(uint32_t)kAdvTile - 1u - (br + pr)] = off;
        }
    }
    __syncthreads();
    const uint32_t n_s = n_list[0], n_r = n_list[1];
    // (reported by the first query block of the tile: LinArgs::adv_counts)
    if (threadIdx.x == 0 && counts) { counts[(size_t)blockIdx.x * (kAdvTile / kLinBlock) * kCounterStride] = n_s; counts[(size_t)blockIdx.x * (kAdvTile / kLinBlock) * kCounterStride + 1] = n_r; }
    if (threadIdx.x == 0 && a.search_count && n_s)            // (option "count_searches": the pass's searches count like k_lin's)
        atomicAdd(a.search_count + (size_t)(blockIdx.x & 63u) * (kCounterStride / 2), (unsigned long long)n_s);
    // ---- the list, 64 entries per wave at a time: the searches (waves in turn), then the refit-only points
    const uint32_t c_s = (n_s + 63u) >> 6, c_r = (n_r + 63u) >> 6;
    auto chunk = [&](auto searching_c, uint32_t e, bool act) {
        constexpr bool searching = decltype(searching_c)::value;
        const uint32_t off = act ? (uint32_t)list[searching ? e : (uint32_t)kAdvTile - 1u - e] : 0u;
        const uint32_t i = base + off;                                      // (idle lanes read the tile's first point and write nothing)
        const float4 s4 = src[i];
        float qx, qy, qz;
        body_to_global(P, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
        uint32_t pos6[6];
        {
            const uint4 x = SX[i];
            const uint2 y = SY[i];
            pos6[0] = x.x; pos6[1] = x.y; pos6[2] = x.z; pos6[3] = x.w; pos6[4] = y.x; pos6[5] = y.y;
        }
        uint32_t cert;
        // (the state accesses behind the search are addressed with the same index behind a compiler barrier: their 64-bit addresses
        //  are formed where they are used instead of riding through the search - as in k_lin)
        uint32_t iw = i;
        if constexpr (searching) {
            Set6 s6;
            lin_search6<kLinSweep, kAdvDepth>(g, runs[wave], a, act, a.warm != 0, pos6, qx, qy, qz, s6, cert);
            asm volatile("" : "+v"(iw));
#pragma unroll
            for (int j = 0; j < 6; ++j) pos6[j] = s6.pos[j];
            if (act) {
                SX[iw] = make_uint4(pos6[0], pos6[1], pos6[2], pos6[3]);
                SY[iw] = make_uint2(pos6[4], pos6[5]);
            }
        } else {
            const uint4 o0 = SV0[i];
            const uint32_t o1 = SW3[i];
            cert = cert_rebased(o0.x, __uint_as_float(o0.z), __uint_as_float(o0.w), __uint_as_float(o1), qx, qy, qz);
        }
        const bool set = act && !cert_is_out(cert);
        if (wave_any(set)) {
            const bool use6 = set && cert_is_set6(cert);
            const bool six = wave_any(use6);
            if (!use6) pos6[5] = kNoIdx;
            if (set) {
                KnnResult<5> nn;
                Fit fit;
                (void)fit_from_set<FAST>(g, a, qx, qy, qz, pos6, six, nn, fit, searching);     // (searched just now: the six are in order)
                SV0[iw] = make_uint4(cert, fit.word, __float_as_uint(qx), __float_as_uint(qy));
                SV1[iw] = dbl2{fit.plane[0], fit.plane[1]};
                SV2[iw] = dbl2{fit.plane[2], fit.plane[3]};
                SW3[iw] = __float_as_uint(qz);
            }
        }
        if (act && !set) {                          // searched and found OUT: certificate and reference position, no fit
            SV0[iw] = make_uint4(cert, kFitNone, __float_as_uint(qx), __float_as_uint(qy));
            SW3[iw] = __float_as_uint(qz);
        }
    };
    for (uint32_t k = (uint32_t)wave; k < c_s; k += (uint32_t)(kLinBlock / kWave)) {
        const uint32_t e = k * 64u + (uint32_t)lane;
        chunk(std::true_type{}, e, e < n_s);
    }
    // (the refit chunks start with the wave after the one that took the last search chunk: the work of a tile spreads over its waves)
    for (uint32_t k = ((uint32_t)wave + (uint32_t)(kLinBlock / kWave) - c_s % (uint32_t)(kLinBlock / kWave)) % (uint32_t)(kLinBlock / kWave); k < c_r;
         k += (uint32_t)(kLinBlock / kWave)) {
        const uint32_t e = k * 64u + (uint32_t)lane;
        chunk(std::false_type{}, e, e < n_r);
    }
}

// ---------------------------------------------------------------- the advance pass for small frames: sixteen lanes per query
// One registration of a 1-10 k-point frame against a large map (the reference's own workload, icp_test_runner.cpp:442-461) is a few
// dozen query blocks on a 256-CU device: every wave alone on its SIMD, each running the lock-step search for its 64 queries - a chain
// of a dozen and more DEPENDENT memory round trips (0.5-0.8 us each: old neighbours, start-bound probe trip by trip, cell table,
// candidate trips, sweep layers, fit gather), which is what such a launch lasts.  k_advance_team turns the roles round, like
// team_search6 but for EVERY query of the launch that needs it, warm or loose: blocks of ONE wave take kTeamTile consecutive points,
// test their certificates (the point, the tested state groups and the six stored positions in one batch of loads), and serve the
// points that fail four at a time, SIXTEEN lanes per query, five round trips per query whatever its ball:
//   1 the six old neighbours gathered by six lanes (the warm bound: the largest of their distances, inflated like lin_search6's);
//   2 the occupied (y,z) rows of the query's ball: a ball of at most sixteen rows lists them all, a larger one reads the row
//     occupancy words of its z layers (one layer per lane) and lists the rows whose bit is set;
//   3 the rows cut to the ball by one lane each (ball_row: two table loads per row, all in flight together);
//   4 the candidates of all rows dealt to the sixteen lanes (at most kTeamCand each, requested together), the ones inside the bound
//     compacted into the group's list with their coordinates;
//   the list ranked by the exact key (distance bits, original index): the first six ranks are the neighbours in canonical order, the
//     seventh distance the exact lower bound SET6 certificates want; certificate and plane fit (on the coordinates the list holds: no
//     second gather) by the group's first lane;
//   5 the state written exactly as k_lin's levels 1 and 2 write it.
// A frame of 8 k points is ~2000 waves spread over the device.  k_lin then finds the certificates fresh and runs the stored-plane
// path.  A query the team cannot serve (more layers, rows, candidates or points inside its bound than the lists hold) is left alone:
// k_lin searches it itself.  Results never depend on who searched (history independence).
constexpr int kTeamTile = 4;                      // points per block (one wave)
constexpr int kTeamG = 16;                        // lanes per query
constexpr int kTeamRows = 64;                     // rows of a ball the group handles (four per lane)
constexpr int kTeamCand = 8;                      // candidates per lane
constexpr int kTeamList = 64;                     // points inside the bound the ranking handles
static_assert(kTeamTile >= 1 && kTeamTile <= 64 && kLinBlock % kTeamTile == 0, "one wave tests the tile; a tile lies inside one query block");
struct TeamPassLds {
    uint32_t q_i[kTeamTile], q_kind[kTeamTile], q_cert[kTeamTile], q_pos[kTeamTile][6];
    float q_x[kTeamTile], q_y[kTeamTile], q_z[kTeamTile], q_q0[kTeamTile][3];
    int16_t row_y[4][kTeamRows], row_z[4][kTeamRows];            // offsets from the query's cell
    uint32_t run_s[4][kTeamRows], run_len[4][kTeamRows];         // (run_len: lengths, then their exclusive prefix sums)
    uint32_t c_d2[4][kTeamList], c_idx[4][kTeamList], c_pos[4][kTeamList];
    float c_x[4][kTeamList], c_y[4][kTeamList], c_z[4][kTeamList];
    uint32_t o_d2[4][8], o_pos[4][8], o_idx[4][8];
    float o_x[4][8], o_y[4][8], o_z[4][8];
};
template <bool FAST, bool GATE = false>
static __global__ __launch_bounds__(kWave) void k_advance_team(const float4 *__restrict__ src, uint32_t n_src, GridDev g, PoseArg pose1,
                                                             const PoseArg *__restrict__ poses, LinArgs a, uint32_t *__restrict__ counts,
                                                             const uint32_t *__restrict__ abort_flag, unsigned long long *__restrict__ stamps, GateArgs gt) {
    if (abort_flag && *abort_flag != 0u) return;
    __shared__ TeamPassLds L;
    // (timing probe, option "team_stamps": lane 0 stores the shader clock at the phase boundaries of the block's first round - eight words per block)
    auto stamp = [&](int k) {
        if (stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + k] = __builtin_readcyclecounter();
        }
    };
    stamp(0);
    const int lane = threadIdx.x, grp = lane >> 4, gl = lane & 15;
    PoseArg P;
    if (poses) P = poses[0]; else P = pose1;
    const size_t ss = a.state_stride;
    uint32_t *const sbase = a.state + (size_t)P.state * kStateRows * ss;
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    uint4 *const SV0 = reinterpret_cast<uint4 *>(sbase + kStV0 * ss), *const SX = reinterpret_cast<uint4 *>(sbase + kStX * ss);
    dbl2 *const SV1 = reinterpret_cast<dbl2 *>(sbase + kStV1 * ss), *const SV2 = reinterpret_cast<dbl2 *>(sbase + kStV2 * ss);
    uint32_t *const SW3 = sbase + kStW3 * ss;
    uint2 *const SY = reinterpret_cast<uint2 *>(sbase + kStY * ss);
    // ---- the tests: one point per lane; the stored positions come with the same batch of loads (they are what a failing point needs next)
    uint32_t n_q;
    {
        const uint32_t i = blockIdx.x * (uint32_t)kTeamTile + (uint32_t)lane;
        const bool have = lane < kTeamTile && i < n_src;
        uint32_t kind = 0;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        uint4 v0 = make_uint4(0u, 0u, 0u, 0u), x = make_uint4(kNoIdx, kNoIdx, kNoIdx, kNoIdx);
        uint2 y = make_uint2(kNoIdx, kNoIdx);
        uint32_t w3 = 0u;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) { s4 = src[i]; v0 = SV0[i]; w3 = SW3[i]; x = SX[i]; y = SY[i]; }
        if constexpr (GATE) {  // gated in the kernel: the pose arrives now, the loads above are in flight meanwhile (pose1: state / fresh)
            unsigned long long gate_v = 0ull, gate_seq = 0ull;
            const bool gate_whole = gate_read<__HIP_MEMORY_SCOPE_AGENT>(gt.dev->w, gate_v, gate_seq);
            if (!gate_wait(gt, gate_v, gate_seq, gate_whole, P)) return;
        }
        if (have) {
            body_to_global(P, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
            const float q0x = __uint_as_float(v0.z), q0y = __uint_as_float(v0.w), q0z = __uint_as_float(w3);
            const bool need = !cert_holds(v0.x, q0x, q0y, q0z, qx, qy, qz);
            const bool refit = !need && !cert_is_out(v0.x) && !fit_holds(v0.y, q0x, q0y, q0z, qx, qy, qz);
            kind = need ? 1u : (refit ? 2u : 0u);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(kind != 0u);
        const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (kind != 0u) {
            L.q_i[slot] = i; L.q_x[slot] = qx; L.q_y[slot] = qy; L.q_z[slot] = qz; L.q_kind[slot] = kind; L.q_cert[slot] = v0.x;
            L.q_q0[slot][0] = __uint_as_float(v0.z); L.q_q0[slot][1] = __uint_as_float(v0.w); L.q_q0[slot][2] = __uint_as_float(w3);
            L.q_pos[slot][0] = x.x; L.q_pos[slot][1] = x.y; L.q_pos[slot][2] = x.z; L.q_pos[slot][3] = x.w; L.q_pos[slot][4] = y.x; L.q_pos[slot][5] = y.y;
        }
        n_q = (uint32_t)__builtin_popcountll(m);
    }
    __builtin_amdgcn_wave_barrier();
    stamp(1);
    uint32_t served_s = 0, served_r = 0;                     // (in the first lane of every group: what it served)
    const unsigned long long gmask = 0xFFFFull << (16 * grp);
    auto gballot = [&](bool b) -> uint32_t { return (uint32_t)((__builtin_amdgcn_ballot_w64(b) & gmask) >> (16 * grp)); };
    auto gscan_excl = [&](uint32_t v, uint32_t &total) -> uint32_t {        // exclusive prefix sum over the sixteen lanes of the group
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < kTeamG; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, kTeamG); if (gl >= d) inc += o; }
        total = (uint32_t)__shfl((int)inc, kTeamG - 1, kTeamG);
        return inc - v;
    };
    for (uint32_t r0 = 0; r0 < n_q; r0 += 4u) {
        const uint32_t e = r0 + (uint32_t)grp;
        const bool act = e < n_q;
        const uint32_t i = act ? L.q_i[e] : 0u;
        const float qx = act ? L.q_x[e] : 0.f, qy = act ? L.q_y[e] : 0.f, qz = act ? L.q_z[e] : 0.f;
        const uint32_t kind = act ? L.q_kind[e] : 0u;
        bool srch = kind == 1u;                               // uniform over the group
        uint32_t why = 0;                                     // (probe: why a search was left to k_lin)
        uint32_t n_in = 0;                                    // points inside the bound (uniform over the group)
        float bound = a.radius_sq_f;
        // ---- 1: the six stored neighbours, one per lane: the warm bound of a search, the points of a refit
        {
            float d2 = 0.f;
            bool okp = gl >= 6;
            if (gl < 6 && kind != 0u && (a.warm || kind == 2u)) {
                const uint32_t p = L.q_pos[e][gl];
                if (p != kNoIdx) {
                    const float4 c = g.pts[p];
                    d2 = dist2_nofma(qx, qy, qz, c); okp = true;
                    if (kind == 2u) { L.o_pos[grp][gl] = p; L.o_idx[grp][gl] = __float_as_uint(c.w); L.o_x[grp][gl] = c.x; L.o_y[grp][gl] = c.y; L.o_z[grp][gl] = c.z; }
                }
            }
            const bool have6 = gballot(okp) == 0xFFFFu;
#pragma unroll
            for (int m = 1; m < kTeamG; m <<= 1) d2 = fmaxf(d2, __shfl_xor(d2, m, kTeamG));
            if (srch) {
                if (have6) bound = fminf(bound, fmaxf(__uint_as_float(__float_as_uint(d2) + 1u), 1.17549435e-38f));      // inclusive, as warm_bound6
                const float infl = a.prune_infl, cap = a.infl_max_d2 * a.prune_infl;
                bound = fminf(fmaxf(bound, fminf(bound * infl, cap)), a.radius_sq_f);
            }
        }
        if (r0 == 0u) stamp(2);
        // ---- 2: the rows of the ball
        uint32_t n_rows = 0;                                  // (uniform over the group)
        BallCells bc{};
        if (srch) {
            const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
            const double lim = (double)a.max_ring + 1.0;
            const bool reach = !(fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim);
            bc = ball_cells(g, reach ? qx : (float)g.ox, reach ? qy : (float)g.oy, reach ? qz : (float)g.oz, bound);
            const int ny_r = bc.yhi - bc.ylo + 1, nz_r = bc.zhi - bc.zlo + 1;
            if (!reach) {
                n_rows = 0;                                   // beyond the grid: nothing inside the bound
            } else if (ny_r * nz_r <= kTeamG) {               // a small ball: every row of its bounding square
                n_rows = (uint32_t)(ny_r * nz_r);
                if (gl < (int)n_rows) { L.row_y[grp][gl] = (int16_t)(bc.ylo + gl % ny_r); L.row_z[grp][gl] = (int16_t)(bc.zlo + gl / ny_r); }
            } else if (nz_r > kTeamG || !g.ymask) {
                srch = false; why = 1;                        // more layers than lanes: left to k_lin
            } else {                                          // the occupied rows of every z layer of the ball, one layer per lane
                uint32_t m0 = 0u, m1 = 0u;
                int y0 = 0, y1 = -1;
                bool wide = false;
                const int z = bc.cz + bc.zlo + gl;
                if (gl < nz_r && z >= 0 && z < g.nz) {
                    const float hf = (float)g.h;
                    const int dz = bc.zlo + gl;
                    const float gz = dz < 0 ? ((float)(-dz - 1) + bc.frz) * hf : (dz > 0 ? ((float)dz - bc.frz) * hf : 0.f);
                    const float rem = bound - gz * gz * 0.99999f;
                    if (!(rem < 0.f)) {
                        const float rc = fminf(sqrt_approx(rem) * 1.00001f * (float)g.inv_h + 1e-4f, 1.0e6f);
                        const int cap = 1 << 24;
                        const int ylo = -min(cap, (int)floorf(rc + 1.f - bc.fry)), yhi = min(cap, (int)floorf(rc + bc.fry));
                        const int xlo = -min(cap, (int)floorf(rc + 1.f - bc.frx)), xhi = min(cap, (int)floorf(rc + bc.frx));
                        y0 = max(bc.cy + ylo, 0); y1 = min(bc.cy + yhi, g.ny - 1);
                        const int b0 = max(bc.cx + xlo, 0) >> 4, b1 = min(bc.cx + xhi, g.nx - 1) >> 4;
                        if (y1 >= y0 && b1 >= b0) {
                            wide = ((y1 >> 5) - (y0 >> 5)) > 1 || b1 - b0 > 1;
                            if (!wide) {
                                const int yw0 = y0 >> 5, yw1 = y1 >> 5, bb = min(b0 + 1, b1);
                                const uint32_t *mw = g.ymask + ((int64_t)z * g.nxb + b0) * g.nyw, *mv = g.ymask + ((int64_t)z * g.nxb + bb) * g.nyw;
                                const uint32_t a0_ = mw[yw0], a1_ = mv[yw0], c0_ = mw[yw1], c1_ = mv[yw1];
                                m0 = a0_ | a1_; m1 = yw1 > yw0 ? (c0_ | c1_) : 0u;
                                const int base0 = yw0 << 5;
                                {   const int lo = max(y0 - base0, 0), hi = min(y1 - base0, 31);
                                    m0 &= (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo); }
                                if (yw1 > yw0) { const int lo = 0, hi = min(y1 - (base0 + 32), 31); m1 &= (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo); }
                            }
                        } else { y1 = y0 - 1; }
                    }
                }
                if (gballot(wide) != 0u) { srch = false; why = 2; }       // a layer spanning more words or x blocks than one lane reads: left to k_lin
                uint32_t total = 0;
                const uint32_t mine = (uint32_t)__builtin_popcount(m0) + (uint32_t)__builtin_popcount(m1);
                uint32_t slot = gscan_excl(srch ? mine : 0u, total);
                if (total > (uint32_t)kTeamRows) { srch = false; why = 3; }           // more occupied rows than the group handles: left to k_lin
                if (srch) {
                    const int base0 = (y0 >> 5) << 5, dzl = bc.zlo + gl;
                    while (m0) { const int bit = __builtin_ctz(m0); m0 &= m0 - 1u; L.row_y[grp][slot] = (int16_t)(base0 + bit - bc.cy); L.row_z[grp][slot] = (int16_t)dzl; ++slot; }
                    while (m1) { const int bit = __builtin_ctz(m1); m1 &= m1 - 1u; L.row_y[grp][slot] = (int16_t)(base0 + 32 + bit - bc.cy); L.row_z[grp][slot] = (int16_t)dzl; ++slot; }
                    n_rows = total;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (r0 == 0u) stamp(3);
        // ---- 3: the rows cut to the ball, up to four per lane, their table loads in flight together
        uint32_t n_runs = 0, n_cand = 0;                      // (uniform over the group)
        if (srch) {
            uint32_t rs[kTeamRows / kTeamG], re[kTeamRows / kTeamG];
#pragma unroll
            for (int t = 0; t < kTeamRows / kTeamG; ++t) {
                const uint32_t k = (uint32_t)(gl + kTeamG * t);
                rs[t] = 0u; re[t] = 0u;
                if (k < n_rows) ball_row(g, bc, bound, (int)L.row_y[grp][k], (int)L.row_z[grp][k], rs[t], re[t]);
            }
#pragma unroll
            for (int t = 0; t < kTeamRows / kTeamG; ++t) {
                const bool ne = re[t] > rs[t];
                const uint32_t bits = gballot(ne);
                const uint32_t slot = n_runs + (uint32_t)__builtin_popcount(bits & ((1u << gl) - 1u));
                if (ne) { L.run_s[grp][slot] = rs[t]; L.run_len[grp][slot] = re[t] - rs[t]; }
                n_runs += (uint32_t)__builtin_popcount(bits);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (r0 == 0u) stamp(4);
        // ---- 4: the candidates of all runs, dealt to the sixteen lanes kTeamCand at a time (flat index f = base + gl + 16 j): a chunk's
        // loads are requested together, the points inside the bound are compacted into the group's list with their coordinates
        if (srch) {
            uint32_t off = 0;
            for (uint32_t k = 0; k < n_runs; ++k) { const uint32_t len = L.run_len[grp][k]; L.run_len[grp][k] = off; off += len; }     // lengths -> exclusive offsets
            n_cand = off;
        }
        // (a list that overflows - a loose bound in a dense part of the map - is ranked as it stands: the seventh smallest distance of
        //  ANY 64 real points bounds the sixth neighbour's from above, so the candidates are taken once more against that bound)
        bool again = srch;                                    // this group (re)takes its candidates in the coming attempt
        for (int attempt = 0; attempt < 2 && wave_any(again); ++attempt) {
            if (again) n_in = 0;
            uint32_t k = 0;                                   // the run this lane's next flat index lies in (it only moves forward)
            for (uint32_t base = 0; wave_any(again && base < n_cand); base += (uint32_t)(kTeamCand * kTeamG)) {
                const bool on = again && base < n_cand;
                float4 c[kTeamCand];
                uint32_t cp[kTeamCand];
#pragma unroll
                for (int j = 0; j < kTeamCand; ++j) {
                    const uint32_t f = base + (uint32_t)(gl + kTeamG * j);
                    cp[j] = kNoIdx;
                    if (on && f < n_cand) {
                        while (k + 1u < n_runs && f >= L.run_len[grp][k + 1u]) ++k;
                        cp[j] = L.run_s[grp][k] + (f - L.run_len[grp][k]);
                    }
                    c[j] = g.pts[cp[j] != kNoIdx ? cp[j] : 0u];
                }
#pragma unroll
                for (int j = 0; j < kTeamCand; ++j) {
                    const float d2 = dist2_nofma(qx, qy, qz, c[j]);
                    const bool pass = cp[j] != kNoIdx && d2 < bound;
                    const uint32_t bits = gballot(pass);
                    const uint32_t slot = n_in + (uint32_t)__builtin_popcount(bits & ((1u << gl) - 1u));
                    if (pass && slot < (uint32_t)kTeamList) {
                        L.c_d2[grp][slot] = __float_as_uint(d2); L.c_idx[grp][slot] = __float_as_uint(c[j].w); L.c_pos[grp][slot] = cp[j];
                        L.c_x[grp][slot] = c[j].x; L.c_y[grp][slot] = c[j].y; L.c_z[grp][slot] = c[j].z;
                    }
                    if (on) n_in += (uint32_t)__builtin_popcount(bits);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (r0 == 0u && attempt == 0) stamp(5);
            // ---- rank by (distance bits, original index): a total order; the first seven ranks go to the out list
            if (again) {
                const uint32_t n_l = min(n_in, (uint32_t)kTeamList);
#pragma unroll
                for (int m = 0; m < kTeamList / kTeamG; ++m) {
                    const uint32_t me = (uint32_t)(gl + kTeamG * m);
                    if (me < n_l) {
                        const unsigned long long key = ((unsigned long long)L.c_d2[grp][me] << 32) | L.c_idx[grp][me];
                        uint32_t rank = 0;
                        for (uint32_t j = 0; j < n_l; ++j) {
                            const unsigned long long kj = ((unsigned long long)L.c_d2[grp][j] << 32) | L.c_idx[grp][j];
                            rank += kj < key ? 1u : 0u;
                        }
                        if (rank < 7u) {
                            L.o_d2[grp][rank] = L.c_d2[grp][me]; L.o_pos[grp][rank] = L.c_pos[grp][me]; L.o_idx[grp][rank] = L.c_idx[grp][me];
                            L.o_x[grp][rank] = L.c_x[grp][me]; L.o_y[grp][rank] = L.c_y[grp][me]; L.o_z[grp][rank] = L.c_z[grp][me];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const bool over = again && n_in > (uint32_t)kTeamList;
            if (over && attempt == 1) { srch = false; why = 4; }          // still more than a list's worth (ties): left to k_lin
            if (over && attempt == 0) bound = fmaxf(__uint_as_float(L.o_d2[grp][6] + 1u), 1.17549435e-38f);     // inclusive: the next float up
            again = over && attempt == 0;
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
        if (r0 == 0u) stamp(6);
        // ---- certificate, fit, state: the group's first lane
        const bool lead = gl == 0 && ((kind == 1u && srch) || kind == 2u);
        uint32_t cert = kCertSearch;
        float4 pt[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) pt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        bool six_known = false;                               // the sixth point exists
        if (lead && kind == 1u) {
            Set6 out;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bool got = (uint32_t)j < n_in;
                out.pos[j] = got ? L.o_pos[grp][j] : kNoIdx;
                out.d2[j] = got ? __uint_as_float(L.o_d2[grp][j]) : bound;
                if (got) pt[j] = make_float4(L.o_x[grp][j], L.o_y[grp][j], L.o_z[grp][j], __uint_as_float(L.o_idx[grp][j]));
            }
            out.lb7 = n_in > 6u ? fminf(__uint_as_float(L.o_d2[grp][6]), bound) : bound;
            out.n_eval = 0; out.n_shell = 1;
            cert = make_cert(out, a);
            six_known = out.pos[5] != kNoIdx;
            SX[i] = make_uint4(out.pos[0], out.pos[1], out.pos[2], out.pos[3]);
            SY[i] = make_uint2(out.pos[4], out.pos[5]);
        } else if (lead) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (L.q_pos[e][j] != kNoIdx) pt[j] = make_float4(L.o_x[grp][j], L.o_y[grp][j], L.o_z[grp][j], __uint_as_float(L.o_idx[grp][j]));
            six_known = L.q_pos[e][5] != kNoIdx;
            cert = cert_rebased(L.q_cert[e], L.q_q0[e][0], L.q_q0[e][1], L.q_q0[e][2], qx, qy, qz);
        }
        const bool set = lead && !cert_is_out(cert);
        if (wave_any(set)) {
            const bool use6 = set && cert_is_set6(cert) && six_known;
            const bool six = wave_any(use6);
            if (set) {
                KnnResult<5> nn;
                Fit fit;
                (void)fit_from_points<FAST>(a, qx, qy, qz, pt, use6, six, nn, fit, false);
                SV0[i] = make_uint4(cert, fit.word, __float_as_uint(qx), __float_as_uint(qy));
                SV1[i] = dbl2{fit.plane[0], fit.plane[1]};
                SV2[i] = dbl2{fit.plane[2], fit.plane[3]};
                SW3[i] = __float_as_uint(qz);
            }
        }
        if (lead && !set) {                         // searched and found OUT: certificate and reference position, no fit
            SV0[i] = make_uint4(cert, kFitNone, __float_as_uint(qx), __float_as_uint(qy));
            SW3[i] = __float_as_uint(qz);
        }
        if (lead) { served_s += kind == 1u ? 1u : 0u; served_r += kind == 2u ? 1u : 0u; }
        if (stamps && gl == 0 && kind != 0u) {      // probe: the outcomes, counted behind the blocks' stamp words
            unsigned long long *hist = stamps + (size_t)gridDim.x * 8;
            const uint32_t slack = __float_as_uint(__uint_as_float(cert & 0x7FFFFFFEu));
            const int code = kind == 2u ? 7 : (!srch ? (int)why : (cert_is_out(cert) ? 5 : (slack == 0u ? 6 : 0)));
            atomicAdd(hist + code, 1ull);
        }
        __builtin_amdgcn_wave_barrier();
        if (r0 == 0u) stamp(7);
    }
    // what the block served (the points it left to k_lin are counted there)
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) { served_s += (uint32_t)__shfl_xor((int)served_s, m); served_r += (uint32_t)__shfl_xor((int)served_r, m); }
    // (added to the entry of the query block the tile belongs to: LinArgs::adv_counts - k_lin takes the sums and zeroes them again)
    if (lane == 0 && counts && (served_s | served_r)) {
        const size_t kb = ((size_t)blockIdx.x * kTeamTile) / kLinBlock;      // (one 128-byte line per query block: atomics on one line are served one after the other)
        if (served_s) atomicAdd(&counts[kb * kCounterStride], served_s);
        if (served_r) atomicAdd(&counts[kb * kCounterStride + 1], served_r);
    }
    if (lane == 0 && a.search_count && served_s)              // (option "count_searches": the pass's searches count like k_lin's)
        atomicAdd(a.search_count + (size_t)(blockIdx.x & 63u) * (kCounterStride / 2), (unsigned long long)served_s);
}

// Batched launches: one block per pose sums that pose's block partials with the SAME association order as the fused
// single-pose path (chunk sums, then chunks in index order), so a batched pose is bitwise equal to the same pose
// linearised alone.  Writes 31 sums + check word to the pinned, host-coherent result row the host spins on (no stream synchronise on
// the hot path).  out row layout: [0..30] sums, [31] = check word (publish_row).
static __global__ __launch_bounds__(kLinBlock) void k_finalize(const double *__restrict__ partials, uint32_t n_blocks, double *__restrict__ out,
                                                            unsigned long long seq) {
    __shared__ double sm[kLinBlock / 32][kSlots];
    const uint32_t pose_id = blockIdx.x;
    const double *base = partials + (size_t)pose_id * n_blocks * kSlots;
    double tot = 0.0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += kChunk)
        tot += block_sum_rows(base + (size_t)c0 * kSlots, min((uint32_t)kChunk, n_blocks - c0), sm);
    double *orow = out + (size_t)pose_id * kSlots;
    if (threadIdx.x < 32) publish_row(orow, tot, seq);
}

// ---------------------------------------------------------------- plain k-NN kernel (p2p metrics, tests)
// SWEEP: bounded searches only (the row sweep covers the ball of the bound: search.hpp knn_shells); experiments (dcreg_knn_timed)
template <int K, bool SWEEP = false>
static __global__ __launch_bounds__(kBlock) void k_knn(const float4 *__restrict__ q, uint32_t n, GridDev g, float bound_f, int max_ring,
                                                 PoseArg pose, int apply_pose, int32_t *__restrict__ idx, float *__restrict__ d2) {
    __shared__ RunList runs[kBlock / kWave];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 s4 = q[i];
    float qx = s4.x, qy = s4.y, qz = s4.z;
    if (apply_pose) {   // pcl::transformPointCloud<PointT,double>: double arithmetic, float store
        body_to_global(pose, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
    }
    KnnResult<K> nn;
    knn_exact<K, SWEEP>(g, runs[threadIdx.x / kWave], qx, qy, qz, bound_f, max_ring, nn);
    const uint32_t oi = __float_as_uint(s4.w);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool ok = nn.idx[j] != kNoIdx;
        idx[(size_t)oi * K + j] = ok ? (int32_t)nn.idx[j] : -1;
        d2[(size_t)oi * K + j] = ok ? nn.d2[j] : __builtin_inff();
    }
}

// ---------------------------------------------------------------- index build kernels
static __global__ void k_pack(const float *__restrict__ xyz, int64_t n, int64_t stride, float4 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_float4(xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2], __uint_as_float((uint32_t)i));
}

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    union { uint32_t u; float f; } cv; cv.u = u; return cv.f;
}

// the window index of a large map (context.hip roi_ensure): the map's points in the cells [x0, x1) x [y0, y0 + ny_box) x [z0, ..) of the WHOLE
// map's index are the x-runs of n_rows (y,z) rows - contiguous ranges of the cell-sorted points.  k_roi_rows: their lengths;
// k_roi_copy (one block per row): the runs one after the other into the window's raw cloud (the points keep their original index in w).
static __global__ void k_roi_rows(const uint32_t *__restrict__ cell_start, int nxs, int ny, int x0s, int x1s, int y0, int ny_box, int z0, int n_rows,
                                  uint32_t *__restrict__ len) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const size_t row = ((size_t)(z0 + r / ny_box) * (size_t)ny + (size_t)(y0 + r % ny_box)) * (size_t)nxs;
    len[r] = cell_start[row + (size_t)x1s] - cell_start[row + (size_t)x0s];
}
static __global__ __launch_bounds__(256) void k_roi_copy(const float4 *__restrict__ pts, const uint32_t *__restrict__ cell_start, int nxs, int ny, int x0s, int x1s,
                                                        int y0, int ny_box, int z0, const uint32_t *__restrict__ off, float4 *__restrict__ out) {
    const int r = blockIdx.x;
    const size_t row = ((size_t)(z0 + r / ny_box) * (size_t)ny + (size_t)(y0 + r % ny_box)) * (size_t)nxs;
    const uint32_t s = cell_start[row + (size_t)x0s], n = cell_start[row + (size_t)x1s] - s, o = off[r];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[(size_t)o + i] = pts[(size_t)s + i];
}

// bounds[0..2] = min (ordered-uint), bounds[3..5] = max
static __global__ void k_bounds(const float4 *__restrict__ p, int64_t n, uint32_t *bounds) {
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 c = p[i];
        mn[0] = fminf(mn[0], c.x); mn[1] = fminf(mn[1], c.y); mn[2] = fminf(mn[2], c.z);
        mx[0] = fmaxf(mx[0], c.x); mx[1] = fmaxf(mx[1], c.y); mx[2] = fmaxf(mx[2], c.z);
        // a NaN passes through fminf / fmaxf unnoticed: any non-finite coordinate makes the upper bound infinite, which the host rejects
        if (!(fabsf(c.x) <= 3.4e38f) || !(fabsf(c.y) <= 3.4e38f) || !(fabsf(c.z) <= 3.4e38f)) mx[0] = __builtin_inff();
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
    }
    // block level first: 6 atomics per block instead of per wave (the 6 target words serialise them)
    __shared__ float smn[4][3], smx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { smn[wave][a] = mn[a]; smx[wave][a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = smn[0][a], hi = smx[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, smn[w][a]); hi = fmaxf(hi, smx[w][a]); }
        atomicMin(&bounds[a], f2ord(lo)); atomicMax(&bounds[3 + a], f2ord(hi));
    }
}

static __global__ void k_cell_keys(const float4 *__restrict__ p, int64_t n, GridDev g, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = p[i];
    // x in sub-cells (GridDev::sx per cell); the sub-cell index over sx is the cell the searches take for this point
    const int nxf = g.nx * g.sx;
    const int cx = clampi((int)floor(((double)c.x - g.ox) * g.inv_h * (double)g.sx), 0, nxf - 1);
    const int cy = clampi((int)floor(((double)c.y - g.oy) * g.inv_h), 0, g.ny - 1);
    const int cz = clampi((int)floor(((double)c.z - g.oz) * g.inv_h), 0, g.nz - 1);
    keys[i] = (uint32_t)(((int64_t)cz * g.ny + cy) * nxf + cx);
    vals[i] = (uint32_t)i;
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
// Hilbert-curve key (Skilling's transpose algorithm, 21 bits per axis).  Consecutive keys are always
// spatial neighbours (no Z-order seams), which keeps the cells a wave touches close together.
__device__ __forceinline__ uint64_t curve_key(const float4 &c, double ox, double oy, double oz, double inv_q, double x_scale) {
    // (x_scale < 1: the curve's cells are 1 / x_scale times as long in x as in y and z - the patches of consecutive points stretch along
    //  the rows of the target index)
    const double fx = ((double)c.x - ox) * inv_q * x_scale, fy = ((double)c.y - oy) * inv_q, fz = ((double)c.z - oz) * inv_q;
    uint32_t X[3] = {(uint32_t)fmin(fmax(fx, 0.0), 2097151.0), (uint32_t)fmin(fmax(fy, 0.0), 2097151.0),
                     (uint32_t)fmin(fmax(fz, 0.0), 2097151.0)};
    const uint32_t M = 1u << 20;
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread21(X[0]) << 2) | (spread21(X[1]) << 1) | spread21(X[2]);
}
static __global__ void k_curve_keys(const float4 *__restrict__ p, int64_t n, double ox, double oy, double oz, double inv_q, double x_scale,
                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = curve_key(p[i], ox, oy, oz, inv_q, x_scale);
    vals[i] = (uint32_t)i;
}

static __global__ void k_gather4(const float4 *__restrict__ in, const uint32_t *__restrict__ order, int64_t n, float4 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[order[i]];
}

// Heavy groups first.  The blocks of a launch are dispatched in index order and the launch ends when its slowest block does: where the
// expensive queries - far from the surface they belong to (many rows to sweep), or in a dense part of the map (many candidates) - sit
// at the end of the order, the device idles while they finish.  A single-pose launch with more query blocks than the device holds
// at once can therefore take them in GROUPS (group_blocks = a multiple of 16 consecutive blocks = one XCD's run, xcd_remap; at most
// kMaxGroups groups) by decreasing ESTIMATED cost - longest processing time first.  The estimate is made at the pose of the first
// such launch after dcreg_set_source / dcreg_set_target and again when a launch whose pose is known up front is a cell or more away
// from it (k_group_cost below samples every group: cells to the nearest occupied one, points in the 3x3x3 block); it is used while
// the caller's misalignment hint (dcreg_hint_misalignment: the engines pass the RMS residual) is above half a cell - on a settled
// trajectory every block costs the same and streams its state rows, which goes 10 % faster in index order.  The order schedules and
// nothing else: partial rows and chunk sums are indexed by query block, so the 31 sums are bitwise the same for every order.
// Measured (1 M x 1 M corridor, first four iterations of a run): 575 / 423 / 337 / 241 us in index order, 431 / 263 / 234 / 211 us so.
constexpr int kMaxGroups = 256;     // (LinArgs::group_order holds bytes)
constexpr int kCostSamples = 4;      // per thread: 1024 samples per group
static __global__ __launch_bounds__(256) void k_group_cost(const float4 *__restrict__ src, uint32_t n, GridDev g, PoseArg P, uint32_t group_points,
                                                          int max_ring, float *__restrict__ cost) {
    __shared__ float sm[256];
    const uint32_t base = blockIdx.x * group_points;
    const uint32_t step = max(group_points / (256u * kCostSamples), 1u);
    float acc = 0.f;
    for (int k = 0; k < kCostSamples; ++k) {
        const uint32_t j = ((uint32_t)k * 256u + threadIdx.x) * step;
        if (j >= group_points || base + j >= n) continue;
        const float4 p = src[base + j];
        float qx, qy, qz;
        body_to_global(P, (double)p.x, (double)p.y, (double)p.z, qx, qy, qz);
        const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
        const double lim = (double)max_ring + 1.0;
        float c = 1.f;                                   // transform, tests, row: every query
        if (!(fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim)) {
            const int cx = clampi((int)floor(fx), 0, g.nx - 1), cy = clampi((int)floor(fy), 0, g.ny - 1), cz = clampi((int)floor(fz), 0, g.nz - 1);
            const int f = g.gap ? (int)g.gap[((int64_t)cz * g.ny + cy) * g.nx + cx] : 0;
            uint32_t c27 = 0;
            const int64_t nxf = (int64_t)g.nx * g.sx;
            for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
                const int y = cy + dy, z = cz + dz;
                if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
                const int64_t row = ((int64_t)z * g.ny + y) * nxf;
                c27 += g.cell_start[row + (int64_t)min(cx + 2, g.nx) * g.sx] - g.cell_start[row + (int64_t)max(cx - 1, 0) * g.sx];
            }
            c += (float)c27 * (1.f / 32.f);
            if (f >= 2) c += f == 255 ? 4.f : 6.f + 1.5f * (float)f;      // the sweep of a ball of f cells and more
        }
        acc += c;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[blockIdx.x] = sm[0];
}

// cell_start[c] = first sorted position whose key >= c (sorted keys ascending), c = 0 .. n_cells: one thread per CELL and a
// binary search in the key array.  (Round 1 had one thread per POINT fill the run of empty cells in front of it - a single
// thread then wrote every cell of a long empty stretch, 0.75 ms on the 1 M corridor; the search is ~20 dependent, cached
// loads per cell whatever the occupancy.)  n_occupied counts cells that own at least one point, one atomic per wave.
static __global__ void k_cell_start(const uint32_t *__restrict__ keys, int64_t n, int64_t n_cells, uint32_t *__restrict__ cell_start,
                             uint32_t *__restrict__ n_occupied) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool occ = false;
    if (c <= n_cells) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
        }
        cell_start[c] = (uint32_t)lo;
        occ = c < n_cells && lo < n && (int64_t)keys[lo] == c;
    }
    const unsigned long long m = __ballot(occ);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_occupied, (uint32_t)__popcll(m));
}

// empty-space distance field of the target grid: gap[c] = 0 on occupied cells, then one dilation pass per ring.
// In place: a pass only turns 255 into `ring`, and only looks for neighbours equal to ring - 1.
static __global__ void k_gap_init(const uint32_t *__restrict__ cell_start, int64_t n_cells, int sx, uint8_t *__restrict__ gap, uint32_t *__restrict__ owner) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // cell; its sx sub-cells are consecutive table entries
    if (c < n_cells) {
        const bool occ = cell_start[(c + 1) * sx] > cell_start[c * sx];
        gap[c] = occ ? 0 : 255;
        owner[c] = occ ? (uint32_t)c : kNoIdx;
    }
}
// The owners the start-bound probe of lin_search6 uses come from a second field of the same kind whose seeds are the DENSE cells only: cells
// whose three-cell x-run - the very run the probe scans - holds at least min_pts points.  (Round 6: with every occupied cell as a seed the
// nearest seed of a query 0.8 m below a ceiling was a cell holding the noise tail of that ceiling, one cell layer in front of it - a handful
// of points, no bound to be had from them; such queries started from the search radius: 350 candidates each, their waves what a run's first
// launch lasted.  profiles/r06_ablation.md section 3.)  gap2 / owner2 are scratch of the index build; k_owner_merge keeps the dense owner
// where there is one within the rings and the plain one elsewhere.
static __global__ void k_gap_init_dense(const uint32_t *__restrict__ cell_start, int nx, int64_t n_cells, int sx, uint32_t min_pts, uint8_t *__restrict__ gap,
                                        uint32_t *__restrict__ owner) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cells) {
        const int x = (int)(c % nx);
        const int64_t row = c - x;
        const bool own = cell_start[(c + 1) * sx] > cell_start[c * sx];
        const uint32_t run = cell_start[(row + min(x + 2, nx)) * sx] - cell_start[(row + max(x - 1, 0)) * sx];
        const bool dense = own && run >= min_pts;
        gap[c] = dense ? 0 : 255;
        owner[c] = dense ? (uint32_t)c : kNoIdx;
    }
}
static __global__ void k_owner_merge(uint32_t *__restrict__ owner, const uint32_t *__restrict__ dense_owner, int64_t n_cells) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cells && dense_owner[c] != kNoIdx) owner[c] = dense_owner[c];
}
// (a cell of ring r takes, of the owners of its neighbours of ring r - 1, the one nearest to itself - vector propagation, so the owner
// stays near the foot of the perpendicular instead of drifting along the diagonal; rings r - 1 are final when ring r is written, and
// a launch only writes cells that are still 255: no cell is read and written in the same launch with a value that matters)
static __global__ void k_gap_dilate(uint8_t *gap, uint32_t *owner, int nx, int ny, int nz, int ring) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_cells = (int64_t)nx * ny * nz;
    if (c >= n_cells || gap[c] != 255) return;
    const int x = (int)(c % nx), y = (int)((c / nx) % ny), z = (int)(c / ((int64_t)nx * ny));
    const uint8_t want = (uint8_t)(ring - 1);
    uint32_t own = kNoIdx;
    int64_t best = INT64_MAX;
    for (int dz = -1; dz <= 1; ++dz) {
        const int zz = z + dz;
        if (zz < 0 || zz >= nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= ny) continue;
            const int64_t row = ((int64_t)zz * ny + yy) * nx;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= nx || gap[row + xx] != want) continue;
                const uint32_t o = owner[row + xx];
                const int64_t ox = o % (uint32_t)nx, oy = (o / (uint32_t)nx) % (uint32_t)ny, oz = o / ((uint32_t)nx * (uint32_t)ny);
                const int64_t d = (ox - x) * (ox - x) + (oy - y) * (oy - y) + (oz - z) * (oz - z);
                if (d < best) { best = d; own = o; }
            }
        }
    }
    if (own != kNoIdx) { gap[c] = (uint8_t)ring; owner[c] = own; }
}

// row occupancy words of the row sweep (GridDev::ymask): one thread per word, 32 rows y of one (z, 16-cell x block)
static __global__ void k_ymask(const uint32_t *__restrict__ cell_start, int nx, int ny, int nz, int sx, int nxb, int nyw, uint32_t *__restrict__ ymask) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (int64_t)nz * nxb * nyw) return;
    const int yw = (int)(w % nyw), xb = (int)((w / nyw) % nxb), z = (int)(w / ((int64_t)nyw * nxb));
    const int xa = xb * 16, xe = min(xa + 16, nx);
    const int64_t nxf = (int64_t)nx * sx;
    uint32_t m = 0;
    for (int b = 0; b < 32; ++b) {
        const int y = yw * 32 + b;
        if (y >= ny) break;
        const int64_t row = ((int64_t)z * ny + y) * nxf;
        if (cell_start[row + (int64_t)xe * sx] > cell_start[row + (int64_t)xa * sx]) m |= 1u << b;
    }
    ymask[w] = m;
}

// reductions for dcreg_p2p_error: sum sqrt(d2), sum d2 [dist<thr], count  (deterministic two-stage)
static __global__ __launch_bounds__(kBlock) void k_p2p_partial(const float *__restrict__ d2, int64_t n, double thr, double *__restrict__ part) {
    __shared__ double tile[kBlock / 64][4];
    double s_d = 0.0, s_sq = 0.0, cnt = 0.0;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const float v = d2[i];
        if (v < __builtin_inff()) {
            const float dist = sqrtf(v);            // std::sqrt(float), utils.hpp:557
            s_d = (double)dist;
            if ((double)dist < thr) { s_sq = (double)v; cnt = 1.0; }   // the reference compares against the double threshold (utils.hpp:560)
        }
    }
    s_d = wave_sum_to_lane63(s_d); s_sq = wave_sum_to_lane63(s_sq); cnt = wave_sum_to_lane63(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) { tile[wave][0] = s_d; tile[wave][1] = s_sq; tile[wave][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += tile[w][threadIdx.x];
        part[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
}

}  // namespace dcreg
