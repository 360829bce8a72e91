// Per-thread device functions of the DCReg hot path (gfx950): exact 5-NN on the cell grid, 5x3 plane fit, point-to-plane
// row (DCReg/src/icp_test_runner.cpp:1714-1907).  Included by kernels.hpp (the __global__ kernels and the reductions).
//
// The same functions also compile for the host when the includer defines DCREG_HOST_EMUL and has provided the handful of device
// types and intrinsics they use (float4, bit casts, a threadIdx): the test suite does that to replay the device algorithm on the
// CPU against the oracle.  That is test infrastructure, outside this package: nothing here links or loads it and
// libdcreg_hip.so has no host path.
//
// Layout in HBM
//   target : float4 {x,y,z,bits(orig_idx)} sorted by linear grid cell (x fastest) + cell_start[n_cells+1]
//            -> the three x-adjacent cells of one (y,z) row are ONE contiguous run of points
//   source : float4 {x,y,z,bits(orig_idx)} sorted by the Hilbert-curve key of the body-frame position, so the 64
//            lanes of a wave walk neighbouring cells (a rigid pose keeps neighbours neighbours)
// Arithmetic: k-NN distances float32, non-fused, summed x,y,z in that order (what FLANN's L2 functor does
// and what the oracle does); everything after the neighbour set is fp64, like the reference.
#pragma once
#include <stddef.h>
#include <stdint.h>
#if defined(DCREG_HOST_EMUL)
#define DCREG_DEVFN inline
#define DCREG_ON_DEVICE 0
#else
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "these kernels are written for gfx950 (CDNA4): inline ISA, wave64 DPP, sc1 coherence protocol"
#endif
#define DCREG_DEVFN __device__ __forceinline__
#define DCREG_ON_DEVICE 1
#endif

namespace dcreg {

#if DCREG_ON_DEVICE
DCREG_DEVFN float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
#define DCREG_STAT(field) ((void)0)
#define DCREG_TRACE(kk, dz, dy, which, trips) ((void)0)
#else
inline float med3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
#define DCREG_STAT(field) (++emu_stats.field)
#define DCREG_TRACE(kk, dz, dy, which, trips) emu_trace_push(((((uint32_t)(kk) << 20) | ((uint32_t)((dz) + 512) << 10) | (uint32_t)((dy) + 512)) << 1) | (uint32_t)(which), (uint32_t)(trips))
#endif

constexpr int kBlock = 256;          // 4 waves: the utility kernels
constexpr int kLinOcc = 4;           // waves per SIMD the linearisation kernel is compiled for (register budget 512 / that)
constexpr int kLinBlock = 256;       // threads per block of the linearisation kernel: one partial row per kLinBlock source points
constexpr int kSlots = 32;           // doubles per partial row
constexpr uint32_t kNoIdx = 0xFFFFFFFFu;

struct GridDev {
    double ox, oy, oz;   // origin (min corner)
    double inv_h, h;
    int nx, ny, nz;
    int sx;                       // x sub-cells per cell (>= 1): the point order and the table are (z, y, x-sub-cell) row-major, so a
                                  // (y,z) row is still ONE contiguous run per x-interval, but the interval is cut sx times finer
    uint32_t n_pts;
    const uint32_t *cell_start;   // [nx*sx*ny*nz + 1], entry ((z*ny + y)*nx + x)*sx + sub
    const float4 *pts;            // sorted target; kPtsPad readable entries follow the last point (candidate loads come in batches)
    const uint8_t *gap;           // [nx*ny*nz] Chebyshev distance (cells) to the nearest occupied cell, 255 = more than
    int gap_cap;                  //   gap_cap; null = not built.  Lets a query in empty space skip the rings it knows are empty
    const uint32_t *owner;        // [nx*ny*nz] with the field: an occupied cell at that distance (kNoIdx beyond gap_cap); null = not built.
                                  //   Real points near a far query: the start bound of its search (lin_search6)
    const uint32_t *ymask;        // [nz][nxb][nyw] row occupancy: bit (y & 31) of word ((z * nxb + (x >> 4)) * nyw + (y >> 5)) is set iff
    int nxb, nyw;                 //   one of the 16 cells (x', y, z), x' >> 4 == x >> 4, holds a point.  The bounded searches of the
                                  //   linearisation sweep the occupied rows of their ball through these words (knn_shells<.., true>)
};

struct PoseArg {
    double R[9]; double t[3];
    uint32_t state;      // which neighbour state this pose reads and updates (kNoIdx = none: search cold, keep nothing); single pose: 0
    uint32_t fresh;      // 1: the state holds nothing yet (never searched, or its clouds changed): every query is searched and the old
                         // contents are not read - a state never needs clearing
};
// Neighbour state of one source cloud against one target (one per single-pose context, one per Monte-Carlo trial slot):
//   uint32 [kStateRows][stride]   rows 0-5: positions in the sorted target of the query's 6 nearest neighbours as of its last search,
//                                 ascending (kNoIdx = fewer were found inside the search bound);
//                                 rows 7-9: the query's position q0 at that search (the float-stored transform, bit patterns);
//                                 row 6: the CERTIFICATE of that search, a float s >= 0 (bit pattern) with a mode in the sign bit and
//                                 the lowest mantissa bit:
//     SET5 (sign 0, low bit 0): while the query stays within s metres of q0 its 5-nearest SET cannot change: s = (a5 - a4) / 2 of
//                   the 5th / 6th neighbour distances a4 / a5 at q0.  A linearisation at a pose that keeps it there needs no search:
//                   it gathers the 5 points, recomputes the five float distances and sorts them - bitwise what a fresh search returns.
//     SET6 (sign 0, low bit 1): within s = (a6 - a4) / 2 metres the five nearest are AMONG the six known points (a6 = a lower
//                   bound of the 7th neighbour's distance, which the search gets almost for free: what it looked at and did not keep,
//                   and the pruning radius it never looked beyond): gather 6, sort, take the first five.  Two neighbour gaps
//                   instead of one: such a certificate fails quadratically less often.
//     OUT (sign 1): the 5th neighbour was beyond the search radius by s metres (or not found at all inside the slightly larger search
//                   bound): within s metres of q0 the query fails the radius gate (:1726) and nothing needs to be loaded at all.
// Nothing of a state changes between two searches of a query, and the test is on the two stored float positions themselves (their
// difference is exact), so neither the length of a trajectory nor the rounding of the float store wears a certificate down.
//                                 row 10: the FIT word, rows 11-18: the plane of the last fit (four doubles as eight words): FitCert below
// IN MEMORY the nineteen words of a point are grouped so that a lane fetches them with five wide loads instead of nineteen narrow ones
// (every load in flight holds a 64-bit address in two registers: thirteen of them at the top of the kernel were the largest single
// block of its register budget).  With S = stride (words per row, a multiple of 64) one state is 19 S words:
//     V0 [S] x 16 B : certificate (row 6), fit word (10), q0.x (7), q0.y (8)
//     V1 [S] x 16 B : plane[0], plane[1] as two doubles (rows 11-14): they land in aligned register pairs, nothing to assemble
//     V2 [S] x 16 B : plane[2], plane[3] (rows 15-18)
//     W3 [S] x  4 B : q0.z (row 9)                                 -- V0, V1, V2, W3 = the 52 B the fast path reads
//     X  [S] x 16 B : positions 0-3 (rows 0-3)
//     Y  [S] x  8 B : positions 4-5 (rows 4-5)
// state_word_index(row, i, S) is the word of `row` of point i (host replay and odd accesses; the kernels use the vector forms).
constexpr int kStateRows = 19;
constexpr uint32_t kCertSearch = 0xFFFFFFFFu;      // (a NaN: no certificate)
constexpr int kStV0 = 0, kStV1 = 4, kStV2 = 8, kStW3 = 12, kStX = 13, kStY = 17;      // group bases in units of S words
DCREG_DEVFN size_t state_word_index(int row, size_t i, size_t S) {
    //                      row:   0   1   2   3   4   5   6  7  8   9 10 11 12 13 14 15 16 17 18
    constexpr uint8_t grp[19] = {13, 13, 13, 13, 17, 17,  0, 0, 0, 12, 0, 4, 4, 4, 4, 8, 8, 8, 8};
    constexpr uint8_t wpe[19] = { 4,  4,  4,  4,  2,  2,  4, 4, 4,  1, 4, 4, 4, 4, 4, 4, 4, 4, 4};
    constexpr uint8_t sub[19] = { 0,  1,  2,  3,  0,  1,  0, 2, 3,  0, 1, 0, 1, 2, 3, 0, 1, 2, 3};
    return (size_t)grp[row] * S + i * wpe[row] + sub[row];
}

struct LinArgs {
    double radius_sq;             // R^2 in double (gate :1726)
    float radius_sq_f;            // the SEARCH bound: smallest float above (R (1 + cert_margin))^2 - searches cover a little more than
                                  // the gate radius so that "5th neighbour beyond R" can be certified with some slack
    float cert_r_out;             // R (1 + 1e-5), rounded up: OUT certificates measure from here
    float cert_r_in;              // R (1 - 1e-5), rounded down: what the 5th neighbour must stay below for the radius gate to hold
    double max_thick_sq, min_norm, w_slope, w_min;
    int use_wd;
    int max_ring;                 // rings needed to cover the search bound
    int warm;                     // searches of a non-fresh state are bounded by the old neighbours' distances from the new position
    int use_cert;                 // ... and skipped for the points whose certificate still holds (0: debug dumps search everything)
    int team_max;                 // a wave with at most this many lanes to search (<= kTeamMax; 0 = never) serves them one at a time
                                  // with all 64 lanes (team_search6) instead of searching in lock-step
    float prune_infl;             // (1 + cert_inflate)^2: the searches prune at the 6th best distance x (1 + cert_inflate), which is
                                  // what makes the 7th neighbour's lower bound - and SET6 certificates - worth something ...
    float infl_max_d2;            // ... for searches bounded by at most this squared distance (a couple of cells)
    unsigned long long *search_count;   // test / profiling hook: points searched are counted here (one atomic per searching wave), or null
    uint32_t *state;              // [state][kStateRows][state_stride], or null (nothing is kept)
    uint32_t state_stride;
    uint32_t group_blocks, n_groups;   // heavy groups first (kernels.hpp k_group_cost): dispatch slot -> group of group_blocks query blocks;
    uint8_t group_order[256];          //   n_groups = 0: index order.  By value: the entry is fetched with the other launch arguments
    uint32_t xcd_chunk;           // block -> query-block mapping: 0 = one contiguous run of query blocks per XCD, c = chunks of c blocks dealt round-robin
    double count_scale;           // 2^26, or 0: the two count slots of a partial row also carry, above the counts themselves, how many
                                  // points of the launch were searched (level 1) and refitted (level 2): slot 29 = n_eff + scale x
                                  // searched, slot 30 = n_pt + scale x refitted - integers far below 2^53, so the fp64 sums stay exact
                                  // and the host splits them again (context.hip linearize_end).  What the host does with them:
                                  // scheduling only (which instantiation the next launch uses) and the launch statistics.  0 for
                                  // clouds of more than 2^26 points.
    uint32_t *adv_counts;         // a launch that runs behind an advance pass (kernels.hpp k_advance / k_advance_team): per QUERY BLOCK of this
                                  // launch the points the pass searched and refitted among its points, [n_blocks][32] (a line each: words 0, 1); the
                                  // block adds them to the counts it reports (count_scale) and zeroes them again for the next launch.
                                  // null: no pass in front
    float far_loose;              // a start bound counts as loose - worth a probe of the points around the nearest occupied cell - when it
                                  // reaches this many cells beyond the distance to that cell (lin_search6)
    int euler;                    // 1: roll/pitch/yaw row of the second engine (:2299-2346) instead of the SO(3) row
    const double *dR;             // euler: 27 doubles in device memory - the bracket coefficients of that row (context.hip make_lin_args),
                                  // row-major (behind a pointer: as a member the 54 words would be hoisted into registers for every launch)
};

// ---------------------------------------------------------------- k-NN heaps (sorted, K entries)

// Exact heap: key = (float bits of d2) << 32 | original index -> total order (d2, idx), ties -> lower index.
template <int K_>
struct HeapExact {
    static constexpr int K = K_;
    static constexpr bool kDeferred = false;
    uint64_t key[K];
    uint32_t pos[K];
    uint32_t n_eval;     // candidates evaluated (statistics only; dead code unless read)
    uint32_t n_shell;    // outermost shell scanned
    float infl, cap;     // the walk prunes at worst_d2() = inflated(K-th best): see HeapFast
    DCREG_DEVFN void init(float bound_f, float infl_ = 1.f, float cap_ = __builtin_inff()) {
        infl = infl_; cap = cap_;
        const uint64_t bound = ((uint64_t)__float_as_uint(bound_f) << 32) | 0xFFFFFFFFull;
#pragma unroll
        for (int i = 0; i < K; ++i) { key[i] = bound; pos[i] = kNoIdx; }
        n_eval = 0; n_shell = 1;
    }
    DCREG_DEVFN void push(float d2, uint32_t idx, uint32_t p, bool valid = true) {
        n_eval += valid ? 1u : 0u;
        const uint64_t k = ((uint64_t)__float_as_uint(d2) << 32) | (uint64_t)idx;
        if (valid && k < key[K - 1]) {
            key[K - 1] = k; pos[K - 1] = p;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                const bool sw = key[j] < key[j - 1];
                const uint64_t ka = key[j - 1], kb = key[j];
                const uint32_t pa = pos[j - 1], pb = pos[j];
                key[j - 1] = sw ? kb : ka; key[j] = sw ? ka : kb;
                pos[j - 1] = sw ? pb : pa; pos[j] = sw ? pa : pb;
            }
        }
    }
    DCREG_DEVFN float worst_d2() const { const float d = __uint_as_float((uint32_t)(key[K - 1] >> 32)); return fmaxf(d, fminf(d * infl, cap)); }
    DCREG_DEVFN float dist(int j) const { return __uint_as_float((uint32_t)(key[j] >> 32)); }
    DCREG_DEVFN bool full() const { return pos[K - 1] != kNoIdx; }
};

// Fast heap: 32-bit keys (d2 only, strict <), branch-light insertion.  It yields the exact neighbour SET unless some
// point outside the final heap has d2 == the K-th best d2; `outside_min` tracks the smallest d2 that was ever kept out
// (rejected candidates and evicted entries alike: max(d2, K-th best before the push) is exactly that value), so the
// tie is detected exactly and the caller re-runs the exact heap.  Order among equal d2 inside the heap is fixed
// afterwards (canonical (d2, idx) order).
template <int K_>
struct HeapFast {
    static constexpr int K = K_;
    static constexpr bool kDeferred = true;
    // a point that was never offered to push() (filtered out by a bound >= the K-th best) stays outside at distance d2
    DCREG_DEVFN void note_outside(float d2) { outside_min = fminf(outside_min, d2); }
    float d[K];
    uint32_t pos[K];
    float outside_min;   // smallest d2 among all points seen that are not in the heap
    uint32_t n_eval, n_shell;
    float infl, cap;     // the walk prunes at worst_d2() = max(d, min(d x infl, cap)) of the K-th best squared distance d: a small ball
                         // (a query among its neighbours) is searched a little beyond what exactness needs, a large one is not -
                         // continuous and non-decreasing in d, so everything the walk never looked at is at least worst_d2() away at
                         // the end; everything it looked at and did not keep is in outside_min - together a lower bound for the
                         // (K+1)-th neighbour
    DCREG_DEVFN void init(float bound_f, float infl_ = 1.f, float cap_ = __builtin_inff()) {
        infl = infl_; cap = cap_;
#pragma unroll
        for (int i = 0; i < K; ++i) { d[i] = bound_f; pos[i] = kNoIdx; }
        outside_min = __builtin_inff();
        n_eval = 0; n_shell = 1;
    }
    // valid == false: the slot is padding (d2 must then be +inf).  Branch-free: with 64 queries per wave some lane
    // accepts almost every candidate, so a divergent "if (d2 < worst)" is taken anyway and only adds exec-mask
    // juggling and merge copies.  Sorted insertion without a dependency chain: entry i becomes the median of
    // (d[i-1], d[i], d2); positions follow the same selection through the masks c[i] = d2 < d[i] (ties stay behind).
    template <bool COUNT = true>
    DCREG_DEVFN void push(float d2, uint32_t /*idx*/, uint32_t p, bool valid = true) {
        if (COUNT) n_eval += valid ? 1u : 0u;
#if DCREG_ON_DEVICE
        if constexpr (K == 5) {
            // 21 VALU instructions, written out: the compiler's select canonicalisation turns the nine position
            // selects into 13-20 when it sees several pushes at once.  All compares read the OLD distances and sit
            // at least five instructions ahead of the v_cndmask that consumes their SGPR mask.
            unsigned long long m0, m1, m2, m3, m4;
            float t;
            asm("v_cmp_lt_f32_e64 %[m0], %[x], %[d0]\n\t"
                "v_cmp_lt_f32_e64 %[m1], %[x], %[d1]\n\t"
                "v_cmp_lt_f32_e64 %[m2], %[x], %[d2]\n\t"
                "v_cmp_lt_f32_e64 %[m3], %[x], %[d3]\n\t"
                "v_cmp_lt_f32_e64 %[m4], %[x], %[d4]\n\t"
                "v_max_f32_e32 %[t], %[x], %[d4]\n\t"
                "v_min_f32_e32 %[om], %[om], %[t]\n\t"
                "v_med3_f32 %[d4], %[d3], %[d4], %[x]\n\t"
                "v_med3_f32 %[d3], %[d2], %[d3], %[x]\n\t"
                "v_med3_f32 %[d2], %[d1], %[d2], %[x]\n\t"
                "v_med3_f32 %[d1], %[d0], %[d1], %[x]\n\t"
                "v_min_f32_e32 %[d0], %[d0], %[x]\n\t"
                "v_cndmask_b32_e64 %[p4], %[p4], %[p], %[m4]\n\t"
                "v_cndmask_b32_e64 %[p4], %[p4], %[p3], %[m3]\n\t"
                "v_cndmask_b32_e64 %[p3], %[p3], %[p], %[m3]\n\t"
                "v_cndmask_b32_e64 %[p3], %[p3], %[p2], %[m2]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[p], %[m2]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[p1], %[m1]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p], %[m1]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p0], %[m0]\n\t"
                "v_cndmask_b32_e64 %[p0], %[p0], %[p], %[m0]"
                : [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]),
                  [p0] "+v"(pos[0]), [p1] "+v"(pos[1]), [p2] "+v"(pos[2]), [p3] "+v"(pos[3]), [p4] "+v"(pos[4]),
                  [om] "+v"(outside_min), [t] "=&v"(t),
                  [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4)
                : [x] "v"(d2), [p] "v"(p));
        } else if constexpr (K == 6) {
            // the same network one entry longer (25 VALU): the searches of the linearisation keep SIX neighbours - the 6th distance is
            // what certifies the 5-set against later moves (make_cert)
            unsigned long long m0, m1, m2, m3, m4, m5;
            float t;
            asm("v_cmp_lt_f32_e64 %[m0], %[x], %[d0]\n\t"
                "v_cmp_lt_f32_e64 %[m1], %[x], %[d1]\n\t"
                "v_cmp_lt_f32_e64 %[m2], %[x], %[d2]\n\t"
                "v_cmp_lt_f32_e64 %[m3], %[x], %[d3]\n\t"
                "v_cmp_lt_f32_e64 %[m4], %[x], %[d4]\n\t"
                "v_cmp_lt_f32_e64 %[m5], %[x], %[d5]\n\t"
                "v_max_f32_e32 %[t], %[x], %[d5]\n\t"
                "v_min_f32_e32 %[om], %[om], %[t]\n\t"
                "v_med3_f32 %[d5], %[d4], %[d5], %[x]\n\t"
                "v_med3_f32 %[d4], %[d3], %[d4], %[x]\n\t"
                "v_med3_f32 %[d3], %[d2], %[d3], %[x]\n\t"
                "v_med3_f32 %[d2], %[d1], %[d2], %[x]\n\t"
                "v_med3_f32 %[d1], %[d0], %[d1], %[x]\n\t"
                "v_min_f32_e32 %[d0], %[d0], %[x]\n\t"
                "v_cndmask_b32_e64 %[p5], %[p5], %[p], %[m5]\n\t"
                "v_cndmask_b32_e64 %[p5], %[p5], %[p4], %[m4]\n\t"
                "v_cndmask_b32_e64 %[p4], %[p4], %[p], %[m4]\n\t"
                "v_cndmask_b32_e64 %[p4], %[p4], %[p3], %[m3]\n\t"
                "v_cndmask_b32_e64 %[p3], %[p3], %[p], %[m3]\n\t"
                "v_cndmask_b32_e64 %[p3], %[p3], %[p2], %[m2]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[p], %[m2]\n\t"
                "v_cndmask_b32_e64 %[p2], %[p2], %[p1], %[m1]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p], %[m1]\n\t"
                "v_cndmask_b32_e64 %[p1], %[p1], %[p0], %[m0]\n\t"
                "v_cndmask_b32_e64 %[p0], %[p0], %[p], %[m0]"
                : [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5]),
                  [p0] "+v"(pos[0]), [p1] "+v"(pos[1]), [p2] "+v"(pos[2]), [p3] "+v"(pos[3]), [p4] "+v"(pos[4]), [p5] "+v"(pos[5]),
                  [om] "+v"(outside_min), [t] "=&v"(t),
                  [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5)
                : [x] "v"(d2), [p] "v"(p));
        } else
#endif
        {
            outside_min = fminf(outside_min, fmaxf(d2, d[K - 1]));
            bool c[K];
#pragma unroll
            for (int i = 0; i < K; ++i) c[i] = d2 < d[i];
#pragma unroll
            for (int i = K - 1; i >= 1; --i) {
                pos[i] = c[i - 1] ? pos[i - 1] : (c[i] ? p : pos[i]);
                d[i] = med3f(d[i - 1], d[i], d2);
            }
            pos[0] = c[0] ? p : pos[0];
            d[0] = fminf(d[0], d2);
        }
    }
    DCREG_DEVFN float worst_d2() const { return fmaxf(d[K - 1], fminf(d[K - 1] * infl, cap)); }
    DCREG_DEVFN float dist(int j) const { return d[j]; }
    DCREG_DEVFN bool full() const { return pos[K - 1] != kNoIdx; }
    // a point outside the heap ties with the K-th best: the set may depend on the index tie-break
    DCREG_DEVFN bool boundary_tie() const { return full() && outside_min == d[K - 1]; }
};

// The six smallest of the squared distances pushed, ascending - distances only (lin_search6's start-bound probe): entry i becomes the
// median of (d[i-1], d[i], x), all read before any is written, as in HeapFast::push.  Padding slots push +inf.
struct Top6 {
    static constexpr int K = 6;
    float d[6];
    uint32_t n_eval, n_shell;      // (interface of the heaps; unused)
    DCREG_DEVFN void init(float bound_f) {
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = bound_f;
        n_eval = 0; n_shell = 1;
    }
    DCREG_DEVFN void push(float x, uint32_t /*idx*/, uint32_t /*p*/, bool /*valid*/ = true) {
        d[5] = med3f(d[4], d[5], x); d[4] = med3f(d[3], d[4], x); d[3] = med3f(d[2], d[3], x);
        d[2] = med3f(d[1], d[2], x); d[1] = med3f(d[0], d[1], x); d[0] = fminf(d[0], x);
    }
};

// float32, NOT contracted to FMA: must round exactly like the oracle's / FLANN's plain mul+add chain
DCREG_DEVFN float dist2_nofma(float qx, float qy, float qz, const float4 &c) {
#pragma clang fp contract(off)
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 dxy = f2{qx, qy} - f2{c.x, c.y};       // (x,y) is the register pair a dwordx4 load leaves aligned for v_pk_*
    dxy = dxy * dxy;
    const float dz = qz - c.z;
    float d2 = dxy.x + dxy.y;
    d2 = d2 + dz * dz;
    return d2;
}

// utils.hpp:630-636 pointBodyToGlobal: double arithmetic (separate mul/add, as un-fused x86 code does), float store
DCREG_DEVFN void body_to_global(const PoseArg &P, double px, double py, double pz, float &qx, float &qy, float &qz) {
#pragma clang fp contract(off)
    qx = (float)(P.R[0] * px + P.R[1] * py + P.R[2] * pz + P.t[0]);
    qy = (float)(P.R[3] * px + P.R[4] * py + P.R[5] * pz + P.t[1]);
    qz = (float)(P.R[6] * px + P.R[7] * py + P.R[8] * pz + P.t[2]);
}

// one run of the ring walk: four candidates per trip, their loads issued together (slots past the end read on in the
// sorted array and push +inf)
template <class H, int NB = 1>
DCREG_DEVFN void scan_run(const GridDev &g, uint32_t s, uint32_t e, float qx, float qy, float qz, H &hp);

DCREG_DEVFN int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }   // lo <= hi at every call site
// square root to 1 ulp (v_sqrt_f32) for reach computations that carry a 1e-5 relative safety margin anyway; sqrtf() expands to
// a ~17-instruction correctly-rounded sequence on gfx950
DCREG_DEVFN float sqrt_approx(float x) {
#if DCREG_ON_DEVICE
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

struct RunList;
template <class H, bool SWEEP, bool NOTE>
DCREG_DEVFN void knn_shells(const GridDev &g, RunList &rl, float qx, float qy, float qz, int cx, int cy, int cz,
                                           double fx, double fy, double fz, float bound_f, int max_ring, H &hp);

// Per-thread list of the non-empty x-runs of the 3x3x3 block, kept in LDS ([slot][thread]: conflict-free).
// Surface data leaves most of the 9 (y,z) rows empty, so the list is short (~3 runs) and a run switch in the
// divergent candidate loop costs one ds_read instead of a 9-way register select.
//
// Deferred insertion: while the runs are scanned a candidate is only FILTERED against the lane's
// bound (one compare instead of the 21-instruction sorted insertion) and, if it passes, appended to the lane's pending list;
// the list is inserted into the heap when some lane of the wave is about to run out of room, and at the end.  With a warm
// bound ~6 of a query's ~34 candidates pass, so the insertion network runs ~10 times per wave instead of ~63 (it runs for
// every lane whenever ANY lane has a candidate, which is always).  The pushes reach the heap in scan order, so the result -
// neighbour set, order among ties, the exact-tie flag - is the one the immediate insertion gives (search.hpp knn_search).
constexpr int kPend = 7;
static_assert(kPend >= 4, "a trip parks up to four candidates: the pending list must hold them");
constexpr int kWave = 64;
struct PendEntry { uint32_t d2_bits, pos; };
// One RunList per WAVE ([slot][lane]); a wave's list is private to it, so once its search is over the same LDS serves as
// that wave's staging area for the MFMA reduction of the rows (kernels.hpp) without a block barrier in between.
constexpr int kRowStride = 9;          // doubles per staged row: 8 values + 1 pad (bank-conflict-free 64-bit writes)
// scratch of the wave-cooperative search (team_search6 below): the candidates of ONE query that lie inside its bound, one per lane,
// and the seven best of them in order
struct TeamLds { uint32_t d2[kWave], pos[kWave], idx[kWave]; uint32_t out_d2[8], out_pos[8]; };
struct alignas(16) RunList {
    union {
        struct {
            uint32_t s[9][kWave];
            uint32_t e[9][kWave];
            uint16_t gap2h[9][kWave];     // squared distance from the query to the row's (y,z) slab: upper half of the float, i.e.
                                          // rounded toward zero - a row is never pruned on a distance it does not have
            PendEntry pend[kPend][kWave];
        };
        double stage[kWave * kRowStride];
        TeamLds team;
    };
};
DCREG_DEVFN bool wave_any(bool x) {
#if DCREG_ON_DEVICE
    return __builtin_amdgcn_ballot_w64(x) != 0ull;
#else
    return x;
#endif
}

template <class H>
DCREG_DEVFN void push_point(H &hp, float qx, float qy, float qz, const float4 &c, uint32_t p, bool valid) {
    const float d2 = dist2_nofma(qx, qy, qz, c);
    hp.push(valid ? d2 : __builtin_inff(), __float_as_uint(c.w), p, valid);     // padding slots can never enter
}

// (NB: trips whose loads are requested together - a scan that nothing else overlaps, like the start-bound probe of a far query, pays one
//  memory round trip per NB trips instead of one per trip; the sorted array is padded for the widest batch, kPtsPad)
constexpr int kPtsPad = 16;
constexpr int kProbeBatch = 2;       // lin_search6's start-bound probe (at most 48 points)
template <class H, int NB>
DCREG_DEVFN void scan_run(const GridDev &g, uint32_t s, uint32_t e, float qx, float qy, float qz, H &hp) {
    static_assert(4 * NB <= kPtsPad, "slots past the end of the array read its padding");
    DCREG_STAT(runs);
    for (uint32_t p = s; p < e; p += 4 * NB) {
        float4 c[4 * NB];
        const float4 *cp4 = g.pts + p;          // slots past the end read the array's padding / the next cell: masked below
#pragma unroll
        for (int u = 0; u < 4 * NB; ++u) c[u] = cp4[u];
#pragma unroll
        for (int u = 0; u < 4 * NB; ++u) {
            if (u % 4 == 0 && p + u < e) DCREG_STAT(trips);
            push_point<H>(hp, qx, qy, qz, c[u], p + u, p + u < e);
        }
    }
}

// The same run with DEFERRED insertion (the rows of the sweep: a far query scans hundreds of candidates of which a few per cent can
// still enter once the first rows have tightened the ball): a candidate is filtered against the pruning bound as of the last flush
// - one compare instead of the sorted insertion - and, if it passes, parked in the lane's pending list; the list
// is pushed into the heap when some lane runs out of room and at the end of the run.  Pushes happen in scan order and everything
// that was filtered out is noted as "seen and not kept", so heap, tie flag and 7th-neighbour bound come out as with scan_run.
template <class H, bool NOTE>
DCREG_DEVFN void scan_run_deferred(const GridDev &g, RunList &rl, uint32_t s, uint32_t e, float qx, float qy, float qz, H &hp) {
    DCREG_STAT(runs);
    const int tid = threadIdx.x & (kWave - 1);
    float lim = hp.worst_d2();           // nothing at or beyond it can enter; tightened at every flush
    float om = __builtin_inff();
    int cnt = 0;
    auto flush = [&]() {
        for (int j = 0; j < cnt; ++j) {
            const PendEntry pe = rl.pend[j][tid];
            hp.template push<false>(__uint_as_float(pe.d2_bits), 0u, pe.pos, true);
        }
        cnt = 0;
        lim = hp.worst_d2();
    };
    // Two register sets, the loads of the next trip requested before the present one is consumed (round 6): a far query's rows are a few
    // trips each and nothing else of the wave overlaps them - one memory round trip per trip was most of what the slowest waves of a run's
    // first launches lasted (profiles/r06_ablation.md section 3).  Same candidates in the same order.
    auto take4 = [&](const float4 (&c)[4], uint32_t p) {
        DCREG_STAT(trips);
        if (wave_any(cnt > kPend - 4)) flush();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d2 = dist2_nofma(qx, qy, qz, c[u]);
            const bool valid = p + u < e;
            const bool pass = valid && d2 < lim;
            hp.n_eval += valid ? 1u : 0u;
            if (pass) { rl.pend[cnt][tid] = PendEntry{__float_as_uint(d2), p + u}; ++cnt; }
            if (NOTE) om = fminf(om, (valid && !pass) ? d2 : __builtin_inff());
        }
    };
    auto load4 = [&](float4 (&c)[4], uint32_t p) {
        const float4 *cp4 = g.pts + p;
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = cp4[u];
    };
    float4 ca[4], cb[4];
    if (s < e) load4(ca, s);
    for (uint32_t p = s; p < e; p += 8) {
        const bool second = p + 4 < e;
        if (second) load4(cb, p + 4);
        take4(ca, p);
        if (!second) break;
        if (p + 8 < e) load4(ca, p + 8);
        take4(cb, p + 4);
    }
    flush();
    if (NOTE) hp.note_outside(om);
}

// Exact K nearest neighbours of q among points closer than sqrt(bound) ; returns with the heap filled.
// Ring k covers all cells at Chebyshev distance <= k from the query's cell; after ring k every point
// closer than k*h is in the heap, so the search stops as soon as the K-th best is inside that ball or
// the ball covers the search radius.
// NOTE = false: the candidates the deferred insertion filters out are not noted as "seen and not kept".  Each of them is at or beyond
// the pruning distance of its moment, hence of the end: the 7th-neighbour bound min(outside_min, final pruning distance, bound) does
// not need them.  The boundary-tie test of knn_exact does (a filtered candidate may equal the K-th best where the pruning distance is
// not inflated), so only search6 - which decides ties on the 5th / 6th entries alone - turns it off.
// DEPTH: register sets of the candidate loop's software pipeline, i.e. trips whose loads are in flight while one is consumed (2: what a
// kernel at four waves per SIMD can afford; 4: the instantiations whose launches leave the SIMDs nearly empty and last as long as one
// wave's chain of round trips - the advance pass).  The scan order - and with it every result - does not depend on it.
template <class H, bool SWEEP = false, bool NOTE = true, int DEPTH = 2>
DCREG_DEVFN void knn_search(const GridDev &g, RunList &rl, float qx, float qy, float qz, float bound_f,
                                           int max_ring, H &hp, float infl = 1.f, float cap = __builtin_inff(), bool empty_block = false) {   // max_ring < 0: unbounded
    hp.init(bound_f, infl, cap);
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double lim = (double)max_ring + 1.0;
    if (max_ring >= 0) {
        // bounded search: a query farther than max_ring cells from the grid has no neighbour inside the radius
        if (fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim) return;
    }
    const double big = 6.0e7;     // cell coordinates stay below 2^26 in magnitude: sub-cell indices (x16) and ring arithmetic fit in 32 bits
    const double flx = floor(fmin(fmax(fx, -big), big)), fly = floor(fmin(fmax(fy, -big), big)), flz = floor(fmin(fmax(fz, -big), big));
    const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    if (max_ring < 0) {   // unbounded: enough rings to sweep the whole grid from this cell
        const int ex = max(abs(cx), abs(cx - (nx - 1))), ey = max(abs(cy), abs(cy - (ny - 1))), ez = max(abs(cz), abs(cz - (nz - 1)));
        max_ring = max(ex, max(ey, ez)) + 1;
    }

    // ---- rings 0+1, phase A: the 9 (y,z) rows of the 3x3x3 block, each one contiguous x-run.  Straight-line code: all 18
    // table loads are unconditional and in flight together (a row that is outside the grid or out of reach loads entry 0
    // twice and so yields an empty run), 32-bit cell arithmetic (the table has at most 2^27 entries).  The x-interval of a
    // row is the part of the three cells the ball of radius sqrt(bound) can reach, in SUB-CELLS (GridDev::sx per cell): the
    // finer cut costs nothing in rows or table loads and spares the candidates of the sub-cells beyond the ball.  All reach
    // tests are conservative (the float chain of dist2_nofma can come out below the exact value by a few ulp; the margins
    // here are 1e-5 relative plus 1e-4 of a sub-cell).  Empty runs are dropped when the list is written, nearest rows first.
    const int tid = threadIdx.x & (kWave - 1);
    int nrun = 0;
    // (empty_block, uniform over the wave: the caller knows from the empty-space field that the 27-cell block of every query of the
    // wave holds no point - queries in empty space, all of them: phases A and B have nothing to find)
    if (!empty_block) {
        const float hf = (float)g.h;
        const float frx = (float)(fx - flx), fry = (float)(fy - fly), frz = (float)(fz - flz);
        const float gyl = fry * hf, gyh = (1.f - fry) * hf, gzl = frz * hf, gzh = (1.f - frz) * hf;
        const float gy2[3] = {gyl * gyl * 0.99999f, 0.f, gyh * gyh * 0.99999f}, gz2[3] = {gzl * gzl * 0.99999f, 0.f, gzh * gzh * 0.99999f};
        const int sx = g.sx, nxf = nx * sx;
        const int cxs = cx * sx;                                   // first sub-cell of the query's cell (|cx| <= 6e7, sx <= 16: no overflow)
        const float uf = frx * (float)sx;                          // query position inside its cell, in sub-cells
        const float kx = (float)g.inv_h * (float)sx * 1.00001f;    // metres -> sub-cells, with the relative margin
        const uint32_t unx = (uint32_t)nxf, uny = (uint32_t)ny, sxy = unx * uny;
        const uint32_t base_c = ((uint32_t)cz * uny + (uint32_t)cy) * unx;          // garbage when (cy, cz) is outside: not used then
        const uint32_t yoff[3] = {base_c - unx, base_c, base_c + unx};
        const bool yok[3] = {(uint32_t)(cy - 1) < uny, (uint32_t)cy < uny, (uint32_t)(cy + 1) < uny};
        const bool zok[3] = {(uint32_t)(cz - 1) < (uint32_t)nz, (uint32_t)cz < (uint32_t)nz, (uint32_t)(cz + 1) < (uint32_t)nz};
        // visiting order (dy,dz): centre, 4 edge rows, 4 corner rows
        constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        uint32_t rs[9], re[9];
        float g2s[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float g2 = gy2[DY[r] + 1] + gz2[DZ[r] + 1];
            g2s[r] = g2;
            // sub-cells of this row the ball can reach, clipped to the three cells of the block and to the grid: a tight bound
            // (warm start) trims the run to a few sub-cells, or drops the row
            const float rxf = fminf(sqrt_approx(fmaxf(bound_f - g2, 0.f)) * kx + 1e-4f, 1.0e6f);
            const int lo = max((int)floorf(uf - rxf), -sx), hi = min((int)floorf(uf + rxf), 2 * sx - 1);
            const int x0 = clampi(cxs + lo, 0, nxf), x1 = clampi(cxs + hi + 1, 0, nxf);   // [x0, x1)
            const bool ok = yok[DY[r] + 1] && zok[DZ[r] + 1] && (x1 > x0) && !(g2 > bound_f);
            const uint32_t row = yoff[DY[r] + 1] + (DZ[r] < 0 ? 0u - sxy : (DZ[r] > 0 ? sxy : 0u));
            rs[r] = g.cell_start[ok ? row + (uint32_t)x0 : 0u];
            re[r] = g.cell_start[ok ? row + (uint32_t)x1 : 0u];
            if (ok) { DCREG_STAT(table_loads); DCREG_STAT(table_loads); }
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (re[r] > rs[r]) {
                rl.s[nrun][tid] = rs[r]; rl.e[nrun][tid] = re[r]; rl.gap2h[nrun][tid] = (uint16_t)(__float_as_uint(g2s[r]) >> 16);
                ++nrun;
            }
        }
    }
    // ---- phase B: flattened walk over the runs (the wave iterates max-over-lanes of the total, not the
    // sum of per-row maxima), 4 candidates in flight per trip
    if (!empty_block) {
        int ri = 0;
        uint32_t p = 0, e = 0;
        // deferred insertion state: lim = the lane's filter bound (the heap's K-th best as of the last flush: nothing at or
        // beyond it can enter), om = smallest distance among the candidates filtered out, cnt = pending entries
        float lim = bound_f, om = __builtin_inff();
        int cnt = 0;
        // switch to the next listed row (one per call, no inner loop: a row that the K-th best has meanwhile put out
        // of reach becomes an empty run and costs one idle trip, which is rare once the search is bounded)
        auto next_run = [&]() {
            const float g2 = __uint_as_float((uint32_t)rl.gap2h[ri][tid] << 16);
            const uint32_t s_ = rl.s[ri][tid], e_ = rl.e[ri][tid];
            ++ri;
            const bool keep = !(g2 > (H::kDeferred ? lim : hp.worst_d2()));
            p = keep ? s_ : 0u; e = keep ? e_ : 0u;
            DCREG_STAT(runs);
        };
        // software-pipelined over two register sets, unrolled twice (no copies): the loads of trip t+1 are in flight
        // while trip t is inserted (a third set, two trips ahead: +2.5 % at 100 k points, -8 % at 1 M where the extra
        // registers cost a wave of occupancy).  Slots past the end of a run are masked out.
        constexpr int W = 4;
        struct Slot { float4 c[W]; uint32_t cp, ce; bool live; };
        bool have = nrun > 0;
        if (have) next_run();
        auto fetch = [&](Slot &sl) {
            sl.live = have; sl.cp = p; sl.ce = e;
            if (have) {
                DCREG_STAT(trips);
                // one address, four loads at immediate offsets: slots past the end of the run read the next points of the
                // sorted array (it is padded by kPtsPad entries) and are masked out by their position when consumed
                const float4 *cp4 = g.pts + p;
#pragma unroll
                for (int u = 0; u < W; ++u) sl.c[u] = cp4[u];
                p += W;
                if (p >= e) {
                    have = ri < nrun;
                    if (have) next_run();
                }
            }
        };
        auto flush = [&]() {
            if constexpr (H::kDeferred) {
                for (int j = 0; j < cnt; ++j) {
                    const PendEntry pe = rl.pend[j][tid];
                    hp.template push<false>(__uint_as_float(pe.d2_bits), 0u, pe.pos, true);
                }
                cnt = 0;
                lim = hp.worst_d2();
            }
        };
        auto consume = [&](const Slot &sl) {
            if constexpr (H::kDeferred) {
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    const float d2 = dist2_nofma(qx, qy, qz, sl.c[u]);
                    const bool valid = sl.cp + u < sl.ce;
                    const bool pass = valid && d2 < lim;
                    hp.n_eval += valid ? 1u : 0u;
                    if (pass) { rl.pend[cnt][tid] = PendEntry{__float_as_uint(d2), sl.cp + u}; ++cnt; }
                    if (NOTE) om = fminf(om, (valid && !pass) ? d2 : __builtin_inff());
                }
            } else {
#pragma unroll
                for (int u = 0; u < W; ++u) push_point<H>(hp, qx, qy, qz, sl.c[u], sl.cp + u, sl.cp + u < sl.ce);
            }
        };
        Slot S[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k) fetch(S[k]);
        bool go = DEPTH > 1 ? S[0].live : have;
        while (go) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                if (H::kDeferred && wave_any(cnt > kPend - W)) flush();      // room for the next W candidates in every lane
                fetch(S[(k + DEPTH - 1) % DEPTH]); consume(S[k]);
                if (!S[(k + 1) % DEPTH].live) { go = false; break; }
            }
        }
        if constexpr (H::kDeferred) {
            flush();
            hp.note_outside(om);
        }
    }
    knn_shells<H, SWEEP, NOTE>(g, rl, qx, qy, qz, cx, cy, cz, fx, fy, fz, bound_f, max_ring, hp);
}

// Rings k >= 2 around cell (cx,cy,cz) (sparse neighbourhoods, cloud borders, large misalignment), global loads.
// Ring kk = the surface of the cube of Chebyshev radius kk: its z faces and y faces are swept as (y,z) rows with the x-run of each
// row trimmed to the cells the K-th-best ball can still reach (kd-tree style pruning on the grid); of the rows in between only the
// two END CELLS belong to the ring (the x faces).  Measured on the GPU (instrumented kernel, scripts/c4_first_iters.py, C4's first
// iteration, the slowest 1 % of the waves): of 857 k ring-walk cycles per wave 17 % were candidate scans, 8 % waits for table entries
// and 76 % the row loop itself - round 1 walked all (2kk+1)^2 rows of every ring with the full x-range arithmetic (sqrt, two double
// floors) although (2kk-1)^2 of them can only contribute an end cell.  So:
//   * x faces: a cell (cx -+ kk, y, z) is reachable iff gx^2 + gy^2 + gz^2 <= K-th best with gx fixed for the ring - two compares per
//     row, and the (y,z) loops stop at the radius the ball still has at that x distance;
//   * faces whose cell layer lies beyond the K-th best are not entered;
//   * the loop bounds are uniform over the wave's lanes up to that radius, so the 64 queries stay in lock-step (a face walk with
//     per-lane iteration order, empty-space culling per face and centre-out rows visited fewer rows
//     and was 25 % slower; batching the table loads of four rows, a flattened collect-then-scan walk and a 2x2x2 block occupancy
//     bitmap that spares 88 % of the table lookups were all slower too: profiles/r02_ablation.md).
template <class H, bool SWEEP, bool NOTE>
DCREG_DEVFN void knn_shells(const GridDev &g, RunList &rl, float qx, float qy, float qz, int cx, int cy, int cz,
                                           double fx, double fy, double fz, float bound_f, int max_ring, H &hp) {
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const float hf = (float)g.h;
    bool done;
    {   // what ring 1's test would say, before anything is loaded: an aligned query is done after the centre block (its K-th best lies
        // within one cell edge), and when that holds for the whole wave the walk costs nothing - not even the field byte below
        const double safe = g.h * (1.0 - 1e-9);
        const double safe2 = safe * safe * (1.0 - 1e-6);
        done = max_ring <= 1 || (double)hp.worst_d2() <= safe2 || safe2 >= (double)bound_f;
        if (!wave_any(!done)) return;
    }
    // empty-space skip: if the nearest occupied cell is f cells away (Chebyshev), rings 1 .. f-1 hold no point
    int k0 = 1;
    if (!SWEEP && g.gap && cx >= 0 && cx < nx && cy >= 0 && cy < ny && cz >= 0 && cz < nz) {
        const int f = min((int)g.gap[((int64_t)cz * ny + cy) * nx + cx], g.gap_cap + 1);
        k0 = max(1, f - 1);
    }
    auto lookup_scan = [&](int64_t c0, int64_t c1) {
        DCREG_STAT(table_loads); DCREG_STAT(table_loads); DCREG_STAT(faces);
        const uint32_t s_ = g.cell_start[c0], e_ = g.cell_start[c1];
        if constexpr (SWEEP && H::kDeferred) scan_run_deferred<H, NOTE>(g, rl, s_, e_, qx, qy, qz, hp);
        else scan_run<H>(g, s_, e_, qx, qy, qz, hp);
    };
    // slab distance (metres, float) from the query to cell index c along an axis (cq = the query's cell, fr = the query's position
    // inside that cell, in cells): single precision - every use carries a 1e-5 relative safety factor against 1e-7 of rounding
    const float frx = (float)(fx - (double)cx), fry = (float)(fy - (double)cy), frz = (float)(fz - (double)cz);
    const float inv_hf = (float)g.inv_h;
    // x in sub-cells (GridDev::sx per cell): rows are trimmed to the sub-cell, the x faces are whole cells
    const int sx = g.sx, nxf = nx * sx;
    const int cxs = cx * sx;                           // |cx| <= 6e7 (knn_search), sx <= 16: no overflow
    const float uf = frx * (float)sx, inv_hfs = inv_hf * (float)sx;
    auto slab = [&](int c, int cq, float fr) -> float {
        return c < cq ? ((float)(cq - c - 1) + fr) * hf : (c > cq ? ((float)(c - cq) - fr) * hf : 0.f);
    };
    // one (y,z) row of a z or y face: x-run [cx-kk, cx+kk] trimmed to the ball (conservative), then scanned
    auto face_row = [&](int y, int z, float dyz, int kk, int dz, int dy) {
        DCREG_STAT(rows);
        const float w = hp.worst_d2();
        if (dyz > w) return;
        // sub-cells of the row the ball still reaches, relative to the query's cell (conservative: 1e-5 relative + 1e-4 of a sub-cell)
        const float xr_c = fminf((sqrt_approx(w - dyz) * 1.00001f) * inv_hfs + 1e-4f, 1.0e6f);
        const int dlo = (int)floorf(uf - xr_c), dhi = (int)floorf(uf + xr_c);           // sub-cell offsets from cxs
        const int kks = min(kk, 1 << 26) * sx;
        const int x0 = max(cxs + max(dlo, -kks), 0), x1 = min(cxs + min(dhi, kks + sx - 1), nxf - 1) + 1;
        if (x1 <= x0) return;
        const int64_t row = ((int64_t)z * ny + y) * nxf;
        DCREG_TRACE(kk, dz, dy, 0, (g.cell_start[row + x1] - g.cell_start[row + x0] + 3u) / 4u);
        lookup_scan(row + x0, row + x1);
    };
    // offsets of the rows a ball of squared radius r2 (in the plane of a face) can reach along an axis on which the query sits at
    // `fr` inside its cell: a row at offset +o is (o - fr) cells away, one at -o is (o - 1 + fr) cells away (conservative by
    // 1e-5 relative + 1e-4 of a cell); clipped to +-cap.  The loops below are bounded per lane; the wave runs to the widest.
    auto reach = [&](float r2, float fr, int cap, int &lo, int &hi) {
        const float rc = fminf(sqrt_approx(fmaxf(r2, 0.f)) * 1.00001f * inv_hf + 1e-4f, 1.0e6f);
        lo = -min(cap, (int)floorf(rc + 1.f - fr));
        hi = min(cap, (int)floorf(rc + fr));
        if (r2 < 0.f) { lo = 1; hi = 0; }
    };
    // ---- row sweep (SWEEP: the searches of the linearisation, whose ball is bounded by the search radius) instead of the ring walk: ring kk costs a table lookup for every (y,z) row the ball reaches in that shell - O(r^2) lookups per ring, nearly
    // all of them on empty rows - while the points it is after sit in the few rows where the ball touches a surface.  The row
    // occupancy words name those rows directly: per z layer of the ball one OR over the x blocks the ball spans gives the mask of
    // occupied rows, and only those are looked up (x-run trimmed to the ball as in face_row) and scanned.  Layers are visited
    // centre-out, every bound is taken from the K-th best as it stands (pruning by worst_d2() only, like the ring walk: the
    // certificate's "never looked at => at least worst_d2() away" holds), and the sweep covers the whole ball, so the lane is
    // finished afterwards.  The 3x3 rows of the centre block were scanned over their three centre cells by phase A: only the parts
    // left and right of those are scanned here.
    if constexpr (SWEEP) {
        const float w0 = hp.worst_d2();
        if (!done) {
            const int nxb = g.nxb, nyw = g.nyw, cap = 1 << 24;      // (the bound is finite; offsets stay far inside 32 bits: |cx| <= 6e7)
            auto sweep_row = [&](int y, int z, float gz) {
                DCREG_STAT(rows);
                const float gy = slab(y, cy, fry);
                const float dyz = (gy * gy + gz * gz) * 0.99999f;
                const float w = hp.worst_d2();
                if (dyz > w) return;
                const float xr_c = fminf((sqrt_approx(w - dyz) * 1.00001f) * inv_hfs + 1e-4f, 1.0e6f);
                const int dlo = (int)floorf(uf - xr_c), dhi = (int)floorf(uf + xr_c);           // sub-cell offsets from cxs
                const int x0 = max(cxs + dlo, 0), x1 = min(cxs + dhi, nxf - 1) + 1;
                if (x1 <= x0) return;
                const int64_t row = ((int64_t)z * ny + y) * nxf;
                if (abs(y - cy) <= 1 && abs(z - cz) <= 1) {
                    const int l1 = min(x1, cxs - sx), r0 = max(x0, cxs + 2 * sx);
                    if (l1 > x0) lookup_scan(row + x0, row + l1);
                    if (x1 > r0) lookup_scan(row + r0, row + x1);
                } else {
                    lookup_scan(row + x0, row + x1);
                }
            };
            int zlo, zhi;
            reach(w0, frz, cap, zlo, zhi);
            const int zmax = max(-zlo, zhi);
            for (int i = 0; i <= 2 * zmax; ++i) {
                const int adz = (i + 1) >> 1, dz = (i & 1) ? -adz : adz;
                const float w = hp.worst_d2();
                { const float nearer = (float)max(adz - 1, 0) * hf; if (nearer * nearer * 0.99999f > w) break; }   // both layers at this |dz| and beyond are out
                const int z = cz + dz;
                if (z < 0 || z >= nz) continue;
                const float gz = slab(z, cz, frz);
                const float rem = w - gz * gz * 0.99999f;
                if (rem < 0.f) continue;
                int ylo, yhi, xlo, xhi;
                reach(rem, fry, cap, ylo, yhi);
                reach(rem, frx, cap, xlo, xhi);
                const int y0 = max(cy + ylo, 0), y1 = min(cy + yhi, ny - 1);
                const int b0 = max(cx + xlo, 0) >> 4, b1 = min(cx + xhi, nx - 1) >> 4;
                if (y1 < y0 || b1 < b0) continue;
                for (int yw = y0 >> 5; yw <= (y1 >> 5); ++yw) {
                    uint32_t m = 0;
                    const uint32_t *mw = g.ymask + ((int64_t)z * nxb + b0) * nyw + yw;
                    for (int b = b0; b <= b1; ++b, mw += nyw) { m |= *mw; DCREG_STAT(table_loads); }
                    const int lo = max(y0 - (yw << 5), 0), hi = min(y1 - (yw << 5), 31);
                    m &= (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo);
                    while (m) {
                        const int bit = __builtin_ctz(m);
                        m &= m - 1;
                        sweep_row((yw << 5) + bit, z, gz);
                    }
                }
            }
        }
        return;
    }
    for (int k = k0; k < max_ring; ++k) {
        // after ring k: every point within k*h (minus a rounding guard) has been seen
        const double safe = (double)k * g.h * (1.0 - 1e-9);
        const double safe2 = safe * safe * (1.0 - 1e-6);
        if ((double)hp.worst_d2() <= safe2) return;             // K-th best already inside the covered ball
        if (safe2 >= (double)bound_f) return;                   // covered ball contains the search radius
        const int kk = k + 1;                                   // scan shell kk
        hp.n_shell = (uint32_t)kk;
        // the six faces: in the grid, within the K-th best, and not provably empty.  The six field bytes are requested together
        // (one wait per ring instead of six dependent ones)
        const int fz_[2] = {cz - kk, cz + kk}, fy_[2] = {cy - kk, cy + kk}, fx_[2] = {cx - kk, cx + kk};
        float d2f[6];
        bool live[6];
        int need[6], gv[6];
        const float w_ring = hp.worst_d2();
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const int axis = f >> 1, side = f & 1;
            const int layer = axis == 0 ? fz_[side] : (axis == 1 ? fy_[side] : fx_[side]);
            const int nq = axis == 0 ? nz : (axis == 1 ? ny : nx);
            const float gl = axis == 0 ? slab(layer, cz, frz) : (axis == 1 ? slab(layer, cy, fry) : slab(layer, cx, frx));
            d2f[f] = gl * gl * 0.99999f;
            live[f] = layer >= 0 && layer < nq && !(d2f[f] > w_ring);
            // the cap the ball cuts out of the face reaches floor(rho / h) + 1 cells from the face cell under the query
            const float rho = sqrt_approx(fmaxf(w_ring - d2f[f], 0.f)) * 1.00001f + 1e-6f * hf;
            const double rho_c = (double)rho * g.inv_h;         // may be astronomically large (unbounded searches): compare before converting
            need[f] = rho_c >= (double)kk ? kk : min(kk, (int)rho_c + 1);
            const int xq = clampi(axis == 2 ? layer : cx, 0, nx - 1), yq = clampi(axis == 1 ? layer : cy, 0, ny - 1),
                      zq = clampi(axis == 0 ? layer : cz, 0, nz - 1);
            gv[f] = (g.gap && live[f]) ? (int)g.gap[((int64_t)zq * ny + yq) * nx + xq] : 0;
            if (g.gap && live[f]) DCREG_STAT(table_loads);
        }
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const int free_r = gv[f] == 255 ? g.gap_cap + 1 : gv[f];     // every cell closer (Chebyshev) than free_r to that cell is empty
            if (g.gap && live[f] && need[f] < free_r) { live[f] = false; DCREG_STAT(face_skips); }
        }
        // ---- z faces: layers z = cz -+ kk, rows y = cy-kk .. cy+kk
        for (int sz = 0; sz < 2; ++sz) {
            const int z = fz_[sz];
            const bool zin = live[sz];
            if (!wave_any(zin)) continue;
            const float gz = slab(z, cz, frz);
            int ylo, yhi;
            reach(zin ? hp.worst_d2() - d2f[sz] : -1.f, fry, kk, ylo, yhi);
            for (int dy = ylo; dy <= yhi; ++dy) {
                const int y = cy + dy;
                if (!(zin && y >= 0 && y < ny)) continue;
                const float gy = slab(y, cy, fry);
                face_row(y, z, (gy * gy + gz * gz) * 0.99999f, kk, sz ? kk : -kk, dy);
            }
        }
        // ---- y faces: layers y = cy -+ kk, rows z = cz-kk+1 .. cz+kk-1
        for (int sy = 0; sy < 2; ++sy) {
            const int y = fy_[sy];
            const bool yin = live[2 + sy];
            if (!wave_any(yin)) continue;
            const float gy = slab(y, cy, fry);
            int zlo, zhi;
            reach(yin ? hp.worst_d2() - d2f[2 + sy] : -1.f, frz, kk - 1, zlo, zhi);
            for (int dz = zlo; dz <= zhi; ++dz) {
                const int z = cz + dz;
                if (!(yin && z >= 0 && z < nz)) continue;
                const float gz = slab(z, cz, frz);
                face_row(y, z, (gy * gy + gz * gz) * 0.99999f, kk, dz, sy ? kk : -kk);
            }
        }
        // ---- x faces: the end cells (cx -+ kk, y, z) of the rows in between.  gx is fixed for the ring: the cell is reachable iff
        // gx^2 + gy^2 + gz^2 <= K-th best, and rows farther than the ball's radius at that x distance need not be visited at all
        {
            const int xa = fx_[0], xb = fx_[1];
            const float gxa2 = live[4] ? d2f[4] : __builtin_inff(), gxb2 = live[5] ? d2f[5] : __builtin_inff();
            const float w0 = hp.worst_d2();
            const float rmax = fmaxf(w0 - gxa2, w0 - gxb2);     // squared (y,z) radius the ball still has on the nearer x face
            int zlo, zhi, ylo, yhi;
            reach(rmax, frz, kk - 1, zlo, zhi);
            reach(rmax, fry, kk - 1, ylo, yhi);
            for (int dz = zlo; dz <= zhi; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z >= nz) continue;
                const float gz = slab(z, cz, frz);
                const float gz2 = gz * gz;
                if (fminf(gxa2, gxb2) + gz2 * 0.99999f > hp.worst_d2()) continue;
                for (int dy = ylo; dy <= yhi; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y >= ny) continue;
                                DCREG_STAT(rows);
                    const float gy = slab(y, cy, fry);
                    const float dyz = (gy * gy + gz2) * 0.99999f;
                    const float w = hp.worst_d2();
                    const int64_t row = ((int64_t)z * ny + y) * nxf, ra = row + (int64_t)xa * sx, rb = row + (int64_t)xb * sx;
                    if (gxa2 + dyz <= w) { DCREG_TRACE(kk, dz, dy, 0, (g.cell_start[ra + sx] - g.cell_start[ra] + 3u) / 4u); lookup_scan(ra, ra + sx); }
                    if (gxb2 + dyz <= hp.worst_d2()) { DCREG_TRACE(kk, dz, dy, 1, (g.cell_start[rb + sx] - g.cell_start[rb] + 3u) / 4u); lookup_scan(rb, rb + sx); }
                }
            }
        }
    }
}


// ---------------------------------------------------------------- exact K-NN of one query (fast path + fallback)
// Runs the 32-bit-key search; if (and only if) a point outside the result ties with the K-th best distance,
// re-runs the exact 64-bit-key search for this lane.  Output: neighbours in canonical (d2, idx) order.
template <int K>
struct KnnResult {
    float d2[K];
    uint32_t idx[K];     // original target index
    float4 pt[K];        // neighbour coordinates (w = idx bits)
    uint32_t pos[K];     // position in the sorted target (kNoIdx = none)
    bool full;           // K neighbours found under the bound
    uint32_t n_eval, n_shell;
};

template <int K, bool SWEEP = false>
DCREG_DEVFN void knn_exact(const GridDev &g, RunList &rl, float qx, float qy, float qz, float bound_f, int max_ring,
                                          KnnResult<K> &res) {
    uint32_t pos[K];
    {
        HeapFast<K> hf;
        knn_search<HeapFast<K>, SWEEP>(g, rl, qx, qy, qz, bound_f, max_ring, hf);
        res.full = hf.full();
        res.n_eval = hf.n_eval; res.n_shell = hf.n_shell;
#pragma unroll
        for (int j = 0; j < K; ++j) { res.d2[j] = hf.d[j]; pos[j] = hf.pos[j]; }
        if (hf.boundary_tie()) {                       // rare (exactly-equal float distances): exact redo
            HeapExact<K> he;
            knn_search<HeapExact<K>, SWEEP>(g, rl, qx, qy, qz, bound_f, max_ring, he);
            res.n_eval += he.n_eval;
#pragma unroll
            for (int j = 0; j < K; ++j) { res.d2[j] = he.dist(j); pos[j] = he.pos[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool ok = pos[j] != kNoIdx;
        res.pos[j] = pos[j];
        res.pt[j] = ok ? g.pts[pos[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
        res.idx[j] = ok ? __float_as_uint(res.pt[j].w) : kNoIdx;
        if (!ok) res.d2[j] = __builtin_inff();
    }
    // canonical order among equal distances (lower index first); entries are already sorted by d2
    bool any_eq = false;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) any_eq |= (res.d2[j] == res.d2[j + 1]) && res.idx[j + 1] != kNoIdx;
    if (any_eq) {
#pragma unroll
        for (int a = 0; a + 1 < K; ++a)
#pragma unroll
            for (int b = 0; b + 1 < K - a; ++b) {
                const bool sw = res.d2[b] == res.d2[b + 1] && res.idx[b] > res.idx[b + 1];
                const uint32_t ia = res.idx[b], ib = res.idx[b + 1];
                const float4 pa = res.pt[b], pb = res.pt[b + 1];
                res.idx[b] = sw ? ib : ia; res.idx[b + 1] = sw ? ia : ib;
                res.pt[b] = sw ? pb : pa; res.pt[b + 1] = sw ? pa : pb;
            }
    }
}

// ---------------------------------------------------------------- 5x3 column-pivoted Householder QR
// Restates Eigen 3.3.7 ColPivHouseholderQR::compute + solve (icp_test_runner.cpp:1747) for [q_j] x = -1,
// including the nonzeroPivots() truncation that decides rank-deficient (coplanar-with-origin / constant-
// zero column) neighbourhoods.  Columns are swapped with selects so everything stays in registers.
DCREG_DEVFN void swap_col(double (&a)[5], double (&b)[5], bool doit) {
#pragma unroll
    for (int i = 0; i < 5; ++i) { const double ta = a[i], tb = b[i]; a[i] = doit ? tb : ta; b[i] = doit ? ta : tb; }
}
DCREG_DEVFN void swap_d(double &a, double &b, bool doit) { const double ta = a, tb = b; a = doit ? tb : ta; b = doit ? ta : tb; }
DCREG_DEVFN void swap_i(int &a, int &b, bool doit) { const int ta = a, tb = b; a = doit ? tb : ta; b = doit ? ta : tb; }

template <int KCOL>
DCREG_DEVFN void householder_step(double (&c0)[5], double (&c1)[5], double (&c2)[5], double (&tau)[3],
                                                 double (&nu)[3], double (&nd)[3]) {
    // acts on column KCOL (rows KCOL..4) and updates the trailing columns; c0,c1,c2 are the CURRENT columns
    double(&ck)[5] = (KCOL == 0) ? c0 : (KCOL == 1 ? c1 : c2);
    double tail = 0.0;
#pragma unroll
    for (int i = KCOL + 1; i < 5; ++i) tail += ck[i] * ck[i];
    const double a0 = ck[KCOL];
    double beta, t;
    if (tail <= 2.2250738585072014e-308) {
        t = 0.0; beta = a0;
#pragma unroll
        for (int i = KCOL + 1; i < 5; ++i) ck[i] = 0.0;
    } else {
        beta = sqrt(a0 * a0 + tail);
        if (a0 >= 0.0) beta = -beta;
        const double inv_den = 1.0 / (a0 - beta);      // one reciprocal + multiplies (fp64 division is ~11 instructions)
#pragma unroll
        for (int i = KCOL + 1; i < 5; ++i) ck[i] = ck[i] * inv_den;
        t = (beta - a0) / beta;
    }
    tau[KCOL] = t;
    ck[KCOL] = beta;
#pragma unroll
    for (int j = KCOL + 1; j < 3; ++j) {
        double(&cj)[5] = (j == 1) ? c1 : c2;
        if (t != 0.0) {
            double tmp = cj[KCOL];
#pragma unroll
            for (int i = KCOL + 1; i < 5; ++i) tmp += ck[i] * cj[i];
            cj[KCOL] -= t * tmp;
#pragma unroll
            for (int i = KCOL + 1; i < 5; ++i) cj[i] -= t * ck[i] * tmp;
        }
        if (nu[j] != 0.0) {   // LAPACK norm downdate (lawn176), as Eigen does
            double tt = fabs(cj[KCOL]) / nu[j];
            tt = (1.0 + tt) * (1.0 - tt);
            tt = tt < 0.0 ? 0.0 : tt;
            const double ratio = nu[j] / nd[j];
            if (tt * ratio * ratio <= 1.4901161193847656e-08) {
                double s = 0.0;
#pragma unroll
                for (int i = KCOL + 1; i < 5; ++i) s += cj[i] * cj[i];
                nd[j] = nu[j] = sqrt(s);
            } else {
                nu[j] *= sqrt(tt);
            }
        }
    }
}

// returns x (plane coefficients, unnormalised); Q row j = neighbour j
DCREG_DEVFN void plane_fit_qr(const double (&qx)[5], const double (&qy)[5], const double (&qz)[5], double (&x)[3]) {
    double c0[5], c1[5], c2[5], tau[3], nu[3], nd[3];
    int p0 = 0, p1 = 1, p2 = 2;
#pragma unroll
    for (int i = 0; i < 5; ++i) { c0[i] = qx[i]; c1[i] = qy[i]; c2[i] = qz[i]; }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { s0 += c0[i] * c0[i]; s1 += c1[i] * c1[i]; s2 += c2[i] * c2[i]; }
    nu[0] = nd[0] = sqrt(s0); nu[1] = nd[1] = sqrt(s1); nu[2] = nd[2] = sqrt(s2);
    const double mx = fmax(nu[0], fmax(nu[1], nu[2]));
    const double eps = 2.220446049250313e-16;
    const double thr_helper = (mx * eps) * (mx * eps) / 5.0;
    int nz = 3;
    // k = 0
    {
        const bool b1 = nu[1] > nu[0], b2 = nu[2] > (b1 ? nu[1] : nu[0]);
        const double big = b2 ? nu[2] : (b1 ? nu[1] : nu[0]);
        if (big * big < thr_helper * 5.0) nz = 0;
        const bool sw1 = b1 && !b2, sw2 = b2;
        swap_col(c0, c1, sw1); swap_d(nu[0], nu[1], sw1); swap_d(nd[0], nd[1], sw1); swap_i(p0, p1, sw1);
        swap_col(c0, c2, sw2); swap_d(nu[0], nu[2], sw2); swap_d(nd[0], nd[2], sw2); swap_i(p0, p2, sw2);
        householder_step<0>(c0, c1, c2, tau, nu, nd);
    }
    // k = 1
    {
        const bool b2 = nu[2] > nu[1];
        const double big = b2 ? nu[2] : nu[1];
        if (nz == 3 && big * big < thr_helper * 4.0) nz = 1;
        swap_col(c1, c2, b2); swap_d(nu[1], nu[2], b2); swap_d(nd[1], nd[2], b2); swap_i(p1, p2, b2);
        householder_step<1>(c0, c1, c2, tau, nu, nd);
    }
    // k = 2
    {
        if (nz == 3 && nu[2] * nu[2] < thr_helper * 3.0) nz = 2;
        householder_step<2>(c0, c1, c2, tau, nu, nd);
    }
    // solve: c = Q^T rhs (first nz reflectors), back-substitute the leading nz x nz triangle
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    if (nz > 0 && tau[0] != 0.0) {
        double tmp = c[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) tmp += c0[i] * c[i];
        c[0] -= tau[0] * tmp;
#pragma unroll
        for (int i = 1; i < 5; ++i) c[i] -= tau[0] * c0[i] * tmp;
    }
    if (nz > 1 && tau[1] != 0.0) {
        double tmp = c[1];
#pragma unroll
        for (int i = 2; i < 5; ++i) tmp += c1[i] * c[i];
        c[1] -= tau[1] * tmp;
#pragma unroll
        for (int i = 2; i < 5; ++i) c[i] -= tau[1] * c1[i] * tmp;
    }
    if (nz > 2 && tau[2] != 0.0) {
        double tmp = c[2];
#pragma unroll
        for (int i = 3; i < 5; ++i) tmp += c2[i] * c[i];
        c[2] -= tau[2] * tmp;
    }
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    // R = [c0[0] c1[0] c2[0]; 0 c1[1] c2[1]; 0 0 c2[2]]
    if (nz > 2) y2 = c[2] / c2[2];
    if (nz > 1) y1 = (c[1] - (nz > 2 ? c2[1] * y2 : 0.0)) / c1[1];
    if (nz > 0) y0 = (c[0] - (nz > 1 ? c1[0] * y1 : 0.0) - (nz > 2 ? c2[0] * y2 : 0.0)) / c0[0];
    // x[perm[i]] = y[i]
    x[0] = (p0 == 0) ? y0 : ((p1 == 0) ? y1 : y2);
    x[1] = (p0 == 1) ? y0 : ((p1 == 1) ? y1 : y2);
    x[2] = (p0 == 2) ? y0 : ((p1 == 2) ? y1 : y2);
}

// ---------------------------------------------------------------- the same plane fit with fewer instructions
// fp64 division and square root cost ~14 instructions each on gfx950 (v_div_scale x2, v_rcp, Newton steps, v_div_fmas,
// v_div_fixup); plane_fit_qr above spends ~17 divisions and ~12 square roots, two thirds of them in the LAPACK-style
// norm downdating that only serves the pivot choice.  This variant computes the SAME factorisation (same pivot rule on the
// remaining column norms, same nonzeroPivots() truncation, same basic solution) with
//   * remaining column norms recomputed directly (squared, no root, no division) - they differ from the downdated
//     estimates by rounding only, so the pivot order can differ only between columns whose norms agree to ~1e-8, where both
//     orders give the same solution to rounding;
//   * the reflector applied un-normalised, H = I - u u^T / (beta (beta - a0)), u = [a0 - beta, tail]: ONE reciprocal per
//     step, which also yields 1 / R_kk = (beta - a0) / (beta (beta - a0)) for the back substitution;
//   * reciprocal and square root by v_rcp_f64 / v_rsq_f64 + Newton steps without the IEEE corner-case scaffolding (the
//     operands are coordinates in metres: no denormals, no overflow).
// Result: the plane of plane_fit_qr to a few ulp (tests: normals / residuals / weights of the fixture, incl. its 751
// rank-2 neighbourhoods, and of the synthetic scenes against the oracle; gate flags bit-exact).
DCREG_DEVFN double fast_rcp(double x) {
#if DCREG_ON_DEVICE
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / x;
#endif
}
DCREG_DEVFN double fast_sqrt(double x) {      // x >= 0, finite
#if DCREG_ON_DEVICE
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
#else
    return sqrt(x);
#endif
}

template <int KCOL>
DCREG_DEVFN void householder_fast(double (&c0)[5], double (&c1)[5], double (&c2)[5], double (&uk)[3], double (&coef)[3],
                                  double (&rdiag)[3], double (&rinv)[3]) {
    double(&ck)[5] = (KCOL == 0) ? c0 : (KCOL == 1 ? c1 : c2);
    double tail = 0.0;
#pragma unroll
    for (int i = KCOL + 1; i < 5; ++i) tail += ck[i] * ck[i];
    const double a0 = ck[KCOL];
    if (tail <= 2.2250738585072014e-308) {        // Eigen makeHouseholder: no reflection, beta = c0
        uk[KCOL] = 0.0; coef[KCOL] = 0.0; rdiag[KCOL] = a0;
        rinv[KCOL] = fast_rcp(a0);                 // a0 == 0 only for a pivot the truncation has already dropped
#pragma unroll
        for (int i = KCOL + 1; i < 5; ++i) ck[i] = 0.0;
        return;
    }
    double beta = fast_sqrt(a0 * a0 + tail);
    if (a0 >= 0.0) beta = -beta;
    const double u = a0 - beta;                    // |u| >= |beta| > 0
    const double c = fast_rcp(beta * (beta - a0)); // = 2 / (u^T u) > 0
    uk[KCOL] = u; coef[KCOL] = c; rdiag[KCOL] = beta;
    rinv[KCOL] = c * (beta - a0);                  // 1 / beta
#pragma unroll
    for (int j = KCOL + 1; j < 3; ++j) {
        double(&cj)[5] = (j == 1) ? c1 : c2;
        double w = u * cj[KCOL];
#pragma unroll
        for (int i = KCOL + 1; i < 5; ++i) w += ck[i] * cj[i];
        w *= c;
        cj[KCOL] -= w * u;
#pragma unroll
        for (int i = KCOL + 1; i < 5; ++i) cj[i] -= w * ck[i];
    }
}

DCREG_DEVFN void plane_fit_qr_fast(const double (&qx)[5], const double (&qy)[5], const double (&qz)[5], double (&x)[3]) {
    double c0[5], c1[5], c2[5], uk[3], coef[3], rdiag[3], rinv[3];
    int p0 = 0, p1 = 1, p2 = 2;
#pragma unroll
    for (int i = 0; i < 5; ++i) { c0[i] = qx[i]; c1[i] = qy[i]; c2[i] = qz[i]; }
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;          // squared norms of the remaining part of each column
#pragma unroll
    for (int i = 0; i < 5; ++i) { n0 += c0[i] * c0[i]; n1 += c1[i] * c1[i]; n2 += c2[i] * c2[i]; }
    const double mx2 = fmax(n0, fmax(n1, n2));
    const double eps = 2.220446049250313e-16;
    const double thr_helper = mx2 * (eps * eps) / 5.0;
    int nz = 3;
    {   // k = 0
        const bool b1 = n1 > n0, b2 = n2 > (b1 ? n1 : n0);
        const double big = b2 ? n2 : (b1 ? n1 : n0);
        if (big < thr_helper * 5.0) nz = 0;
        const bool sw1 = b1 && !b2, sw2 = b2;
        swap_col(c0, c1, sw1); swap_i(p0, p1, sw1);
        swap_col(c0, c2, sw2); swap_i(p0, p2, sw2);
        householder_fast<0>(c0, c1, c2, uk, coef, rdiag, rinv);
    }
    {   // k = 1
        n1 = 0.0; n2 = 0.0;
#pragma unroll
        for (int i = 1; i < 5; ++i) { n1 += c1[i] * c1[i]; n2 += c2[i] * c2[i]; }
        const bool b2 = n2 > n1;
        const double big = b2 ? n2 : n1;
        if (nz == 3 && big < thr_helper * 4.0) nz = 1;
        swap_col(c1, c2, b2); swap_i(p1, p2, b2);
        householder_fast<1>(c0, c1, c2, uk, coef, rdiag, rinv);
    }
    {   // k = 2
        n2 = 0.0;
#pragma unroll
        for (int i = 2; i < 5; ++i) n2 += c2[i] * c2[i];
        if (nz == 3 && n2 < thr_helper * 3.0) nz = 2;
        householder_fast<2>(c0, c1, c2, uk, coef, rdiag, rinv);
    }
    // c = Q^T rhs (first nz reflectors; a reflector with coef == 0 is the identity), then the leading nz x nz triangle
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    if (nz > 0) {
        double w = uk[0] * c[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) w += c0[i] * c[i];
        w *= coef[0];
        c[0] -= w * uk[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) c[i] -= w * c0[i];
    }
    if (nz > 1) {
        double w = uk[1] * c[1];
#pragma unroll
        for (int i = 2; i < 5; ++i) w += c1[i] * c[i];
        w *= coef[1];
        c[1] -= w * uk[1];
#pragma unroll
        for (int i = 2; i < 5; ++i) c[i] -= w * c1[i];
    }
    if (nz > 2) {
        double w = uk[2] * c[2];
#pragma unroll
        for (int i = 3; i < 5; ++i) w += c2[i] * c[i];
        w *= coef[2];
        c[2] -= w * uk[2];
    }
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    // R = [rdiag0 c1[0] c2[0]; 0 rdiag1 c2[1]; 0 0 rdiag2]
    if (nz > 2) y2 = c[2] * rinv[2];
    if (nz > 1) y1 = (c[1] - (nz > 2 ? c2[1] * y2 : 0.0)) * rinv[1];
    if (nz > 0) y0 = (c[0] - (nz > 1 ? c1[0] * y1 : 0.0) - (nz > 2 ? c2[0] * y2 : 0.0)) * rinv[0];
    x[0] = (p0 == 0) ? y0 : ((p1 == 0) ? y1 : y2);
    x[1] = (p0 == 1) ? y0 : ((p1 == 1) ? y1 : y2);
    x[2] = (p0 == 2) ? y0 : ((p1 == 2) ? y1 : y2);
}

// ---------------------------------------------------------------- one source point: search, certificate, row
// Step 2 of an iteration for one query (icp_test_runner.cpp:1720-1726) is an exact 5-NN search; the linearisation searches for SIX
// neighbours and keeps, next to the positions, how far the query may move before the set of the nearest five can change.
struct PointQuery {
    float qx, qy, qz;     // transformed query, float (utils.hpp:630-636)
    bool reach;           // the query is close enough to the grid for a neighbour inside the radius to exist
};

struct Set6 {
    uint32_t pos[6];      // positions in the sorted target, ascending distance (kNoIdx = not found inside the bound)
    float d2[6];          // squared distances (the bound where not found: a lower bound for that neighbour)
    float lb7;            // lower bound of the 7th neighbour's squared distance
    uint32_t n_eval, n_shell;
};

// Exact 6 nearest neighbours under `bound_f` (strict).  The 32-bit-key search decides the SET of the nearest five exactly unless the
// 5th and 6th best distances are equal floats; then - lattices, duplicated points - the 64-bit-key search (distance, original index)
// is run for this lane and its first five entries are the canonical set.  (A tie between the 6th best and a point outside does not
// matter: neither belongs to the five, and both are at the distance the certificate uses.)
template <bool SWEEP, int DEPTH = 2>
DCREG_DEVFN void search6(const GridDev &g, RunList &rl, float qx, float qy, float qz, float bound_f, int max_ring, float infl, float cap, Set6 &out,
                         bool empty_block = false) {
    HeapFast<6> hf;
    knn_search<HeapFast<6>, SWEEP, false, DEPTH>(g, rl, qx, qy, qz, bound_f, max_ring, hf, infl, cap, empty_block);
    out.n_eval = hf.n_eval; out.n_shell = hf.n_shell;
#pragma unroll
    for (int j = 0; j < 6; ++j) { out.pos[j] = hf.pos[j]; out.d2[j] = hf.d[j]; }
    // the 7th neighbour: nothing the walk saw and did not keep is nearer than outside_min, and it never looked at anything nearer
    // than its final pruning distance or, failing that, the bound it started from
    out.lb7 = fminf(hf.outside_min, fminf(hf.worst_d2(), bound_f));
    if (hf.pos[5] != kNoIdx && hf.d[4] == hf.d[5]) {
        HeapExact<6> he;
        knn_search<HeapExact<6>, SWEEP>(g, rl, qx, qy, qz, bound_f, max_ring, he, 1.f, __builtin_inff(), empty_block);
        out.n_eval += he.n_eval;
#pragma unroll
        for (int j = 0; j < 6; ++j) { out.pos[j] = he.pos[j]; out.d2[j] = he.dist(j); }
        out.lb7 = out.d2[5];         // (no claim beyond the 6th: such a query is searched again every time)
    }
}

// The certificate of a search (kStateRows comment above).  a4 / a5 = distances of the 5th / 6th neighbour, a6 = the lower bound of
// the 7th's, or - where the search found fewer inside its bound - the bound, which every point it did not return lies beyond.  All
// distances are floats of dist2_nofma (within 3e-7 relative of the real squared distance) and of a 1-ulp square root, i.e. within
// 2.1e-7 of the real distance: each enters with a 2e-6 relative margin on its bad side, so that a certified set also keeps a FLOAT
// distance gap of > 1.7e-6 (a4 + a5) to the first point outside it wherever the certificate holds - ten times what the rounding of
// the fresh distances can close.
DCREG_DEVFN uint32_t make_cert(const Set6 &s, const LinArgs &a) {
    const float a4 = sqrt_approx(s.d2[4]), a5 = sqrt_approx(s.d2[5]), a6 = sqrt_approx(s.lb7);
    const float s_out = a4 * 0.999998f - a.cert_r_out;                                          // > 0: the 5th neighbour is beyond the gate radius
    // SET5: the five nearest stay the five nearest while a4 + m < a5 - m.  SET6: the five nearest stay AMONG the six known points
    // while a4 + m < a6 - m (the fifth smallest of the six new distances is at most a4 + m, everything else at least a6 - m): two
    // gaps instead of one - whenever a sixth neighbour is known this is the larger radius
    const float s5 = s.pos[4] != kNoIdx ? 0.5f * (a5 * 0.999998f - a4 * 1.000002f) : -1.f;
    const float s6 = s.pos[5] != kNoIdx ? 0.5f * (a6 * 0.999998f - a4 * 1.000002f) : -1.f;
    if (s_out > 0.f && s_out >= fmaxf(s5, s6)) return __float_as_uint(s_out) | 0x80000000u;
    // (clearing / setting the lowest mantissa bit moves s by at most one ulp: rounded down where it matters)
    if (s6 > s5) return (__float_as_uint(fmaxf(s6 * 0.9999998f, 0.f)) & ~1u) | 1u;
    return __float_as_uint(fmaxf(s5, 0.f)) & ~1u;                                                // 0: valid now, to be searched again next time
}
DCREG_DEVFN bool cert_is_out(uint32_t cert) { return (cert & 0x80000000u) != 0u; }
DCREG_DEVFN bool cert_is_set6(uint32_t cert) { return (cert & 0x80000001u) == 1u; }

// A later linearisation of the same query: does the certificate of its last search still hold at the new position?  The displacement
// is the difference of two stored floats per coordinate - exact unless a coordinate changed by more than a factor of two, and then it is
// rounded to 6e-8 relative -; its squared length and the square of s round at 1e-7: a 1e-5 relative margin covers all of it.
DCREG_DEVFN bool cert_holds(uint32_t cert, float q0x, float q0y, float q0z, float qx, float qy, float qz) {
    const float dx = qx - q0x, dy = qy - q0y, dz = qz - q0z;
    const float m2 = (dx * dx + dy * dy + dz * dz) * 1.00001f;
    const float s = __uint_as_float(cert & 0x7FFFFFFFu);            // kCertSearch is a NaN: the comparison below is false
    return m2 < s * s;
}

// Bound of a search from what the last one found: the 6th-neighbour distance is at most the largest distance to ANY six distinct
// target points.  oldpos = the state's six positions (all valid).  Inclusive bound for a strict '<' heap: the next float up.
DCREG_DEVFN float warm_bound6(const GridDev &g, const uint32_t (&oldpos)[6], float qx, float qy, float qz, float bound) {
    float4 pv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) pv[j] = g.pts[oldpos[j]];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) m = fmaxf(m, dist2_nofma(qx, qy, qz, pv[j]));
    const float incl = fmaxf(__uint_as_float(__float_as_uint(m) + 1u), 1.17549435e-38f);
    return fminf(bound, incl);
}

// search of one query inside a linearisation: bound (warm or cold), reach test, 6-NN, certificate
template <bool SWEEP, int DEPTH = 2>
DCREG_DEVFN void lin_search6(const GridDev &g, RunList &runs, const LinArgs &a, bool have_q, bool warm, const uint32_t (&oldpos)[6],
                             float qx, float qy, float qz, Set6 &st, uint32_t &cert) {
    float bound = a.radius_sq_f;
    if (warm && oldpos[5] != kNoIdx) bound = warm_bound6(g, oldpos, qx, qy, qz, bound);
    // a search whose ball is small (the query sits among its neighbours: the regime in which certificates get used) looks a little
    // further than it must - bound and pruning distance inflated alike (HeapFast::worst_d2), so that what lies beyond is a useful
    // lower bound for the 7th neighbour; a search over many cells (a query far from the surface it belongs to) is expensive enough
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double lim = (double)a.max_ring + 1.0;
    // a query farther than max_ring cells from the grid has no neighbour inside the search bound
    const bool reach = have_q && !(fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim);
    bool all_in_space = false;        // ... and that holds for every searching query of the wave (knn_search skips the block then)
    if (g.owner) {
        // A query whose bound is loose (nothing known, or neighbours of a pose far from this one): the ball of that bound cuts a wide
        // cap out of the surface it faces - candidates ~ bound - d^2, hundreds where six are wanted.  The field names an occupied
        // cell near the foot of the perpendicular (the query's own cell when that holds points); the sixth nearest of the points
        // around it (the x-run of that cell and its two x neighbours, at most 48 points) is a distance six REAL points lie within,
        // i.e. a valid bound, and a far tighter one.  "Loose" = beyond 1.5 cells more than the distance to that cell.  Measured on
        // the first iteration of a C4 run: profiles/r03_ablation.md section 9.
        bool far = false;
        uint32_t oc = kNoIdx;
        bool in_space = false;            // the query's cell is at least two cells from any occupied one: its 27-cell block is empty
        const float loose0 = a.far_loose * (float)g.h;
        // (a bound within 1.5 cells is tight wherever the query sits: the usual case once a trajectory converges - no field byte is
        // loaded for it, and a wave of such queries skips the block)
        // (round 6: a query OUTSIDE the grid's box - a source displaced beyond the map's extent, the far end of a misaligned corridor - takes the
        //  grid cell nearest to it: its owner is as real a starting point.  Such queries used to start from the search radius itself: 400
        //  candidates per lane where the others of the launch evaluate 70, and their waves - the slowest of the launch by a factor of three - were
        //  what the first launch of a run lasted, profiles/r06_ablation.md section 3)
        const bool maybe = reach && bound > loose0 * loose0;
        if (wave_any(maybe)) {
            if (maybe) {
                const int ix = clampi((int)floor(fx), 0, g.nx - 1), iy = clampi((int)floor(fy), 0, g.ny - 1), iz = clampi((int)floor(fz), 0, g.nz - 1);
                // how far outside the box the query lies (cells, the largest axis, rounded up; 0 inside)
                const double outd = fmax(fmax(fmax(-fx, fx - (double)g.nx), fmax(-fy, fy - (double)g.ny)), fmax(fmax(-fz, fz - (double)g.nz), 0.0));
                const int out = (int)ceil(fmin(outd, 1.0e6));
                const int64_t cell = ((int64_t)iz * g.ny + iy) * g.nx + ix;
                const int f0 = (int)g.gap[cell];
                const int f = f0 + out;
                in_space = f >= 2 && (out == 0 || out >= 2);        // (a query less than two cells outside: its block may still touch occupied cells)
                const float loose = ((float)f + a.far_loose) * (float)g.h;
                // (a bound that is still the search radius itself - nothing known about this query - is loose whatever the field says: seven
                //  and more cells from its surface "1.5 cells beyond the nearest occupied cell" lies past the radius, and such queries went
                //  unprobed: a cap of half a metre around the foot of their perpendicular, 450 candidates)
                if (f0 != 255 && (bound > loose * loose || bound >= a.radius_sq_f)) { oc = g.owner[cell]; far = oc != kNoIdx; }
            }
            all_in_space = !wave_any(reach && !in_space);
        }
        if (wave_any(far)) {
            // (only the sixth smallest DISTANCE of the probed points is wanted - no positions, no tie bookkeeping: a sorted six of floats
            //  kept by the median trick of HeapFast::push, six instructions per point instead of the 25 of the full insertion network)
            Top6 hb;
            hb.init(bound);
            uint32_t s_ = 0, e_ = 0;
            if (far) {
                const uint32_t ox = oc % (uint32_t)g.nx;
                const uint32_t row = (oc - ox) * (uint32_t)g.sx;                      // first table entry of the cell's (y,z) row
                s_ = g.cell_start[row + (ox > 0u ? ox - 1u : 0u) * (uint32_t)g.sx];
                e_ = g.cell_start[row + min(ox + 2u, (uint32_t)g.nx) * (uint32_t)g.sx];
                e_ = min(e_, s_ + 48u);
            }
            scan_run<decltype(hb), kProbeBatch>(g, s_, e_, qx, qy, qz, hb);
            // (round 6) The run along x holds fewer than six points where the surface the query faces is PERPENDICULAR to x - the end wall of a
            // corridor: three cells of the run, one of them on the wall.  Such queries kept the search radius as their bound (400 candidates
            // each; their waves were the slowest of a run's first launch by a factor of three): the runs of the four (y,z) rows around the
            // owner's are probed as well, while some lane of the wave still lacks its six points.
            // (only where the run itself was short: a run of six and more points none of which is inside the bound says the query has no
            //  six neighbours within reach here - an OUT point of a sparse map -, and four more scans per launch were a fifth of a small
            //  frame's registration)
            bool more = far && e_ - s_ < 6u && !(hb.d[5] < bound);
            if (wave_any(more)) {
                uint32_t ox = 0, oy = 0, oz = 0;
                if (more) { ox = oc % (uint32_t)g.nx; const uint32_t r_ = oc / (uint32_t)g.nx; oy = r_ % (uint32_t)g.ny; oz = r_ / (uint32_t)g.ny; }
#pragma unroll 1
                for (int k = 0; k < 4 && wave_any(more); ++k) {
                    uint32_t s2 = 0, e2 = 0;
                    if (more) {
                        const int yy = (int)oy + (k == 0 ? -1 : (k == 1 ? 1 : 0)), zz = (int)oz + (k == 2 ? -1 : (k == 3 ? 1 : 0));
                        if (yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
                            const uint32_t row2 = ((uint32_t)zz * (uint32_t)g.ny + (uint32_t)yy) * (uint32_t)g.nx * (uint32_t)g.sx;
                            s2 = g.cell_start[row2 + (ox > 0u ? ox - 1u : 0u) * (uint32_t)g.sx];
                            e2 = g.cell_start[row2 + min(ox + 2u, (uint32_t)g.nx) * (uint32_t)g.sx];
                            e2 = min(e2, s2 + 48u);
                        }
                    }
                    scan_run<decltype(hb), kProbeBatch>(g, s2, e2, qx, qy, qz, hb);
                    more = far && !(hb.d[5] < bound);
                }
            }
            // (six points closer than the bound: d[5] < bound, and only then does the new bound differ from the old one)
            if (far) bound = fminf(bound, fmaxf(__uint_as_float(__float_as_uint(hb.d[5]) + 1u), 1.17549435e-38f));   // inclusive, as warm_bound6
        }
    }
    const float infl = a.prune_infl, cap = a.infl_max_d2 * a.prune_infl;
    bound = fminf(fmaxf(bound, fminf(bound * infl, cap)), a.radius_sq_f);
#pragma unroll
    for (int j = 0; j < 6; ++j) { st.pos[j] = kNoIdx; st.d2[j] = bound; }
    st.lb7 = bound; st.n_eval = 0; st.n_shell = 1;
    if (reach) search6<SWEEP, DEPTH>(g, runs, qx, qy, qz, bound, a.max_ring, infl, cap, st, SWEEP && all_in_space);
    cert = make_cert(st, a);
}

// ---------------------------------------------------------------- wave-cooperative search of a few queries (sparse waves)
// Once a trajectory converges, a wave that has to search at all has one to four lanes that do (profiles/r03_ablation.md section 11);
// the lock-step search above then spends the whole wave - ~7 k instructions issued for one wave alone, ~29 us - on them.  Such
// queries are all of one kind: they moved a little since their last search, so the ball that holds their six old neighbours (the warm
// bound) lies inside their 27-cell block.  For them the roles are turned round: the 64 lanes serve ONE query at a time -
//   * the nine (y,z) rows of the block, cut to the x sub-cells the ball reaches (the arithmetic of knn_search's phase A, one row per
//     lane, up to seven queries' rows in one pass: 18 table loads in flight per query instead of 18 per lane);
//   * every row read by all lanes at once (pts[s + lane]: coalesced), one float distance each, the candidates inside the bound
//     compacted into a per-wave list (ballot + prefix count);
//   * the list ranked by the exact key (distance bits, original index): lane i counts the keys below its own - the first seven ranks
//     are the six neighbours in the canonical order (what the 64-bit-key search returns, ties included) and the distance of the
//     seventh, i.e. the lower bound SET6 certificates want, exact.
// Everything inside the ball is looked at (the rows are cut conservatively, as there), so "not in the list => at least the bound
// away" holds as it does for the lock-step search.  A wave with a query that has more than 64 points inside its bound goes through the lock-step search after all.
// Device only (wave intrinsics); team_row is plain per-thread code and is replayed on the host by the test suite.

// [s, e) of row (dy, dz) of the query's 27-cell block, cut to the sub-cells the ball of squared radius bound_f reaches: knn_search's
// phase A for one row with run-time offsets (an empty interval where the row is outside the grid or out of reach)
DCREG_DEVFN void team_row(const GridDev &g, float qx, float qy, float qz, float bound_f, int dy, int dz, uint32_t &s_out, uint32_t &e_out) {
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double big = 6.0e7;
    const double flx = floor(fmin(fmax(fx, -big), big)), fly = floor(fmin(fmax(fy, -big), big)), flz = floor(fmin(fmax(fz, -big), big));
    const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const float hf = (float)g.h;
    const float frx = (float)(fx - flx), fry = (float)(fy - fly), frz = (float)(fz - flz);
    const float gyl = fry * hf, gyh = (1.f - fry) * hf, gzl = frz * hf, gzh = (1.f - frz) * hf;
    const float gy2 = dy < 0 ? gyl * gyl * 0.99999f : (dy > 0 ? gyh * gyh * 0.99999f : 0.f);
    const float gz2 = dz < 0 ? gzl * gzl * 0.99999f : (dz > 0 ? gzh * gzh * 0.99999f : 0.f);
    const int sx = g.sx, nxf = nx * sx;
    const int cxs = cx * sx;
    const float uf = frx * (float)sx;
    const float kx = (float)g.inv_h * (float)sx * 1.00001f;
    const uint32_t unx = (uint32_t)nxf, uny = (uint32_t)ny, sxy = unx * uny;
    const bool yok = (uint32_t)(cy + dy) < uny, zok = (uint32_t)(cz + dz) < (uint32_t)nz;
    const float g2 = gy2 + gz2;
    const float rxf = fminf(sqrt_approx(fmaxf(bound_f - g2, 0.f)) * kx + 1e-4f, 1.0e6f);
    const int lo = max((int)floorf(uf - rxf), -sx), hi = min((int)floorf(uf + rxf), 2 * sx - 1);
    const int x0 = clampi(cxs + lo, 0, nxf), x1 = clampi(cxs + hi + 1, 0, nxf);   // [x0, x1)
    const bool ok = yok && zok && (x1 > x0) && !(g2 > bound_f);
    const uint32_t row = ((uint32_t)(cz + dz) * uny + (uint32_t)(cy + dy)) * unx;     // (garbage when outside: not used then)
    (void)sxy;
    s_out = g.cell_start[ok ? row + (uint32_t)x0 : 0u];
    e_out = g.cell_start[ok ? row + (uint32_t)x1 : 0u];
}

// The rows of a query's BALL, any radius (the small-frame advance pass, kernels.hpp k_advance_team): cell of the query and, per axis,
// the offsets of the cell layers the ball of squared radius bound_f reaches - the arithmetic of knn_shells' reach(), conservative alike
// (1e-5 relative + 1e-4 of a cell)
struct BallCells { int cx, cy, cz; float frx, fry, frz; int ylo, yhi, zlo, zhi; };
DCREG_DEVFN BallCells ball_cells(const GridDev &g, float qx, float qy, float qz, float bound_f) {
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double big = 6.0e7;
    const double flx = floor(fmin(fmax(fx, -big), big)), fly = floor(fmin(fmax(fy, -big), big)), flz = floor(fmin(fmax(fz, -big), big));
    BallCells b;
    b.cx = (int)flx; b.cy = (int)fly; b.cz = (int)flz;
    b.frx = (float)(fx - flx); b.fry = (float)(fy - fly); b.frz = (float)(fz - flz);
    // (a query beyond +-6e7 cells: fr is huge, every slab distance comes out beyond any bound - no row is reached, as it must be)
    const float rc = fminf(sqrt_approx(fmaxf(bound_f, 0.f)) * 1.00001f * (float)g.inv_h + 1e-4f, 1.0e6f);
    const int cap = 1 << 24;
    b.ylo = -min(cap, (int)floorf(rc + 1.f - b.fry)); b.yhi = min(cap, (int)floorf(rc + b.fry));
    b.zlo = -min(cap, (int)floorf(rc + 1.f - b.frz)); b.zhi = min(cap, (int)floorf(rc + b.frz));
    return b;
}
// [s, e) of the row (cy + dy, cz + dz) cut to the sub-cells the ball reaches (sweep_row's arithmetic); empty where the row lies
// outside the grid or beyond the ball
DCREG_DEVFN void ball_row(const GridDev &g, const BallCells &b, float bound_f, int dy, int dz, uint32_t &s_out, uint32_t &e_out) {
    const float hf = (float)g.h;
    const float gy = dy < 0 ? ((float)(-dy - 1) + b.fry) * hf : (dy > 0 ? ((float)dy - b.fry) * hf : 0.f);
    const float gz = dz < 0 ? ((float)(-dz - 1) + b.frz) * hf : (dz > 0 ? ((float)dz - b.frz) * hf : 0.f);
    const float dyz = (gy * gy + gz * gz) * 0.99999f;
    const int sx = g.sx, nxf = g.nx * sx;
    const int cxs = b.cx * sx;
    const float uf = b.frx * (float)sx, inv_hfs = (float)g.inv_h * (float)sx;
    const float xr_c = fminf((sqrt_approx(fmaxf(bound_f - dyz, 0.f)) * 1.00001f) * inv_hfs + 1e-4f, 1.0e6f);
    const int dlo = (int)floorf(uf - xr_c), dhi = (int)floorf(uf + xr_c);
    const int x0 = max(cxs + dlo, 0), x1 = min(cxs + dhi, nxf - 1) + 1;
    const int y = b.cy + dy, z = b.cz + dz;
    const bool ok = !(dyz > bound_f) && y >= 0 && y < g.ny && z >= 0 && z < g.nz && x1 > x0;
    const uint32_t row = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)nxf;     // (garbage when outside: not used then)
    s_out = g.cell_start[ok ? row + (uint32_t)x0 : 0u];
    e_out = g.cell_start[ok ? row + (uint32_t)x1 : 0u];
}

// the bound a warm search of the linearisation starts from when nothing about it is loose (lin_search6: old neighbours' distances,
// inflated like the pruning distance), and whether the ball of that bound provably lies inside the query's 27-cell block and the
// query close enough to the grid: such a search ends with the cell-table phase (knn_shells' first test), which is all the team does
DCREG_DEVFN float team_bound(const GridDev &g, const LinArgs &a, const uint32_t (&oldpos)[6], float qx, float qy, float qz, bool &tight) {
    float bound = warm_bound6(g, oldpos, qx, qy, qz, a.radius_sq_f);
    const float infl = a.prune_infl, cap = a.infl_max_d2 * a.prune_infl;
    bound = fminf(fmaxf(bound, fminf(bound * infl, cap)), a.radius_sq_f);
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double lim = (double)a.max_ring + 1.0;
    const bool reach = !(fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim);
    const double safe = g.h * (1.0 - 1e-9);
    tight = reach && (double)bound <= safe * safe * (1.0 - 1e-6);
    return bound;
}

#if DCREG_ON_DEVICE
constexpr int kTeamPre = 4;            // rows of a query whose first 64 points are requested together
constexpr int kTeamMax = 7;            // queries of one wave the team serves (nine rows each: 63 lanes for the table phase)
DCREG_DEVFN float readlane_f(float v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }
DCREG_DEVFN uint32_t readlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
DCREG_DEVFN float shfl_f(float v, int src) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)__float_as_uint(v))); }

// team_mask: the lanes (at most kTeamMax) whose queries (qx, qy, qz, bound: valid in those lanes) are searched.  Returns the lanes
// that were served; each of them holds its six positions (ascending; kNoIdx where fewer than six points lie inside the bound) and
// the certificate of the search (make_cert).
DCREG_DEVFN unsigned long long team_search6(const GridDev &g, TeamLds &T, const LinArgs &a, unsigned long long team_mask, float qx, float qy,
                                            float qz, float bound, uint32_t (&pos_out)[6], uint32_t &cert_out) {
    const int lane = threadIdx.x & (kWave - 1);
    // ---- the rows of all queries: lane 9 k + r = row r of the k-th query
    const int k_of = lane / 9, r_of = lane - 9 * k_of;
    int src = 0;
    bool has = false;
    {
        unsigned long long m = team_mask;
#pragma unroll
        for (int k = 0; k < kTeamMax; ++k) {
            const bool any = m != 0ull;
            const int L = any ? __builtin_ctzll(m) : 0;
            m &= m - 1ull;
            if (any && k_of == k) { src = L; has = true; }
        }
    }
    const float ax = shfl_f(qx, src), ay = shfl_f(qy, src), az = shfl_f(qz, src), ab = shfl_f(bound, src);
    uint32_t rs = 0, re = 0;
    if (has) team_row(g, ax, ay, az, ab, r_of % 3 - 1, r_of / 3 - 1, rs, re);
    unsigned long long served = 0ull;
    int k = 0;
    for (unsigned long long m = team_mask; m != 0ull; m &= m - 1ull, ++k) {
        const int L = __builtin_ctzll(m);
        const float ux = readlane_f(qx, L), uy = readlane_f(qy, L), uz = readlane_f(qz, L), ub = readlane_f(bound, L);
        uint32_t n = 0;                                  // candidates inside the bound so far (uniform)
        auto take = [&](const float4 &c, uint32_t p, bool in) {
            const float d2 = dist2_nofma(ux, uy, uz, c);
            const bool pass = in && d2 < ub;
            const unsigned long long pm = __builtin_amdgcn_ballot_w64(pass);
            const uint32_t slot = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
            if (pass && slot < (uint32_t)kWave) { T.d2[slot] = __float_as_uint(d2); T.pos[slot] = p; T.idx[slot] = __float_as_uint(c.w); }
            n += (uint32_t)__builtin_popcountll(pm);
        };
        // the first 64 points of (up to) kTeamPre non-empty rows are requested together, then taken in row order; what is left - more
        // rows, longer rows - follows one load at a time
        const unsigned long long live = __builtin_amdgcn_ballot_w64(re > rs) >> (9 * k);
        uint32_t rows = (uint32_t)live & 0x1FFu;
        uint32_t s4[kTeamPre] = {}, e4[kTeamPre] = {};
        float4 c4[kTeamPre];
        int n4 = 0;
#pragma unroll
        for (int i = 0; i < kTeamPre; ++i) {
            if (rows != 0u) {
                const int r = __builtin_ctz(rows);
                rows &= rows - 1u;
                s4[i] = readlane_u(rs, 9 * k + r); e4[i] = readlane_u(re, 9 * k + r);
                const uint32_t p = s4[i] + (uint32_t)lane;
                c4[i] = g.pts[p < e4[i] ? p : s4[i]];
                n4 = i + 1;
            }
        }
#pragma unroll
        for (int i = 0; i < kTeamPre; ++i) {
            if (i < n4) {
                take(c4[i], s4[i] + (uint32_t)lane, s4[i] + (uint32_t)lane < e4[i]);
                for (uint32_t p0 = s4[i] + (uint32_t)kWave; p0 < e4[i]; p0 += (uint32_t)kWave) {
                    const uint32_t p = p0 + (uint32_t)lane;
                    const float4 c = g.pts[p < e4[i] ? p : p0];
                    take(c, p, p < e4[i]);
                }
            }
        }
        while (rows != 0u) {
            const int r = __builtin_ctz(rows);
            rows &= rows - 1u;
            const uint32_t s_ = readlane_u(rs, 9 * k + r), e_ = readlane_u(re, 9 * k + r);
            for (uint32_t p0 = s_; p0 < e_; p0 += (uint32_t)kWave) {
                const uint32_t p = p0 + (uint32_t)lane;
                const float4 c = g.pts[p < e_ ? p : p0];
                take(c, p, p < e_);
            }
        }
        if (n > (uint32_t)kWave) continue;               // more than a list's worth inside the bound: the lock-step search takes it
        __builtin_amdgcn_wave_barrier();
        // ---- rank by (distance bits, original index): a total order, so the ranks are a permutation
        const bool mine = (uint32_t)lane < n;
        const uint32_t my_d2 = mine ? T.d2[lane] : 0xFFFFFFFFu, my_idx = mine ? T.idx[lane] : 0xFFFFFFFFu, my_pos = mine ? T.pos[lane] : kNoIdx;
        const unsigned long long key = ((unsigned long long)my_d2 << 32) | my_idx;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const unsigned long long kj = ((unsigned long long)readlane_u(my_d2, (int)j) << 32) | readlane_u(my_idx, (int)j);
            rank += kj < key ? 1u : 0u;
        }
        if (mine && rank < 7u) { T.out_d2[rank] = my_d2; T.out_pos[rank] = my_pos; }
        __builtin_amdgcn_wave_barrier();
        if (lane == L) {
            Set6 out;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bool got = (uint32_t)j < n;
                out.pos[j] = got ? T.out_pos[j] : kNoIdx;
                out.d2[j] = got ? __uint_as_float(T.out_d2[j]) : ub;
            }
            out.lb7 = n > 6u ? fminf(__uint_as_float(T.out_d2[6]), ub) : ub;
            out.n_eval = 0; out.n_shell = 1;
            cert_out = make_cert(out, a);
#pragma unroll
            for (int j = 0; j < 6; ++j) pos_out[j] = out.pos[j];
        }
        __builtin_amdgcn_wave_barrier();
        served |= 1ull << L;
    }
    return served;
}
#endif


// Steps 3-4a for one query with its ordered neighbour set (icp_test_runner.cpp:1727-1773): plane fit and the two gates that depend on
// the neighbours alone.  plane = {a, b, c, d} of a x + b y + c z + d = 0 with |(a,b,c)| = 1.  Returns 0 (ok), 2 (|x| < min_normal_norm,
// :1752) or 3 (plane thickness, :1773).  The plane is a function of the five points and their ORDER only - not of the query -, which
// is what lets a later linearisation reuse it (FitCert below).
template <bool FASTMATH>
DCREG_DEVFN uint8_t plane_of_set(const LinArgs &a, const KnnResult<5> &nn, double (&plane)[4]) {
    double nqx[5], nqy[5], nqz[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { nqx[j] = nn.pt[j].x; nqy[j] = nn.pt[j].y; nqz[j] = nn.pt[j].z; }
    double x[3];
    if (FASTMATH) plane_fit_qr_fast(nqx, nqy, nqz, x); else plane_fit_qr(nqx, nqy, nqz, x);
    const double ps2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    const double ps = FASTMATH ? fast_sqrt(ps2) : sqrt(ps2);
    plane[0] = plane[1] = plane[2] = plane[3] = 0.0;
    if (ps < a.min_norm) return 2;                                          // :1752
    const double pd = FASTMATH ? fast_rcp(ps) : 1.0 / ps;
    const double pa = x[0] * pd, pb = x[1] * pd, pc = x[2] * pd;
    double maxd = 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {                                           // :1763-1770
        double d = pa * nqx[j] + pb * nqy[j] + pc * nqz[j] + pd;
        d *= d;
        maxd = d > maxd ? d : maxd;
    }
    plane[0] = pa; plane[1] = pb; plane[2] = pc; plane[3] = pd;
    if (!(maxd < a.max_thick_sq)) return 3;                                 // :1773
    return 0;
}

// Steps 4b-5 for one query and its plane (icp_test_runner.cpp:1774-1812, 1863-1907): residual, weight, weight gate, float stores,
// Jacobian row.  row = [A0..A5, b, r]: the weighted Jacobian row, the right-hand side entry and the residual; it must be zero on
// entry and stays zero unless the point is effective (flag 1).  The 31 sums of the linearisation are the products of this row with
// itself, added over the points (row_products below; on the device the MFMA reduction of kernels.hpp), plus two counts: effective
// points (flag 1) and points that passed the radius gate (flag != 0, :1731).  Returns flag 1 or 4; nrm / r_out / s_out receive the
// plane normal, residual and weight.
// one rotation entry of the Euler row (see row_of_plane): plain multiplies and adds in the order icp_test_runner.cpp:2323-2335 writes them
DCREG_DEVFN double euler_entry(const double *D, double c0, double c1, double c2, double px, double py, double pz) {
#pragma clang fp contract(off)
    const double b0 = D[0] * py + D[1] * pz + D[2] * px;
    const double b1 = D[3] * py + D[4] * pz + D[5] * px;
    const double b2 = D[6] * py + D[7] * pz + D[8] * px;
    return b0 * c0 + b1 * c1 + b2 * c2;
}

template <bool FASTMATH>
DCREG_DEVFN uint8_t row_of_plane(const PoseArg &P, const LinArgs &a, const float4 &s4, float qxf, float qyf, float qzf, const double (&plane)[4],
                                 double (&row)[8], double (&nrm)[3], double &r_out, double &s_out) {
    float sxf = s4.x, syf = s4.y, szf = s4.z;
#if DCREG_ON_DEVICE
    asm volatile("" : "+v"(sxf), "+v"(syf), "+v"(szf));   // re-convert instead of keeping doubles alive across the search
    asm volatile("" : "+v"(qxf), "+v"(qyf), "+v"(qzf));
#endif
    const double px = sxf, py = syf, pz = szf;
    const double pa = plane[0], pb = plane[1], pc = plane[2], pd = plane[3];
    const double r = pa * (double)qxf + pb * (double)qyf + pc * (double)qzf + pd;   // :1774
    double s = 1.0 - a.w_slope * fabs(r);                                   // :1776
    s = s < 0.0 ? 0.0 : s;
    double ds = 0.0;
    if (a.use_wd && s > 0.0 && s < 1.0) ds = -a.w_slope * (r > 0.0 ? 1.0 : -1.0);   // :1780-1783
    nrm[0] = pa; nrm[1] = pb; nrm[2] = pc; r_out = r; s_out = s;
    if (!(s > a.w_min)) return 4;                                           // :1785
    const float cxf = (float)(s * pa), cyf = (float)(s * pb), czf = (float)(s * pc);   // :1787-1789
    const float cif = (float)(s * r);                                                    // :1790
    const double inv_s = FASTMATH ? fast_rcp(s) : 1.0 / s;
    const double nx = (double)cxf * inv_s, ny = (double)cyf * inv_s, nz = (double)czf * inv_s;   // :1889
    double A[6];
    if (!a.euler) {
        // J_r = [ (p x m)^T , m^T ],  m = R^T n   (math_utils.hpp:102-121)
        const double m0 = P.R[0] * nx + P.R[3] * ny + P.R[6] * nz;
        const double m1 = P.R[1] * nx + P.R[4] * ny + P.R[7] * nz;
        const double m2 = P.R[2] * nx + P.R[5] * ny + P.R[8] * nz;
        const double w = s + r * ds;                                                     // :1898
        A[0] = w * (py * m2 - pz * m1); A[1] = w * (pz * m0 - px * m2); A[2] = w * (px * m1 - py * m0);
        A[3] = w * m0; A[4] = w * m1; A[5] = w * m2;
    } else {
        // second engine (:2299-2346): row = [ arz, arx, ary, c^T ], c = the float-stored weighted normal s*n; no weight derivative, no
        // division by s.  Every rotation entry is three brackets (linear in the point, in the reference's relabelled order y, z, x) times
        // the three components of c: the 27 coefficients are the host's (context.hip make_lin_args) - the reference's literal trigonometric
        // products, or with DCREG_PARAM_EULER_EXACT the derivatives of R -; sums in the order of the source text, no FMA contraction.
        const double c0 = (double)cxf, c1 = (double)cyf, c2 = (double)czf;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double *D = a.dR + 9 * k;
            A[k] = euler_entry(D, c0, c1, c2, px, py, pz);
        }
        A[3] = c0; A[4] = c1; A[5] = c2;
    }
    const double b = -(double)cif;                                                       // :1906
#pragma unroll
    for (int j = 0; j < 6; ++j) row[j] = A[j];
    row[6] = b;
    row[7] = r;
    return 1;
}

// What a query keeps of its last plane fit (state rows 10-18): the plane, and a FIT word = a float s >= 0 (bit pattern) whose two
// lowest mantissa bits hold the outcome of the neighbour-only gates (0 ok, 2, 3): while the query stays within s metres of q0,
//   * the SET of its five nearest cannot change (where that set is "five of these six": s <= half the gap between the fifth and the
//     sixth), so the plane fit - a function of the set alone in the default instantiation: the five are fitted in the order of their
//     original index - would come out bitwise the same, gates included (the parity instantiation, fast_plane_fit = 0, fits in DISTANCE
//     order like the reference: there s also stays below half of every gap between consecutive distances, so the ORDER holds), and
//   * the 5th neighbour stays inside the search radius (s <= R - a4), so the radius gate (:1726) still passes:
// such a linearisation needs neither the neighbours nor the fit - it evaluates residual, weight and row on the stored plane.
// kFitNone (a NaN): nothing stored.
constexpr uint32_t kFitNone = 0xFFFFFFFFu;
struct Fit { double plane[4]; uint32_t word; };

// Steps 2b-4a for a query whose nearest-five SET is known - exactly (five positions; six = false or this lane's pos[5] = kNoIdx) or as
// "the five nearest of these six" (SET6 certificate): gather the points, recompute the float distances from the query's current
// position, put them into the canonical (distance, original index) order - the first five are bitwise the result list of a fresh
// search at this pose -, radius gate, plane fit, neighbour-only gates, and the fit word that says how far this all stays valid.
// `six` is uniform over the wave; `nn` receives the ordered five (debug dumps).  Returns 0 (radius gate failed: no plane), else 1 with
// fit.word's gate bits set.
// (fit_from_points: the same for a caller that holds the points themselves - pt[0..4], and pt[5] where use6 -: kernels.hpp k_advance_team)
template <bool FASTMATH>
DCREG_DEVFN uint8_t fit_from_points(const LinArgs &a, float qx, float qy, float qz, float4 (&pt)[6], bool use6, bool six, KnnResult<5> &nn, Fit &fit,
                                    bool presorted);
template <bool FASTMATH>
DCREG_DEVFN uint8_t fit_from_set(const GridDev &g, const LinArgs &a, float qx, float qy, float qz, const uint32_t (&pos)[6], bool six,
                                 KnnResult<5> &nn, Fit &fit, bool presorted = false) {
    float4 pt[6];
    const bool use6 = six && pos[5] != kNoIdx;
#pragma unroll
    for (int j = 0; j < 5; ++j) pt[j] = g.pts[pos[j]];
    pt[5] = use6 ? g.pts[pos[5]] : make_float4(0.f, 0.f, 0.f, 0.f);
    return fit_from_points<FASTMATH>(a, qx, qy, qz, pt, use6, six, nn, fit, presorted);
}
template <bool FASTMATH>
DCREG_DEVFN uint8_t fit_from_points(const LinArgs &a, float qx, float qy, float qz, float4 (&pt)[6], bool use6, bool six, KnnResult<5> &nn, Fit &fit,
                                    bool presorted) {
    float d2[6];
#pragma unroll
    for (int j = 0; j < 5; ++j) d2[j] = dist2_nofma(qx, qy, qz, pt[j]);
    d2[5] = use6 ? dist2_nofma(qx, qy, qz, pt[5]) : __builtin_inff();            // a lane with five sorts its padding last
    // sorting network on the key (distance bits, original index): distances are >= 0, so their bit patterns order like the values
    auto cswap = [&](int x, int y) {
        const uint64_t kx = ((uint64_t)__float_as_uint(d2[x]) << 32) | __float_as_uint(pt[x].w);
        const uint64_t ky = ((uint64_t)__float_as_uint(d2[y]) << 32) | __float_as_uint(pt[y].w);
        const bool sw = ky < kx;
        const float dx = d2[x], dy = d2[y];
        const float4 px = pt[x], py = pt[y];
        d2[x] = sw ? dy : dx; d2[y] = sw ? dx : dy;
        pt[x].x = sw ? py.x : px.x; pt[x].y = sw ? py.y : px.y; pt[x].z = sw ? py.z : px.z; pt[x].w = sw ? py.w : px.w;
        pt[y].x = sw ? px.x : py.x; pt[y].y = sw ? px.y : py.y; pt[y].z = sw ? px.z : py.z; pt[y].w = sw ? px.w : py.w;
    };
    // `presorted` (uniform over the wave): every lane's positions come straight from this launch's search, i.e. in ascending distance
    // already - the recomputed floats are the search's own - and only equal distances may still be out of (d2, idx) order
    bool sort = !presorted;
    if (presorted) {
        bool eq = false;
#pragma unroll
        for (int j = 0; j < 5; ++j) eq |= d2[j] == d2[j + 1] && (j < 4 || use6);
        sort = wave_any(eq);
    }
    if (sort) {
        if (six) {          // 12 compare-exchanges for six
            cswap(0, 1); cswap(2, 3); cswap(4, 5); cswap(0, 2); cswap(3, 5); cswap(1, 4);
            cswap(0, 1); cswap(2, 3); cswap(4, 5); cswap(1, 2); cswap(3, 4); cswap(2, 3);
        } else {            // 9 for five
            cswap(0, 1); cswap(3, 4); cswap(2, 4); cswap(2, 3); cswap(0, 3); cswap(0, 2); cswap(1, 4); cswap(1, 3); cswap(1, 2);
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) { nn.d2[j] = d2[j]; nn.pt[j] = pt[j]; nn.idx[j] = __float_as_uint(pt[j].w); nn.pos[j] = 0u; }
    nn.full = true; nn.n_eval = 0; nn.n_shell = 0;
    fit.word = 0u;                                                               // (s = 0: nothing to reuse)
    fit.plane[0] = fit.plane[1] = fit.plane[2] = fit.plane[3] = 0.0;
    if (!((double)d2[4] < a.radius_sq)) return 0;                                // :1726
    float s;
    uint8_t gate;
    if constexpr (FASTMATH) {
        // The default (fast) fit takes the five points in the order of their ORIGINAL INDEX, not of their distance: the plane is then a
        // function of the neighbour SET alone.  (The reference fits them in distance order, :1735-1747; a row permutation of the 5x3
        // least-squares system changes its solution by rounding only - the same few ulp this instantiation's reciprocal-based arithmetic
        // differs from Eigen's by anyway, and what the parity tests bound.)  What that buys: the stored plane stays valid as long as the
        // SET does - while the five nearest merely change places among themselves, which on a converging trajectory happens five times
        // as often as one of them being replaced, nothing has to be gathered, ordered and factorised again.
        KnnResult<5> byidx;
#pragma unroll
        for (int j = 0; j < 5; ++j) byidx.pt[j] = pt[j];
        {
            auto iswap = [&](int x, int y) {
                const bool sw = __float_as_uint(byidx.pt[y].w) < __float_as_uint(byidx.pt[x].w);
                const float4 px = byidx.pt[x], py = byidx.pt[y];
                byidx.pt[x].x = sw ? py.x : px.x; byidx.pt[x].y = sw ? py.y : px.y; byidx.pt[x].z = sw ? py.z : px.z; byidx.pt[x].w = sw ? py.w : px.w;
                byidx.pt[y].x = sw ? px.x : py.x; byidx.pt[y].y = sw ? px.y : py.y; byidx.pt[y].z = sw ? px.z : py.z; byidx.pt[y].w = sw ? px.w : py.w;
            };
            iswap(0, 1); iswap(3, 4); iswap(2, 4); iswap(2, 3); iswap(0, 3); iswap(0, 2); iswap(1, 4); iswap(1, 3); iswap(1, 2);
        }
        gate = plane_of_set<FASTMATH>(a, byidx, fit.plane);
        // how far the SET of the five - and the radius gate - hold beyond what the set certificate says: the room of the fifth below the
        // radius and, when the set is "five of these six", half the gap between the fifth and the sixth; 2e-6 relative margins on the
        // float distances as in make_cert.  (Equal fifth and sixth distances give 0: such a query is refitted every time.)
        const float sd4 = sqrt_approx(d2[4]), sd5 = sqrt_approx(d2[5]);
        s = a.cert_r_in - sd4 * 1.000002f;
        if (use6) s = fminf(s, 0.5f * (sd5 * 0.999998f - sd4 * 1.000002f));
    } else {
        // The parity instantiation (option fast_plane_fit = 0: the Eigen factorisation step for step) fills the rows of the 5x3 system in
        // DISTANCE order, as the reference does (matA0, :1733-1747): the plane is then a function of the ORDERED list, and the stored
        // plane is only reused while that order provably holds - half the smallest gap between consecutive distances (among the five,
        // and up to the sixth when the set is "five of these six"), and the room of the fifth below the radius; margins as above.  (A
        // pair of equal distances gives 0: such a query is refitted every time.)
        gate = plane_of_set<FASTMATH>(a, nn, fit.plane);
        float sd[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) sd[j] = sqrt_approx(d2[j]);
        s = a.cert_r_in - sd[4] * 1.000002f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fminf(s, 0.5f * (sd[j + 1] * 0.999998f - sd[j] * 1.000002f));
        if (use6) s = fminf(s, 0.5f * (sd[5] * 0.999998f - sd[4] * 1.000002f));
    }
    fit.word = (__float_as_uint(fmaxf(s, 0.f)) & ~3u) | (uint32_t)gate;
    return 1;
}
DCREG_DEVFN bool fit_holds(uint32_t word, float q0x, float q0y, float q0z, float qx, float qy, float qz) {
    const float dx = qx - q0x, dy = qy - q0y, dz = qz - q0z;
    const float m2 = (dx * dx + dy * dy + dz * dz) * 1.00001f;
    const float s = __uint_as_float(word & ~3u);                    // kFitNone is a NaN: the comparison below is false
    return m2 < s * s;
}
// the certificate of the last search, re-based on the query's present position q (inside it): what is left of its radius there
DCREG_DEVFN uint32_t cert_rebased(uint32_t cert, float q0x, float q0y, float q0z, float qx, float qy, float qz) {
    const float dx = qx - q0x, dy = qy - q0y, dz = qz - q0z;
    const float m = sqrt_approx(dx * dx + dy * dy + dz * dz) * 1.00001f;
    const float s = fmaxf(__uint_as_float(cert & 0x7FFFFFFEu) - m, 0.f) * 0.9999998f;
    return (__float_as_uint(s) & 0x7FFFFFFEu) | (cert & 0x80000001u);
}

// The 31 sums' contribution of one point, from its row and flag: [0..20] upper triangle of A A^T (row-major), [21..26]
// A b, [27] r^2, [28] b^2, [29] effective, [30] passed the radius gate.  (Host replay and tests; the device sums the same
// products on the matrix cores.)
DCREG_DEVFN void row_products(const double (&row)[8], uint8_t flag, double (&acc)[31]) {
    int idx = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int k = j; k < 6; ++k) acc[idx++] = row[j] * row[k];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[21 + j] = row[j] * row[6];
    acc[27] = row[7] * row[7];
    acc[28] = row[6] * row[6];
    acc[29] = flag == 1 ? 1.0 : 0.0;
    acc[30] = flag != 0 ? 1.0 : 0.0;
}

}  // namespace dcreg
