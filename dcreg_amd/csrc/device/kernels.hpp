// gfx950 kernels of the DCReg hot path: fused exact 5-NN + plane fit + point-to-plane row + J^T J / J^T r
// reduction (DCReg/src/icp_test_runner.cpp:1714-1915), plus the index-build and k-NN utility kernels.
//
// Layout in HBM
//   target : float4 {x,y,z,bits(orig_idx)} sorted by linear grid cell (x fastest) + cell_start[n_cells+1]
//            -> the three x-adjacent cells of one (y,z) row are ONE contiguous run of points
//   source : float4 {x,y,z,bits(orig_idx)} sorted by the Hilbert-curve key of the body-frame position, so the 64
//            lanes of a wave walk neighbouring cells (a rigid pose keeps neighbours neighbours)
//   partial: double[pose][block][32]  (21 H + 6 g + sum r^2 + sum b^2 + n_eff + n_pt + pad)
// Arithmetic: k-NN distances float32, non-fused, summed x,y,z in that order (what FLANN's L2 functor does
// and what the oracle does); everything after the neighbour set is fp64, like the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcreg {

constexpr int kBlock = 256;          // 4 waves
constexpr int kSlots = 32;           // doubles per partial row
constexpr uint32_t kNoIdx = 0xFFFFFFFFu;

struct GridDev {
    double ox, oy, oz;   // origin (min corner)
    double inv_h, h;
    int nx, ny, nz;
    uint32_t n_pts;
    const uint32_t *cell_start;   // [nx*ny*nz + 1]
    const float4 *pts;            // sorted target
    const uint8_t *gap;           // [nx*ny*nz] Chebyshev distance (cells) to the nearest occupied cell, 255 = more than
    int gap_cap;                  //   gap_cap; null = not built.  Lets a query in empty space skip the rings it knows are empty
};

struct PoseArg { double R[9]; double t[3]; };

struct LinArgs {
    double radius_sq;             // R^2 in double (gate :1726)
    float radius_sq_f;            // smallest float >= R^2 (candidate prefilter)
    double max_thick_sq, min_norm, w_slope, w_min;
    int use_wd;
    int max_ring;                 // rings needed to cover the radius
    uint32_t *prev;               // [5][prev_stride] sorted-target positions of each query's last neighbour set, or null
    uint32_t prev_stride;
    int euler;                    // 1: LOAM roll/pitch/yaw row (second engine, :2296-2347) instead of the SO(3) row
    double dR[27];                // euler: dR/droll, dR/dpitch, dR/dyaw of R = Rz(yaw) Ry(pitch) Rx(roll), row-major
};

// ---------------------------------------------------------------- k-NN heaps (sorted, K entries)

// Exact heap: key = (float bits of d2) << 32 | original index -> total order (d2, idx), ties -> lower index.
template <int K_>
struct HeapExact {
    static constexpr int K = K_;
    uint64_t key[K];
    uint32_t pos[K];
    uint32_t n_eval;     // candidates evaluated (statistics only; dead code unless read)
    uint32_t n_shell;    // outermost shell scanned
    __device__ __forceinline__ void init(float bound_f) {
        const uint64_t bound = ((uint64_t)__float_as_uint(bound_f) << 32) | 0xFFFFFFFFull;
#pragma unroll
        for (int i = 0; i < K; ++i) { key[i] = bound; pos[i] = kNoIdx; }
        n_eval = 0; n_shell = 1;
    }
    __device__ __forceinline__ void push(float d2, uint32_t idx, uint32_t p, bool valid = true) {
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
    __device__ __forceinline__ float worst_d2() const { return __uint_as_float((uint32_t)(key[K - 1] >> 32)); }
    __device__ __forceinline__ float dist(int j) const { return __uint_as_float((uint32_t)(key[j] >> 32)); }
    __device__ __forceinline__ bool full() const { return pos[K - 1] != kNoIdx; }
};

// Fast heap: 32-bit keys (d2 only, strict <), branch-light insertion.  It yields the exact neighbour SET unless some
// point outside the final heap has d2 == the K-th best d2; `outside_min` tracks the smallest d2 that was ever kept out
// (rejected candidates and evicted entries alike: max(d2, K-th best before the push) is exactly that value), so the
// tie is detected exactly and the caller re-runs the exact heap.  Order among equal d2 inside the heap is fixed
// afterwards (canonical (d2, idx) order).
template <int K_>
struct HeapFast {
    static constexpr int K = K_;
    float d[K];
    uint32_t pos[K];
    float outside_min;   // smallest d2 among all points seen that are not in the heap
    uint32_t n_eval, n_shell;
    __device__ __forceinline__ void init(float bound_f) {
#pragma unroll
        for (int i = 0; i < K; ++i) { d[i] = bound_f; pos[i] = kNoIdx; }
        outside_min = __builtin_inff();
        n_eval = 0; n_shell = 1;
    }
    // valid == false: the slot is padding (d2 must then be +inf).  Branch-free: with 64 queries per wave some lane
    // accepts almost every candidate, so a divergent "if (d2 < worst)" is taken anyway and only adds exec-mask
    // juggling and merge copies.  Sorted insertion without a dependency chain: entry i becomes the median of
    // (d[i-1], d[i], d2); positions follow the same selection through the masks c[i] = d2 < d[i] (ties stay behind).
    __device__ __forceinline__ void push(float d2, uint32_t /*idx*/, uint32_t p, bool valid = true) {
        n_eval += valid ? 1u : 0u;
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
        } else {
            outside_min = fminf(outside_min, fmaxf(d2, d[K - 1]));
            bool c[K];
#pragma unroll
            for (int i = 0; i < K; ++i) c[i] = d2 < d[i];
#pragma unroll
            for (int i = K - 1; i >= 1; --i) {
                pos[i] = c[i - 1] ? pos[i - 1] : (c[i] ? p : pos[i]);
                d[i] = __builtin_amdgcn_fmed3f(d[i - 1], d[i], d2);
            }
            pos[0] = c[0] ? p : pos[0];
            d[0] = fminf(d[0], d2);
        }
    }
    __device__ __forceinline__ float worst_d2() const { return d[K - 1]; }
    __device__ __forceinline__ float dist(int j) const { return d[j]; }
    __device__ __forceinline__ bool full() const { return pos[K - 1] != kNoIdx; }
    // a point outside the heap ties with the K-th best: the set may depend on the index tie-break
    __device__ __forceinline__ bool boundary_tie() const { return full() && outside_min == d[K - 1]; }
};

// float32, NOT contracted to FMA: must round exactly like the oracle's / FLANN's plain mul+add chain
__device__ __forceinline__ float dist2_nofma(float qx, float qy, float qz, const float4 &c) {
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
__device__ __forceinline__ void body_to_global(const PoseArg &P, double px, double py, double pz, float &qx, float &qy, float &qz) {
#pragma clang fp contract(off)
    qx = (float)(P.R[0] * px + P.R[1] * py + P.R[2] * pz + P.t[0]);
    qy = (float)(P.R[3] * px + P.R[4] * py + P.R[5] * pz + P.t[1]);
    qz = (float)(P.R[6] * px + P.R[7] * py + P.R[8] * pz + P.t[2]);
}

// one run of the ring walk: four candidates per trip, their loads issued together (slots past the end are clamped
// loads that push +inf)
template <class H>
__device__ __forceinline__ void scan_run(const GridDev &g, uint32_t s, uint32_t e, float qx, float qy, float qz, H &hp);

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <class H>
__device__ __forceinline__ void knn_shells(const GridDev &g, float qx, float qy, float qz, int cx, int cy, int cz,
                                           double fx, double fy, double fz, float bound_f, int max_ring, H &hp);

// Per-thread list of the non-empty x-runs of the 3x3x3 block, kept in LDS ([slot][thread]: conflict-free).
// Surface data leaves most of the 9 (y,z) rows empty, so the list is short (~3 runs) and a run switch in the
// divergent candidate loop costs one ds_read instead of a 9-way register select.
struct RunList {
    uint32_t s[9][kBlock];
    uint32_t e[9][kBlock];
    float gap2[9][kBlock];     // squared distance from the query to the row's (y,z) slab
};

template <class H>
__device__ __forceinline__ void push_point(H &hp, float qx, float qy, float qz, const float4 &c, uint32_t p, bool valid) {
    const float d2 = dist2_nofma(qx, qy, qz, c);
    hp.push(valid ? d2 : __builtin_inff(), __float_as_uint(c.w), p, valid);     // padding slots can never enter
}

template <class H>
__device__ __forceinline__ void scan_run(const GridDev &g, uint32_t s, uint32_t e, float qx, float qy, float qz, H &hp) {
    for (uint32_t p = s; p < e; p += 4) {
        const uint32_t last = e - 1;
        float4 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = g.pts[min(p + u, last)];
#pragma unroll
        for (int u = 0; u < 4; ++u) push_point<H>(hp, qx, qy, qz, c[u], p + u, p + u < e);
    }
}

// Exact K nearest neighbours of q among points closer than sqrt(bound) ; returns with the heap filled.
// Ring k covers all cells at Chebyshev distance <= k from the query's cell; after ring k every point
// closer than k*h is in the heap, so the search stops as soon as the K-th best is inside that ball or
// the ball covers the search radius.
template <class H>
__device__ __forceinline__ void knn_search(const GridDev &g, RunList &rl, float qx, float qy, float qz, float bound_f,
                                           int max_ring, H &hp, unsigned long long *stamp = nullptr) {   // max_ring < 0: unbounded
    hp.init(bound_f);
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double lim = (double)max_ring + 1.0;
    if (max_ring >= 0) {
        // bounded search: a query farther than max_ring cells from the grid has no neighbour inside the radius
        if (fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim) return;
    }
    const double big = 1.0e9;
    const double flx = floor(fmin(fmax(fx, -big), big)), fly = floor(fmin(fmax(fy, -big), big)), flz = floor(fmin(fmax(fz, -big), big));
    const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    if (max_ring < 0) {   // unbounded: enough rings to sweep the whole grid from this cell
        const int ex = max(abs(cx), abs(cx - (nx - 1))), ey = max(abs(cy), abs(cy - (ny - 1))), ez = max(abs(cz), abs(cz - (nz - 1)));
        max_ring = max(ex, max(ey, ez)) + 1;
    }

    // ---- rings 0+1, phase A: the 9 (y,z) rows of the 3x3x3 block, each one contiguous x-run; all 18 table
    // loads are issued together, empty / out-of-reach rows are dropped, nearest rows come first
    const int tid = threadIdx.x;
    int nrun = 0;
    {
        const float hf = (float)g.h;
        const float frx = (float)(fx - flx), fry = (float)(fy - fly), frz = (float)(fz - flz);
        const float gxl = frx * hf, gxh = (1.f - frx) * hf;
        const float gyl = fry * hf, gyh = (1.f - fry) * hf, gzl = frz * hf, gzh = (1.f - frz) * hf;
        // visiting order (dy,dz): centre, 4 edge rows, 4 corner rows
        constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        uint32_t rs[9], re[9];
        float g2s[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float gy = DY[r] < 0 ? gyl : (DY[r] > 0 ? gyh : 0.f), gz = DZ[r] < 0 ? gzl : (DZ[r] > 0 ? gzh : 0.f);
            const float g2 = (gy * gy + gz * gz) * 0.99999f;
            g2s[r] = g2;
            // x-cells of this row the ball of radius sqrt(bound) can reach (conservative): a tight bound (warm
            // start) trims the three-cell run to one or two cells, or drops the row
            const float xr = sqrtf(fmaxf(bound_f - g2, 0.f)) * 1.00001f + 1e-6f * hf;
            const int x0 = clampi(cx - (gxl <= xr ? 1 : 0), 0, nx), x1 = clampi(cx + 1 + (gxh <= xr ? 1 : 0), 0, nx);   // [x0, x1)
            const int y = cy + DY[r], z = cz + DZ[r];
            const bool ok = (x1 > x0) && y >= 0 && y < ny && z >= 0 && z < nz && !(g2 > bound_f);
            const int64_t row = ok ? ((int64_t)z * ny + y) * nx : 0;
            rs[r] = ok ? g.cell_start[row + x0] : 0u;
            re[r] = ok ? g.cell_start[row + x1] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (re[r] > rs[r]) {
                rl.s[nrun][tid] = rs[r]; rl.e[nrun][tid] = re[r]; rl.gap2[nrun][tid] = g2s[r];
                ++nrun;
            }
        }
    }
    if (stamp) stamp[0] = clock64();
    // ---- phase B: flattened walk over the runs (the wave iterates max-over-lanes of the total, not the
    // sum of per-row maxima), 4 candidates in flight per trip
    {
        int ri = 0;
        uint32_t p = 0, e = 0;
        // switch to the next listed row (one per call, no inner loop: a row that the K-th best has meanwhile put out
        // of reach becomes an empty run and costs one idle trip, which is rare once the search is bounded)
        auto next_run = [&]() {
            const float g2 = rl.gap2[ri][tid];
            const uint32_t s_ = rl.s[ri][tid], e_ = rl.e[ri][tid];
            ++ri;
            const bool keep = !(g2 > hp.worst_d2());
            p = keep ? s_ : 0u; e = keep ? e_ : 0u;
        };
        // software-pipelined over two register sets, unrolled twice (no copies): the loads of trip t+1 are in flight
        // while trip t is inserted (a third set, two trips ahead: +2.5 % at 100 k points, -8 % at 1 M where the extra
        // registers cost a wave of occupancy).  Slots past the end of a run are clamped loads that push +inf.
        constexpr int W = 4;
        struct Slot { float4 c[W]; uint32_t cp, ce; bool live; };
        bool have = nrun > 0;
        if (have) next_run();
        auto fetch = [&](Slot &sl) {
            sl.live = have; sl.cp = p; sl.ce = e;
            if (have) {
                const uint32_t last = max(e, 1u) - 1u;
#pragma unroll
                for (int u = 0; u < W; ++u) sl.c[u] = g.pts[min(p + u, last)];
                p += W;
                if (p >= e) {
                    have = ri < nrun;
                    if (have) next_run();
                }
            }
        };
        auto consume = [&](const Slot &sl) {
#pragma unroll
            for (int u = 0; u < W; ++u) push_point<H>(hp, qx, qy, qz, sl.c[u], sl.cp + u, sl.cp + u < sl.ce);
        };
        Slot A, B;
        fetch(A);
        while (A.live) {
            fetch(B); consume(A);
            if (!B.live) break;
            fetch(A); consume(B);
        }
    }
    if (stamp) stamp[1] = clock64();
    knn_shells<H>(g, qx, qy, qz, cx, cy, cz, fx, fy, fz, bound_f, max_ring, hp);
}

// shells k >= 2 around cell (cx,cy,cz), global loads (sparse neighbourhoods, cloud borders, large
// misalignment).  kd-tree style pruning on the grid: a (y,z) row is skipped when its slab is farther than
// the current K-th best, and its x-run is trimmed to the cells the K-th-best ball can still reach.
// (fx,fy,fz) = query position in cell units.
template <class H>
__device__ __forceinline__ void knn_shells(const GridDev &g, float qx, float qy, float qz, int cx, int cy, int cz,
                                           double fx, double fy, double fz, float bound_f, int max_ring, H &hp) {
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const float hf = (float)g.h;
    // empty-space skip: if the nearest occupied cell is f cells away (Chebyshev), rings 1 .. f-1 hold no point
    int k0 = 1;
    if (g.gap && cx >= 0 && cx < nx && cy >= 0 && cy < ny && cz >= 0 && cz < nz) {
        const int f = min((int)g.gap[((int64_t)cz * ny + cy) * nx + cx], g.gap_cap + 1);
        k0 = max(1, f - 1);
    }
    for (int k = k0; k < max_ring; ++k) {
        // after ring k: every point within k*h (minus a rounding guard) has been seen
        const double safe = (double)k * g.h * (1.0 - 1e-9);
        const double safe2 = safe * safe * (1.0 - 1e-6);
        if ((double)hp.worst_d2() <= safe2) return;             // K-th best already inside the covered ball
        if (safe2 >= (double)bound_f) return;                   // covered ball contains the search radius
        const int kk = k + 1;                                   // scan shell kk
        hp.n_shell = (uint32_t)kk;
        const int z_lo = max(cz - kk, 0), z_hi = min(cz + kk, nz - 1);
        const int y_lo = max(cy - kk, 0), y_hi = min(cy + kk, ny - 1);
        for (int z = z_lo; z <= z_hi; ++z) {
            const int dz = z - cz;
            const float gz = dz < 0 ? (float)(fz - (double)(z + 1)) * hf : (dz > 0 ? (float)((double)z - fz) * hf : 0.f);
            if (gz * gz * 0.99999f > hp.worst_d2()) continue;
            for (int y = y_lo; y <= y_hi; ++y) {
                const int dy = y - cy;
                const float gy = dy < 0 ? (float)(fy - (double)(y + 1)) * hf : (dy > 0 ? (float)((double)y - fy) * hf : 0.f);
                const float dyz = (gy * gy + gz * gz) * 0.99999f;
                const float w = hp.worst_d2();
                if (dyz > w) continue;
                // cells the ball of radius sqrt(w) around q can reach in this row (conservative)
                const float xr = sqrtf(w - dyz) * 1.00001f + 1e-6f * hf;
                const double xr_c = (double)xr * g.inv_h;
                const int xmin = (int)floor(fmax(fx - xr_c, -1.0)), xmax = (int)floor(fmin(fx + xr_c, (double)nx));
                const int64_t row = ((int64_t)z * ny + y) * nx;
                const bool full = (dz == -kk || dz == kk || dy == -kk || dy == kk);
                if (full) {
                    const int x0 = max(max(cx - kk, xmin), 0), x1 = min(min(cx + kk, xmax), nx - 1) + 1;
                    if (x1 > x0) scan_run<H>(g, g.cell_start[row + x0], g.cell_start[row + x1], qx, qy, qz, hp);
                } else {
                    const int xa = cx - kk, xb = cx + kk;
                    if (xa >= 0 && xa < nx && xa >= xmin) scan_run<H>(g, g.cell_start[row + xa], g.cell_start[row + xa + 1], qx, qy, qz, hp);
                    if (xb >= 0 && xb < nx && xb <= xmax) scan_run<H>(g, g.cell_start[row + xb], g.cell_start[row + xb + 1], qx, qy, qz, hp);
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

template <int K>
__device__ __forceinline__ void knn_exact(const GridDev &g, RunList &rl, float qx, float qy, float qz, float bound_f, int max_ring,
                                          KnnResult<K> &res, unsigned long long *stamp = nullptr) {
    uint32_t pos[K];
    {
        HeapFast<K> hf;
        knn_search<HeapFast<K>>(g, rl, qx, qy, qz, bound_f, max_ring, hf, stamp);
        if (stamp) stamp[2] = clock64();
        res.full = hf.full();
        res.n_eval = hf.n_eval; res.n_shell = hf.n_shell;
#pragma unroll
        for (int j = 0; j < K; ++j) { res.d2[j] = hf.d[j]; pos[j] = hf.pos[j]; }
        if (hf.boundary_tie()) {                       // rare (exactly-equal float distances): exact redo
            HeapExact<K> he;
            knn_search<HeapExact<K>>(g, rl, qx, qy, qz, bound_f, max_ring, he);
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
__device__ __forceinline__ void swap_col(double (&a)[5], double (&b)[5], bool doit) {
#pragma unroll
    for (int i = 0; i < 5; ++i) { const double ta = a[i], tb = b[i]; a[i] = doit ? tb : ta; b[i] = doit ? ta : tb; }
}
__device__ __forceinline__ void swap_d(double &a, double &b, bool doit) { const double ta = a, tb = b; a = doit ? tb : ta; b = doit ? ta : tb; }
__device__ __forceinline__ void swap_i(int &a, int &b, bool doit) { const int ta = a, tb = b; a = doit ? tb : ta; b = doit ? ta : tb; }

template <int KCOL>
__device__ __forceinline__ void householder_step(double (&c0)[5], double (&c1)[5], double (&c2)[5], double (&tau)[3],
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
__device__ __forceinline__ void plane_fit_qr(const double (&qx)[5], const double (&qy)[5], const double (&qz)[5], double (&x)[3]) {
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

// XCD-aware block remap: hardware places block b on XCD b%8; give each XCD a contiguous run of query
// blocks so spatially adjacent (Hilbert-ordered) queries share that XCD's L2.  Bijective for any n.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t nx = 8u;
    const uint32_t q = n / nx, r = n % nx, xcd = b % nx, k = b / nx;
    // XCD x owns q + (x < r) blocks, laid out back to back
    const uint32_t base = xcd * q + (xcd < r ? xcd : r);
    return base + k;
}

// ---------------------------------------------------------------- the fused linearisation kernel
// MODE 0: reduction only.  MODE 1: also dump per-point results (parity tests).
struct DebugDev {
    int32_t *nn_idx; float *nn_d2; uint8_t *flag; double *normal; double *r; double *s;
    uint32_t *stats;   // per point: candidates evaluated | outermost shell << 16
    unsigned long long *clocks;   // per wave: 8 shader-clock stamps (phase breakdown), may be null
};

// Single-pose launches finish inside the kernel (no second launch): blocks are grouped in chunks of kChunk consecutive
// partial rows; the last block to finish in a chunk (ticket counter) sums that chunk's rows in a fixed order and writes
// the chunk row, stamped with the launch's sequence number, straight into pinned host-coherent memory.  The host spins on
// the stamps and adds the few chunk rows in index order.  WHO sums is timing dependent, WHAT is summed in which order is
// not: the result is deterministic.  Batched launches (many poses, few blocks each) use k_finalize instead: a ticket
// per block costs more there than the second launch (measured, profiles/r01_search_ablation.md addendum 5).
constexpr int kChunk = 64;
struct FinArgs {
    unsigned int *tickets;         // [n_chunks], zero between launches (the last arrival resets its ticket)
    double *out;                   // pinned, device-mapped: [n_chunks][kSlots]; slot 31 = sequence number
    unsigned long long seq;
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

// sum of `count` (<= kChunk) rows of kSlots doubles, fixed order: lane group g adds rows g, g+8, ... then the 8 group
// sums are added in order.  All 256 threads call; threads < 31 return the total of their slot.
__device__ __forceinline__ double block_sum_rows(const double *rows, uint32_t count, double (*sm)[kSlots]) {
    const int j = threadIdx.x & 31, grp = threadIdx.x >> 5;
    double v[kChunk / 8];
#pragma unroll
    for (int u = 0; u < kChunk / 8; ++u) {
        const uint32_t r = (uint32_t)grp + 8u * u;
        v[u] = r < count ? ld_agent(rows + (size_t)r * kSlots + j) : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < kChunk / 8; ++u) t += v[u];
    __syncthreads();
    sm[grp][j] = t;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < 31) {
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) tot += sm[gi][threadIdx.x];
    }
    return tot;
}

template <int MODE, bool FUSED>
static __global__ __launch_bounds__(kBlock, 4) void k_linearize(const float4 *__restrict__ src, uint32_t n_src, GridDev g,
                                                       PoseArg pose1, const PoseArg *__restrict__ poses, LinArgs a,
                                                       double *__restrict__ partials, uint32_t n_blocks_x, FinArgs fin, DebugDev dbg) {
    __shared__ double red[8][kSlots];
    __shared__ int s_role;
    __shared__ RunList runs;
    const uint32_t pose_id = blockIdx.y;
    const uint32_t vb = xcd_remap(blockIdx.x, n_blocks_x);
    const uint32_t i = vb * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PoseArg P;
    if (poses) P = poses[pose_id]; else P = pose1;

    double acc[31];
#pragma unroll
    for (int k = 0; k < 31; ++k) acc[k] = 0.0;
    unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
    if (MODE == 1) clk[0] = clock64();

    // ---- query, cell, reach test
    const bool have_q = i < n_src;
    const float4 s4 = have_q ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // warm start: the K-th neighbour distance is at most the largest distance to ANY K distinct target points, so
    // the neighbour set of the previous linearisation (any pose) bounds this search; the result is the same exact
    // set, found after visiting only the cells that ball touches.  The position loads and the point gathers are
    // issued here, ahead of the pose transform and the cell-table loads, so their latency overlaps with those.
    uint32_t pp[5];
    float4 pv[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) pp[j] = (a.prev && have_q) ? a.prev[(size_t)j * a.prev_stride + i] : kNoIdx;
    const bool warm = pp[4] != kNoIdx;
#pragma unroll
    for (int j = 0; j < 5; ++j) pv[j] = warm ? g.pts[pp[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
    const double px = s4.x, py = s4.y, pz = s4.z;
    float qx, qy, qz;
    body_to_global(P, px, py, pz, qx, qy, qz);
    const double fx = ((double)qx - g.ox) * g.inv_h, fy = ((double)qy - g.oy) * g.inv_h, fz = ((double)qz - g.oz) * g.inv_h;
    const double lim = (double)a.max_ring + 1.0;
    // a query farther than max_ring cells from the grid has no neighbour inside the radius
    const bool reach = have_q && !(fx < -lim || fy < -lim || fz < -lim || fx > g.nx + lim || fy > g.ny + lim || fz > g.nz + lim);

    KnnResult<5> nn;
    nn.full = false; nn.n_eval = 0; nn.n_shell = 1;
    if (MODE == 1) clk[1] = clock64();
    unsigned long long sst[3] = {0, 0, 0};
    float bound = a.radius_sq_f;
    if (warm) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) m = fmaxf(m, dist2_nofma(qx, qy, qz, pv[j]));
        // inclusive bound for a strict '<' heap: next float above m (m >= 0, finite)
        const float incl = fmaxf(__uint_as_float(__float_as_uint(m) + 1u), 1.17549435e-38f);
        bound = fminf(bound, incl);
    }
    if (reach) knn_exact<5>(g, runs, qx, qy, qz, bound, a.max_ring, nn, MODE == 1 ? sst : nullptr);
    if (a.prev && have_q) {
        const bool keep = reach && nn.full;
#pragma unroll
        for (int j = 0; j < 5; ++j) a.prev[(size_t)j * a.prev_stride + i] = keep ? nn.pos[j] : kNoIdx;
    }
    if (MODE == 1) clk[2] = clock64();

    uint8_t flag = 0;
    if (have_q) {
        const bool have5 = reach && nn.full;
        const bool in_radius = have5 && (double)nn.d2[4] < a.radius_sq;      // :1726
        if (MODE == 1) {
            const uint32_t oi = __float_as_uint(s4.w);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const bool ok = reach && nn.idx[j] != kNoIdx;
                if (dbg.nn_idx) dbg.nn_idx[5 * (size_t)oi + j] = ok ? (int32_t)nn.idx[j] : -1;
                if (dbg.nn_d2) dbg.nn_d2[5 * (size_t)oi + j] = ok ? nn.d2[j] : __builtin_inff();
            }
        }
        if (in_radius) {
            acc[30] = 1.0;                                                          // :1731
            double nqx[5], nqy[5], nqz[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) { nqx[j] = nn.pt[j].x; nqy[j] = nn.pt[j].y; nqz[j] = nn.pt[j].z; }
            double x[3];
            plane_fit_qr(nqx, nqy, nqz, x);
            const double ps = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            flag = 2;
            if (!(ps < a.min_norm)) {                                               // :1752
                const double pd = 1.0 / ps;
                const double pa = x[0] * pd, pb = x[1] * pd, pc = x[2] * pd;
                double maxd = 0.0;
#pragma unroll
                for (int j = 0; j < 5; ++j) {                                       // :1763-1770
                    double d = pa * nqx[j] + pb * nqy[j] + pc * nqz[j] + pd;
                    d *= d;
                    maxd = d > maxd ? d : maxd;
                }
                flag = 3;
                if (maxd < a.max_thick_sq) {                                        // :1773
                    const double r = pa * (double)qx + pb * (double)qy + pc * (double)qz + pd;   // :1774
                    double s = 1.0 - a.w_slope * fabs(r);                           // :1776
                    s = s < 0.0 ? 0.0 : s;
                    double ds = 0.0;
                    if (a.use_wd && s > 0.0 && s < 1.0) ds = -a.w_slope * (r > 0.0 ? 1.0 : -1.0);   // :1780-1783
                    if (MODE == 1) {
                        const uint32_t oi = __float_as_uint(s4.w);
                        if (dbg.normal) { dbg.normal[3 * (size_t)oi] = pa; dbg.normal[3 * (size_t)oi + 1] = pb; dbg.normal[3 * (size_t)oi + 2] = pc; }
                        if (dbg.r) dbg.r[oi] = r;
                        if (dbg.s) dbg.s[oi] = s;
                    }
                    flag = 4;
                    if (s > a.w_min) {                                              // :1785
                        flag = 1;
                        const float cxf = (float)(s * pa), cyf = (float)(s * pb), czf = (float)(s * pc);   // :1787-1789
                        const float cif = (float)(s * r);                                                    // :1790
                        const double inv_s = 1.0 / s;
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
                            // second engine (:2296-2347): row = [ c^T dR/droll p, c^T dR/dpitch p, c^T dR/dyaw p, c^T ] with
                            // c = the float-stored weighted normal s*n; no weight derivative, no division by s
                            const double c0 = (double)cxf, c1 = (double)cyf, c2 = (double)czf;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const double *D = a.dR + 9 * k;
                                A[k] = c0 * (D[0] * px + D[1] * py + D[2] * pz) + c1 * (D[3] * px + D[4] * py + D[5] * pz) +
                                       c2 * (D[6] * px + D[7] * py + D[8] * pz);
                            }
                            A[3] = c0; A[4] = c1; A[5] = c2;
                        }
                        const double b = -(double)cif;                                                       // :1906
                        int idx = 0;
#pragma unroll
                        for (int j = 0; j < 6; ++j)
#pragma unroll
                            for (int k = j; k < 6; ++k) acc[idx++] = A[j] * A[k];
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[21 + j] = A[j] * b;
                        acc[27] = r * r;
                        acc[28] = b * b;
                        acc[29] = 1.0;
                    }
                }
            }
        }
        if (MODE == 1 && dbg.flag) dbg.flag[__float_as_uint(s4.w)] = flag;
        if (MODE == 1 && dbg.stats) dbg.stats[__float_as_uint(s4.w)] = (nn.n_eval & 0xFFFFu) | ((nn.n_shell & 0x7FFFu) << 16);
    }

    if (MODE == 1) clk[3] = clock64();
    // transposed wave64 reduction -> lane l holds the wave total of slot (l & 31) -> LDS -> block partial
    // (fixed order, no float atomics)
    {
        double v[32];
#pragma unroll
        for (int k = 0; k < 31; ++k) v[k] = acc[k];
        v[31] = 0.0;
        const double t = wave_transpose_reduce32(v, lane);
        if (lane < 32) red[wave][lane] = t;
    }
    if (MODE == 1) clk[4] = clock64();
    __syncthreads();
    double *my_rows = partials + (size_t)pose_id * n_blocks_x * kSlots;
    if (threadIdx.x < kSlots) {
        double t = 0.0;
        if (threadIdx.x < 31) {
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) t += red[w][threadIdx.x];
        }
        if (FUSED) {
            st_agent(my_rows + (size_t)vb * kSlots + threadIdx.x, t);
            wait_stores();                                 // the row is at the coherence point before the ticket is taken
        } else {
            my_rows[(size_t)vb * kSlots + threadIdx.x] = t;
        }
    }
    if (FUSED) {
        const uint32_t chunk = vb / kChunk;
        const uint32_t csize = min((uint32_t)kChunk, n_blocks_x - chunk * kChunk);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int prev = __hip_atomic_fetch_add(&fin.tickets[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_role = (prev == csize - 1) ? 1 : 0;
            if (prev == csize - 1) __hip_atomic_store(&fin.tickets[chunk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_role == 1) {                                  // last block of this chunk: sum its rows, publish to the host
            const double t = block_sum_rows(my_rows + (size_t)chunk * kChunk * kSlots, csize, red);
            double *orow = fin.out + (size_t)chunk * kSlots;
            if (threadIdx.x < 31) { st_system(orow + threadIdx.x, t); wait_stores(); }
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store((unsigned long long *)(orow + 31), fin.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (MODE == 1 && dbg.clocks && lane == 0) {
        clk[5] = clock64();
        unsigned long long *o = dbg.clocks + ((size_t)vb * (kBlock / 64) + wave) * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = clk[k];
        o[6] = sst[0] ? (sst[0] - clk[1]) | ((sst[1] - sst[0]) << 20) | ((sst[2] - sst[1]) << 40) : 0;   // phase A | phase B | shells
        o[7] = blockIdx.x;
    }
}

// Batched launches: one block per pose sums that pose's block partials with the SAME association order as the fused
// single-pose path (chunk sums, then chunks in index order), so a batched pose is bitwise equal to the same pose
// linearised alone.  Writes 31 sums to the pinned, host-coherent result row, then publishes the sequence number the
// host spins on (no stream synchronise on the hot path).  out row layout: [0..30] sums, [31] = sequence number.
static __global__ __launch_bounds__(kBlock) void k_finalize(const double *__restrict__ partials, uint32_t n_blocks, double *__restrict__ out,
                                                            unsigned long long seq) {
    __shared__ double sm[8][kSlots];
    const uint32_t pose_id = blockIdx.x;
    const double *base = partials + (size_t)pose_id * n_blocks * kSlots;
    double tot = 0.0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += kChunk)
        tot += block_sum_rows(base + (size_t)c0 * kSlots, min((uint32_t)kChunk, n_blocks - c0), sm);
    double *orow = out + (size_t)pose_id * kSlots;
    if (threadIdx.x < 31) { st_system(orow + threadIdx.x, tot); wait_stores(); }
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store((unsigned long long *)(orow + 31), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------- plain k-NN kernel (p2p metrics, tests)
template <int K>
static __global__ __launch_bounds__(kBlock) void k_knn(const float4 *__restrict__ q, uint32_t n, GridDev g, float bound_f, int max_ring,
                                                 PoseArg pose, int apply_pose, int32_t *__restrict__ idx, float *__restrict__ d2) {
    __shared__ RunList runs;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 s4 = q[i];
    float qx = s4.x, qy = s4.y, qz = s4.z;
    if (apply_pose) {   // pcl::transformPointCloud<PointT,double>: double arithmetic, float store
        body_to_global(pose, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
    }
    KnnResult<K> nn;
    knn_exact<K>(g, runs, qx, qy, qz, bound_f, max_ring, nn);
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

// bounds[0..2] = min (ordered-uint), bounds[3..5] = max
static __global__ void k_bounds(const float4 *__restrict__ p, int64_t n, uint32_t *bounds) {
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 c = p[i];
        mn[0] = fminf(mn[0], c.x); mn[1] = fminf(mn[1], c.y); mn[2] = fminf(mn[2], c.z);
        mx[0] = fmaxf(mx[0], c.x); mx[1] = fmaxf(mx[1], c.y); mx[2] = fmaxf(mx[2], c.z);
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
    const int cx = clampi((int)floor(((double)c.x - g.ox) * g.inv_h), 0, g.nx - 1);
    const int cy = clampi((int)floor(((double)c.y - g.oy) * g.inv_h), 0, g.ny - 1);
    const int cz = clampi((int)floor(((double)c.z - g.oz) * g.inv_h), 0, g.nz - 1);
    keys[i] = (uint32_t)(((int64_t)cz * g.ny + cy) * g.nx + cx);
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
static __global__ void k_curve_keys(const float4 *__restrict__ p, int64_t n, double ox, double oy, double oz, double inv_q,
                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = p[i];
    const double fx = ((double)c.x - ox) * inv_q, fy = ((double)c.y - oy) * inv_q, fz = ((double)c.z - oz) * inv_q;
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
    keys[i] = (spread21(X[0]) << 2) | (spread21(X[1]) << 1) | spread21(X[2]);
    vals[i] = (uint32_t)i;
}

static __global__ void k_gather4(const float4 *__restrict__ in, const uint32_t *__restrict__ order, int64_t n, float4 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = in[order[i]];
}

// cell_start[c] = first sorted position whose key >= c ; sorted keys ascending
static __global__ void k_cell_start(const uint32_t *__restrict__ keys, int64_t n, int64_t n_cells, uint32_t *__restrict__ cell_start,
                             uint32_t *__restrict__ n_occupied) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const int64_t lo = (i == 0) ? 0 : (int64_t)keys[i - 1] + 1;
    const int64_t hi = (i == n) ? n_cells : (int64_t)keys[i];
    for (int64_t c = lo; c <= hi; ++c) cell_start[c] = (uint32_t)i;
    if (i < n && (i == 0 || keys[i] != keys[i - 1])) atomicAdd(n_occupied, 1u);
}

// empty-space distance field of the target grid: gap[c] = 0 on occupied cells, then one dilation pass per ring.
// In place: a pass only turns 255 into `ring`, and only looks for neighbours equal to ring - 1.
static __global__ void k_gap_init(const uint32_t *__restrict__ cell_start, int64_t n_cells, uint8_t *__restrict__ gap) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cells) gap[c] = cell_start[c + 1] > cell_start[c] ? 0 : 255;
}
static __global__ void k_gap_dilate(uint8_t *gap, int nx, int ny, int nz, int ring) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_cells = (int64_t)nx * ny * nz;
    if (c >= n_cells || gap[c] != 255) return;
    const int x = (int)(c % nx), y = (int)((c / nx) % ny), z = (int)(c / ((int64_t)nx * ny));
    const uint8_t want = (uint8_t)(ring - 1);
    bool hit = false;
    for (int dz = -1; dz <= 1 && !hit; ++dz) {
        const int zz = z + dz;
        if (zz < 0 || zz >= nz) continue;
        for (int dy = -1; dy <= 1 && !hit; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= ny) continue;
            const int64_t row = ((int64_t)zz * ny + yy) * nx;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx >= 0 && xx < nx && gap[row + xx] == want) { hit = true; break; }
            }
        }
    }
    if (hit) gap[c] = (uint8_t)ring;
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
