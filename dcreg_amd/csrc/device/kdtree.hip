// A kd-tree over the target cloud, as a COMPARATOR of the grid index ("benchmark both, keep whichever wins per
// density regime"; north_star names a KD-tree search).  Not on the product path: the linearisation searches the grid (search.hpp);
// this file exists so that the choice is measured - scripts/kdtree_compare.py, profiles/r03_kdtree_comparison.md - and tested
// (tests/test_gpu_parity.py: both indices return the same neighbours, bit for bit).  Declared in include/dcreg_debug.h.
//
// The tree is what PCL / FLANN / nanoflann build for `KdTreeFLANN::nearestKSearch` (the reference's search, utils.hpp:403,
// icp_test_runner.cpp:1722), in the shape a GPU likes: a COMPLETE implicit binary tree (node i -> 2i + 1, 2i + 2) of depth D with the
// split at the median along the widest axis of the node's points, so every leaf holds n / 2^D (<= leaf_size) points that lie
// contiguously in a re-ordered float4 array; an internal node is 5 bytes (split value + axis).  Built on the host with
// std::nth_element (the reference builds its tree on the host too, outside the timed loop: icp_test_runner.cpp:408-442).
// Search: one thread per query, depth-first, nearer child first, the farther child pushed with the squared distance to the split
// plane as its lower bound (per-axis, so it is at most the float distance dist2_nofma gives any point beyond the plane - no point is
// pruned on a bound it does not have), exact (d2, original index) keys: the result list is the canonical one of dcreg_knn.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../../include/dcreg_debug.h"
#include "context.hpp"

using namespace dcreg;

#define HIP_TRY3(ctx, expr)                                                                      \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return DCREG_E_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

namespace dcreg {

static inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

struct KdDev {
    const float *split;          // [n_internal]
    const uint8_t *axis;         // [n_internal]
    const uint32_t *leaf_start;  // [n_leaves + 1] positions in pts
    const float4 *pts;           // re-ordered target, w = bits(original index); 4 readable entries follow the last point
    uint32_t n_internal;         // 2^D - 1
};

struct KdTree {
    float *d_split = nullptr;
    uint8_t *d_axis = nullptr;
    uint32_t *d_leaf_start = nullptr;
    float4 *d_pts = nullptr;
    float4 *d_q = nullptr; size_t q_cap = 0;
    int32_t *d_idx = nullptr; float *d_d2 = nullptr; size_t out_cap = 0;
    uint32_t n_internal = 0, n_leaves = 0;
    int depth = 0, leaf_size = 0;
    int64_t n = 0;
    double build_ms = 0.0;
    KdDev dev() const { return KdDev{d_split, d_axis, d_leaf_start, d_pts, n_internal}; }
    void release() {
        for (void *p : {(void *)d_split, (void *)d_axis, (void *)d_leaf_start, (void *)d_pts, (void *)d_q, (void *)d_idx, (void *)d_d2}) if (p) (void)hipFree(p);
        *this = KdTree{};
    }
};

void kdtree_free(void *p) {
    if (!p) return;
    KdTree *t = (KdTree *)p;
    t->release();
    delete t;
}

constexpr int kKdStack = 40;      // depth <= 31 internal levels: one pending sibling per level

template <int K>
static __global__ __launch_bounds__(kBlock) void k_kdtree_knn(const float4 *__restrict__ q, uint32_t n, KdDev t, float bound_f,
                                                              int32_t *__restrict__ idx, float *__restrict__ d2out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 s4 = q[i];
    const float qx = s4.x, qy = s4.y, qz = s4.z;
    HeapExact<K> he;
    he.init(bound_f);
    uint32_t st_node[kKdStack];
    float st_bound[kKdStack];
    int sp = 0;
    st_node[sp] = 0u; st_bound[sp] = 0.f; ++sp;
    while (sp > 0) {
        --sp;
        uint32_t node = st_node[sp];
        const float nb = st_bound[sp];
        if (nb > he.worst_d2()) continue;                       // (equal: a point there may still win on its index)
        while (node < t.n_internal) {
            const int ax = (int)t.axis[node];
            const float diff = (ax == 0 ? qx : (ax == 1 ? qy : qz)) - t.split[node];
            const uint32_t left = 2u * node + 1u;
            const bool right_near = diff >= 0.f;                // left holds coordinates <= split, right >= split
            const float fb = fmaxf(nb, diff * diff);            // everything beyond the plane is at least that far (float, per axis)
            if (fb <= he.worst_d2()) { st_node[sp] = left + (right_near ? 0u : 1u); st_bound[sp] = fb; ++sp; }
            node = left + (right_near ? 1u : 0u);
        }
        const uint32_t leaf = node - t.n_internal;
        const uint32_t s = t.leaf_start[leaf], e = t.leaf_start[leaf + 1];
        for (uint32_t p = s; p < e; p += 4) {                   // four loads in flight; slots past the end are masked
            float4 c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = t.pts[p + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) he.push(dist2_nofma(qx, qy, qz, c[u]), __float_as_uint(c[u].w), p + u, p + u < e);
        }
    }
    const uint32_t oi = __float_as_uint(s4.w);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool ok = he.pos[j] != kNoIdx;
        idx[(size_t)oi * K + j] = ok ? (int32_t)(uint32_t)(he.key[j] & 0xFFFFFFFFull) : -1;
        d2out[(size_t)oi * K + j] = ok ? he.dist(j) : __builtin_inff();
    }
}

// host build: median split along the widest axis, down to depth D (complete tree); order = the points' positions in the leaves
static void kd_build_host(const std::vector<float4> &pts, int leaf_size, std::vector<float> &split, std::vector<uint8_t> &axis,
                          std::vector<uint32_t> &leaf_start, std::vector<uint32_t> &order, int &depth) {
    const size_t n = pts.size();
    depth = 0;
    while (((size_t)leaf_size << depth) < n && depth < 30) ++depth;
    const uint32_t n_internal = (1u << depth) - 1u, n_leaves = 1u << depth;
    split.assign(n_internal, 0.f); axis.assign(n_internal, 0); leaf_start.assign((size_t)n_leaves + 1, 0u);
    order.resize(n);
    std::iota(order.begin(), order.end(), 0u);
    struct Job { uint32_t node; size_t lo, hi; };
    std::vector<Job> jobs{{0u, 0, n}};
    auto coord = [&](uint32_t id, int a) { const float4 &p = pts[id]; return a == 0 ? p.x : (a == 1 ? p.y : p.z); };
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        if (j.node >= n_internal) { leaf_start[j.node - n_internal] = (uint32_t)j.lo; continue; }
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = j.lo; k < j.hi; ++k) for (int a = 0; a < 3; ++a) { const float v = coord(order[k], a); mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v); }
        int a = 0;
        if (j.hi > j.lo) { if (mx[1] - mn[1] > mx[a] - mn[a]) a = 1; if (mx[2] - mn[2] > mx[a] - mn[a]) a = 2; }
        const size_t mid = j.lo + (j.hi - j.lo) / 2;
        float s = 0.f;
        if (j.hi > j.lo) {
            std::nth_element(order.begin() + (ptrdiff_t)j.lo, order.begin() + (ptrdiff_t)mid, order.begin() + (ptrdiff_t)j.hi,
                             [&](uint32_t x, uint32_t y) { return coord(x, a) < coord(y, a); });
            s = coord(order[mid], a);              // left: [lo, mid) <= s, right: [mid, hi) >= s
        }
        split[j.node] = s; axis[j.node] = (uint8_t)a;
        jobs.push_back({2u * j.node + 1u, j.lo, mid});
        jobs.push_back({2u * j.node + 2u, mid, j.hi});
    }
    leaf_start[n_leaves] = (uint32_t)n;
}

}  // namespace dcreg

extern "C" {

int dcreg_kdtree_build(dcreg_ctx *c, int leaf_size) {
    if (!c) return DCREG_E_INVALID;
    (void)roi_deactivate(c);
    if (c->n_tgt <= 0) { c->fail("target cloud is not set"); return DCREG_E_STATE; }
    if (leaf_size < 1 || leaf_size > 256) { c->fail("kd-tree leaf size must be 1 .. 256"); return DCREG_E_INVALID; }
    HIP_TRY3(c, hipSetDevice(c->device));
    if (!c->kd) c->kd = new KdTree();
    KdTree &T = *(KdTree *)c->kd;
    T.release();
    const int64_t n = c->n_tgt;
    std::vector<float4> raw((size_t)n);
    HIP_TRY3(c, hipMemcpyAsync(raw.data(), c->d_tgt_raw, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY3(c, hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<float> split; std::vector<uint8_t> axis; std::vector<uint32_t> leaf_start, order;
    kd_build_host(raw, leaf_size, split, axis, leaf_start, order, T.depth);
    std::vector<float4> pts((size_t)n + 4, float4{0.f, 0.f, 0.f, 0.f});
    for (int64_t k = 0; k < n; ++k) pts[(size_t)k] = raw[order[(size_t)k]];
    T.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    T.n = n; T.leaf_size = leaf_size; T.n_internal = (uint32_t)split.size(); T.n_leaves = T.n_internal + 1u;
    HIP_TRY3(c, hipMalloc((void **)&T.d_split, sizeof(float) * std::max<size_t>(split.size(), 1)));
    HIP_TRY3(c, hipMalloc((void **)&T.d_axis, std::max<size_t>(axis.size(), 1)));
    HIP_TRY3(c, hipMalloc((void **)&T.d_leaf_start, sizeof(uint32_t) * leaf_start.size()));
    HIP_TRY3(c, hipMalloc((void **)&T.d_pts, sizeof(float4) * pts.size()));
    if (!split.empty()) {
        HIP_TRY3(c, hipMemcpy(T.d_split, split.data(), sizeof(float) * split.size(), hipMemcpyHostToDevice));
        HIP_TRY3(c, hipMemcpy(T.d_axis, axis.data(), axis.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY3(c, hipMemcpy(T.d_leaf_start, leaf_start.data(), sizeof(uint32_t) * leaf_start.size(), hipMemcpyHostToDevice));
    HIP_TRY3(c, hipMemcpy(T.d_pts, pts.data(), sizeof(float4) * pts.size(), hipMemcpyHostToDevice));
    return DCREG_OK;
}

int dcreg_knn_timed(dcreg_ctx *c, const float *q, int64_t n, int64_t stride, int k, double max_radius, int index, int repeats,
                    int32_t *idx, float *d2, double *kernel_ms) {
    if (!c) return DCREG_E_INVALID;
    if (!q || !idx || !d2 || n <= 0 || stride < 3 || (k != 1 && k != 5) || repeats < 1 || (index < 0 || index > 2)) { c->fail("invalid arguments"); return DCREG_E_INVALID; }
    (void)roi_deactivate(c);
    if (c->n_tgt <= 0) { c->fail("target index is not set"); return DCREG_E_STATE; }
    if (index == 1 && (!c->kd || ((KdTree *)c->kd)->n != c->n_tgt)) { c->fail("dcreg_kdtree_build first (after the last dcreg_set_target)"); return DCREG_E_STATE; }
    HIP_TRY3(c, hipSetDevice(c->device));
    if (!c->kd) c->kd = new KdTree();
    KdTree &T = *(KdTree *)c->kd;
    if ((size_t)n > T.q_cap) { if (T.d_q) (void)hipFree(T.d_q); T.d_q = nullptr; T.q_cap = 0; HIP_TRY3(c, hipMalloc((void **)&T.d_q, sizeof(float4) * (size_t)n)); T.q_cap = (size_t)n; }
    if ((size_t)n * k > T.out_cap) {
        if (T.d_idx) (void)hipFree(T.d_idx);
        if (T.d_d2) (void)hipFree(T.d_d2);
        T.d_idx = nullptr; T.d_d2 = nullptr; T.out_cap = 0;
        HIP_TRY3(c, hipMalloc((void **)&T.d_idx, sizeof(int32_t) * (size_t)n * k));
        HIP_TRY3(c, hipMalloc((void **)&T.d_d2, sizeof(float) * (size_t)n * k));
        T.out_cap = (size_t)n * k;
    }
    {
        std::vector<float4> hq((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            union { uint32_t u; float f; } w; w.u = (uint32_t)i;
            hq[(size_t)i] = float4{q[i * stride], q[i * stride + 1], q[i * stride + 2], w.f};
        }
        HIP_TRY3(c, hipMemcpy(T.d_q, hq.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice));
    }
    float bound = 3.0e38f;
    if (max_radius > 0.0 && std::isfinite(max_radius)) {            // as launch_knn: candidates with d2 < bound are kept
        const double r2 = max_radius * max_radius;
        float rf = (float)r2; if ((double)rf < r2) rf = std::nextafterf(rf, INFINITY);
        bound = std::nextafterf(rf, INFINITY);
    }
    auto launch = [&]() -> int {
        if (index != 1) return launch_knn(c, c->grid, T.d_q, n, k, max_radius, nullptr, T.d_idx, T.d_d2, index == 2);
        if (k == 1) hipLaunchKernelGGL(k_kdtree_knn<1>, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, T.d_q, (uint32_t)n, T.dev(), bound, T.d_idx, T.d_d2);
        else hipLaunchKernelGGL(k_kdtree_knn<5>, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, T.d_q, (uint32_t)n, T.dev(), bound, T.d_idx, T.d_d2);
        HIP_TRY3(c, hipGetLastError());
        return DCREG_OK;
    };
    int rc = launch();                                              // warm-up (and the result)
    if (rc) return rc;
    HIP_TRY3(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < repeats; ++r) { rc = launch(); if (rc) return rc; }
    HIP_TRY3(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY3(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIP_TRY3(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (kernel_ms) *kernel_ms = (double)ms / repeats;
    HIP_TRY3(c, hipMemcpy(idx, T.d_idx, sizeof(int32_t) * (size_t)n * k, hipMemcpyDeviceToHost));
    HIP_TRY3(c, hipMemcpy(d2, T.d_d2, sizeof(float) * (size_t)n * k, hipMemcpyDeviceToHost));
    return DCREG_OK;
}

int dcreg_kdtree_info(const dcreg_ctx *c, int32_t *depth, int32_t *leaf_size, double *build_ms) {
    if (!c || !c->kd) return DCREG_E_STATE;
    const KdTree &T = *(const KdTree *)c->kd;
    if (depth) *depth = T.depth;
    if (leaf_size) *leaf_size = T.leaf_size;
    if (build_ms) *build_ms = T.build_ms;
    return DCREG_OK;
}

}  // extern "C"
