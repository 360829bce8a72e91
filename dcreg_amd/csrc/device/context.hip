// Device seam of the C-ABI (include/dcreg.h): owns HBM buffers, the spatial index and the launches.
// Replaces ICPContext::setTargetCloud (DCReg/include/utils.hpp:393-424) and the per-iteration body of
// Point2PlaneICP_SO3_OpenMP up to AtA/Atb (DCReg/src/icp_test_runner.cpp:1704-1919).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../../include/dcreg_debug.h"
#include "context.hpp"
#include "kernels.hpp"

namespace dcreg {

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return DCREG_E_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

template <typename T>
static int ensure(dcreg_ctx *c, T *&ptr, size_t &cap, size_t need) {
    if (need <= cap && ptr) return DCREG_OK;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; cap = 0;
    size_t n = std::max<size_t>(need, 1);
    hipError_t e = hipMalloc((void **)&ptr, n * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();      // (the failed allocation must not surface as the "launch error" of whatever is queued next)
        c->fail("hipMalloc(%zu B) failed: %s", n * sizeof(T), hipGetErrorString(e));
        return DCREG_E_NOMEM;
    }
    cap = n;
    return DCREG_OK;
}

constexpr double kCountScale = 67108864.0;      // 2^26 (search.hpp LinArgs::count_scale)
constexpr size_t kSearchCountBytes = 64 * kCounterStride * sizeof(uint32_t);     // 64 counters, one per 128-byte line
static inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }
static void drop_warm(dcreg_ctx *c);

// ------------------------------------------------------------------------------------------ index build
static int sort_pairs_u32(dcreg_ctx *c, uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n, int bits) {
    size_t tmp = 0;
    HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, n, 0, bits, c->stream));
    if (ensure(c, c->sort_tmp, c->sort_tmp_cap, tmp) != DCREG_OK) return DCREG_E_NOMEM;
    HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out, n, 0, bits, c->stream));
    return DCREG_OK;
}
// sorts by the bits [lo_bit, 63) of the keys only (stable: equal prefixes keep their order)
static int sort_pairs_u64(dcreg_ctx *c, uint64_t *keys_in, uint64_t *keys_out, uint32_t *vals_in, uint32_t *vals_out, size_t n, int lo_bit = 0) {
    size_t tmp = 0;
    HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, n, lo_bit, 63, c->stream));
    if (ensure(c, c->sort_tmp, c->sort_tmp_cap, tmp) != DCREG_OK) return DCREG_E_NOMEM;
    HIP_TRY(c, rocprim::radix_sort_pairs(c->sort_tmp, tmp, keys_in, keys_out, vals_in, vals_out, n, lo_bit, 63, c->stream));
    return DCREG_OK;
}

// bounding box of a cloud the caller handed over in HOST memory, for clouds small enough that a loop over them costs less than the
// kernel + copy + stream synchronise of device_bounds (a frame of a few thousand points: the per-registration path)
static void host_bounds(const float *xyz, int64_t n, int64_t stride, double mn[3], double mx[3]) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        const float *p = xyz + i * stride;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]);
            if (!(std::fabs(p[a]) <= 3.4e38f)) hi[0] = INFINITY;       // (a NaN compares false everywhere: make it visible, as k_bounds does)
        }
    }
    for (int a = 0; a < 3; ++a) { mn[a] = lo[a]; mx[a] = hi[a]; }
}

static int device_bounds(dcreg_ctx *c, const float4 *pts, int64_t n, double mn[3], double mx[3]) {
    uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    HIP_TRY(c, hipMemcpyAsync(c->d_scratch, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const unsigned nb = std::min<unsigned>(blocks_for(n, 256), 256);      // grid-stride: one block per CU is plenty
    hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(256), 0, c->stream, pts, n, c->d_scratch);
    uint32_t out[6];
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(out[a]); mx[a] = ord2f(out[3 + a]); }
    return DCREG_OK;
}

struct GridDst {   // where a grid build puts its products
    const float4 *raw; int64_t n;
    float4 **sorted; size_t *sorted_cap;
    uint32_t **cell_start; size_t *cell_cap;
    GridDev *grid; int64_t *n_cells;
};

// builds keys/sort/cell_start for the given cell edge; returns number of occupied cells
// (sx = x sub-cells per cell, GridDev::sx; *occupied then counts sub-cells)
static int build_grid_at(dcreg_ctx *c, const GridDst &d, double h, const double mn[3], const double mx[3], uint32_t *occupied, int sx = 1) {
    const int64_t n = d.n;
    GridDev g{};
    g.h = h; g.inv_h = 1.0 / h;
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
    g.nx = (int)std::floor((mx[0] - mn[0]) * g.inv_h) + 1;
    g.ny = (int)std::floor((mx[1] - mn[1]) * g.inv_h) + 1;
    g.nz = (int)std::floor((mx[2] - mn[2]) * g.inv_h) + 1;
    g.sx = sx;
    const int64_t n_cells = (int64_t)g.nx * sx * g.ny * g.nz;       // table entries
    g.n_pts = (uint32_t)n;
    if (ensure(c, c->d_keys, c->keys_cap, (size_t)n) || ensure(c, c->d_keys2, c->keys2_cap, (size_t)n) ||
        ensure(c, c->d_vals, c->vals_cap, (size_t)n) || ensure(c, c->d_vals2, c->vals2_cap, (size_t)n) ||
        ensure(c, *d.cell_start, *d.cell_cap, (size_t)n_cells + 1) || ensure(c, *d.sorted, *d.sorted_cap, (size_t)n + kPtsPad))
        return DCREG_E_NOMEM;
    hipLaunchKernelGGL(k_cell_keys, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream, d.raw, n, g, c->d_keys, c->d_vals);
    int bits = 1;
    while (((int64_t)1 << bits) < n_cells && bits < 32) ++bits;
    int rc = sort_pairs_u32(c, c->d_keys, c->d_keys2, c->d_vals, c->d_vals2, (size_t)n, bits);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gather4, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream, d.raw, c->d_vals2, n, *d.sorted);
    HIP_TRY(c, hipMemsetAsync(*d.sorted + n, 0, kPtsPad * sizeof(float4), c->stream));   // tail padding
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(k_cell_start, dim3(blocks_for(n_cells + 1, 256)), dim3(256), 0, c->stream, c->d_keys2, n, n_cells, *d.cell_start, c->d_scratch);
    HIP_TRY(c, hipMemcpyAsync(occupied, c->d_scratch, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    g.cell_start = *d.cell_start;
    g.pts = *d.sorted;
    *d.grid = g;
    *d.n_cells = n_cells / sx;
    return DCREG_OK;
}

static double cap_cell_for_budget(double h, const double mn[3], const double mx[3], double max_cells) {
    // enlarge h until the dense grid fits the cell budget
    for (int it = 0; it < 64; ++it) {
        const double nx = std::floor((mx[0] - mn[0]) / h) + 1, ny = std::floor((mx[1] - mn[1]) / h) + 1, nz = std::floor((mx[2] - mn[2]) / h) + 1;
        if (nx * ny * nz <= max_cells && nx < 2e9 && ny < 2e9 && nz < 2e9) return h;
        h *= 1.26;
    }
    return h;
}

static int build_index(dcreg_ctx *c, const GridDst &d, double radius_hint, uint32_t *occupied_out) {
    const int64_t n = d.n;
    if (n <= 0) { c->fail("cloud is empty"); return DCREG_E_INVALID; }
    double mn[3], mx[3];
    int rc = device_bounds(c, d.raw, n, mn, mx);
    if (rc) return rc;
    for (int a = 0; a < 3; ++a) if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) { c->fail("target cloud has non-finite coordinates"); return DCREG_E_INVALID; }
    const double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-6});
    // The cell table is DENSE (one uint32 per x sub-cell of the bounding box): on a device with 288 GB of HBM a table of 2^30 entries (4 GB:
    // a 700 m x 700 m x 12 m prior map at 0.25 m cells, two x sub-cells each) is cheaper than any indirection on the search's critical path.
    // Rounds 1-5 capped it at 2^27 entries and enlarged the cell edge beyond that - a 50 M-point map then got 0.4 m cells without x
    // sub-cells and three times the candidates per query.  The cap is an option ("max_table_entries", at most 2^31: table indices are
    // 32-bit in the kernels); beyond it the cell edge still grows.
    const double max_cells = (double)c->opt_max_table_entries;
    // the radius caps the cell edge (slightly above R so that one ring already covers the radius)
    const double h_cap = radius_hint > 0.0 ? radius_hint * 1.00001 : ext / std::cbrt((double)n) * 4.0;
    uint32_t occ = 0;
    double h = c->opt_cell > 0.0 ? c->opt_cell : h_cap;
    // (how far the table budget pushed the cell edge beyond what radius and density ask for, and whether it took the x sub-cells: dcreg_ctx::whole_capped)
    c->last_build_capped = false;
    double cap_ratio = 1.0;
    {
        const double h0 = h;
        h = cap_cell_for_budget(h, mn, mx, max_cells);
        cap_ratio = h / h0;
    }
    rc = build_grid_at(c, d, h, mn, mx, &occ);
    if (rc) return rc;
    if (c->opt_cell <= 0.0) {
        // density-adaptive cell: aim at `target_occ` points per occupied cell (surface data: occupancy ~ h^2)
        const double target_occ = 1.59 * c->opt_cell_factor * c->opt_cell_factor;   // = rho*h^2 at h = factor * r5
        double m1 = (double)n / std::max<uint32_t>(occ, 1);
        double h1 = h, expo = 2.0;
        for (int pass = 0; pass < 2 && m1 > target_occ * 1.3; ++pass) {
            double h2 = h1 * std::pow(target_occ / m1, 1.0 / expo);
            h2 = std::max(h2, h_cap / 64.0);
            {
                const double h20 = h2;
                h2 = cap_cell_for_budget(h2, mn, mx, max_cells);
                cap_ratio = h2 / h20;                              // (the last pass decides the cell edge)
            }
            if (h2 >= h1 * 0.95) break;
            rc = build_grid_at(c, d, h2, mn, mx, &occ);
            if (rc) return rc;
            const double m2 = (double)n / std::max<uint32_t>(occ, 1);
            if (m2 < m1 && h2 < h1) expo = std::min(3.0, std::max(1.0, std::log(m1 / m2) / std::log(h1 / h2)));
            h1 = h2; m1 = m2;
        }
    }
    if (occupied_out) *occupied_out = occ;
    // the cell edge is settled: cut x into sub-cells (same rows, same table loads, tighter candidate runs), as far as the
    // table budget allows
    int sx = c->opt_x_subdiv;
    while (sx > 1 && (double)d.grid->nx * sx * d.grid->ny * d.grid->nz > max_cells) sx >>= 1;
    // one step of the budget (x 1.26) with sub-cells left is what a 50 M-point, 700 m map gets: a window of its own has nothing to add there
    // (measured, window forced: 0.79 against 0.73 ms per registration); two steps, or no sub-cells at all, is where the window pays (100 M points: 0.93 -> 0.76 ms)
    c->last_build_capped = cap_ratio > 1.5 || (sx == 1 && c->opt_x_subdiv > 1 && cap_ratio > 1.0);
    if (sx > 1) {
        uint32_t occ_sub = 0;
        rc = build_grid_at(c, d, d.grid->h, mn, mx, &occ_sub, sx);
        if (rc) return rc;
    }
    return DCREG_OK;
}

// empty-space distance field of the target grid (queries far from any point skip the rings they know are empty)
static int build_gap_field(dcreg_ctx *c, double radius_hint) {
    GridDev &g = c->grid;
    g.gap = nullptr; g.gap_cap = 0; g.owner = nullptr;
    if (!c->opt_gap_field) return DCREG_OK;
    const int64_t n_cells = c->n_cells;
    int rings = 1;                                            // rings that a search up to the radius can need
    while (rings < 12 && (double)rings * g.h < (radius_hint > 0.0 ? radius_hint : 4.0 * g.h)) ++rings;
    if (rings < 2) return DCREG_OK;                           // one ring covers the radius: nothing to skip
    if (ensure(c, c->d_gap, c->gap_cap, (size_t)n_cells) || ensure(c, c->d_owner, c->owner_cap, (size_t)n_cells)) return DCREG_E_NOMEM;
    hipLaunchKernelGGL(k_gap_init, dim3(blocks_for(n_cells, 256)), dim3(256), 0, c->stream, c->d_cell_start, n_cells, g.sx, c->d_gap, c->d_owner);
    for (int r = 1; r <= rings; ++r)
        hipLaunchKernelGGL(k_gap_dilate, dim3(blocks_for(n_cells, 256)), dim3(256), 0, c->stream, c->d_gap, c->d_owner, g.nx, g.ny, g.nz, r);
    if (c->opt_far_bound) {      // the probe's owners: nearest DENSE cell (kernels.hpp k_gap_init_dense); scratch of this build only
        uint8_t *gap2 = nullptr; uint32_t *own2 = nullptr;
        if (hipMalloc((void **)&gap2, (size_t)n_cells) == hipSuccess && hipMalloc((void **)&own2, sizeof(uint32_t) * (size_t)n_cells) == hipSuccess) {
            hipLaunchKernelGGL(k_gap_init_dense, dim3(blocks_for(n_cells, 256)), dim3(256), 0, c->stream, c->d_cell_start, g.nx, n_cells, g.sx, 6u, gap2, own2);
            for (int r = 1; r <= rings; ++r)
                hipLaunchKernelGGL(k_gap_dilate, dim3(blocks_for(n_cells, 256)), dim3(256), 0, c->stream, gap2, own2, g.nx, g.ny, g.nz, r);
            hipLaunchKernelGGL(k_owner_merge, dim3(blocks_for(n_cells, 256)), dim3(256), 0, c->stream, c->d_owner, own2, n_cells);
        } else {
            (void)hipGetLastError();     // (no memory for the scratch: the plain owners serve)
        }
        const hipError_t se = hipStreamSynchronize(c->stream);
        if (gap2) (void)hipFree(gap2);
        if (own2) (void)hipFree(own2);
        HIP_TRY(c, se);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    g.gap = c->d_gap; g.gap_cap = rings;
    g.owner = c->opt_far_bound ? c->d_owner : nullptr;
    return DCREG_OK;
}

// row occupancy words of the target grid (GridDev::ymask): what the searches of the linearisation sweep instead of walking rings
static int build_row_words(dcreg_ctx *c) {
    GridDev &g = c->grid;
    g.ymask = nullptr;
    g.nxb = (g.nx + 15) >> 4; g.nyw = (g.ny + 31) >> 5;
    const int64_t n_words = (int64_t)g.nz * g.nxb * g.nyw;
    if (ensure(c, c->d_ymask, c->ymask_cap, (size_t)n_words)) return DCREG_E_NOMEM;
    hipLaunchKernelGGL(k_ymask, dim3(blocks_for(n_words, 256)), dim3(256), 0, c->stream, c->d_cell_start, g.nx, g.ny, g.nz, g.sx, g.nxb, g.nyw, c->d_ymask);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    g.ymask = c->d_ymask;
    return DCREG_OK;
}

static GridDst target_dst(dcreg_ctx *c) {
    return GridDst{c->d_tgt_raw, c->n_tgt, &c->d_tgt, &c->tgt_cap, &c->d_cell_start, &c->cell_cap, &c->grid, &c->n_cells};
}

// ------------------------------------------------------------------------------------------ the window index of a large map (context.hpp)
static void swap_index(dcreg_ctx *c) {
    dcreg_ctx::IndexSet &s = c->roi_store;
    std::swap(c->d_tgt_raw, s.raw); std::swap(c->tgt_raw_cap, s.raw_cap); std::swap(c->n_tgt, s.n);
    std::swap(c->d_tgt, s.sorted); std::swap(c->tgt_cap, s.sorted_cap);
    std::swap(c->d_cell_start, s.cell_start); std::swap(c->cell_cap, s.cell_cap);
    std::swap(c->grid, s.grid); std::swap(c->n_cells, s.n_cells); std::swap(c->occupied_cells, s.occupied);
    std::swap(c->d_gap, s.gap); std::swap(c->gap_cap, s.gap_cap);
    std::swap(c->d_owner, s.owner); std::swap(c->owner_cap, s.owner_cap);
    std::swap(c->d_ymask, s.ymask); std::swap(c->ymask_cap, s.ymask_cap);
    c->roi_active = !c->roi_active;
    drop_warm(c);                 // the states' positions are positions in the other index's order
    c->order_valid = false;
    c->last_pose_valid = false;
}

int roi_deactivate(dcreg_ctx *c) {
    if (c && c->roi_active) swap_index(c);
    return DCREG_OK;
}

// bounding box of the source at a pose: the box of the eight corners of its body-frame box
static void source_box_at(const dcreg_ctx *c, const double *R, const double *t, double lo[3], double hi[3]) {
    for (int a = 0; a < 3; ++a) { lo[a] = 1e300; hi[a] = -1e300; }
    for (int k = 0; k < 8; ++k) {
        const double p[3] = {(k & 1) ? c->src_mx[0] : c->src_mn[0], (k & 2) ? c->src_mx[1] : c->src_mn[1], (k & 4) ? c->src_mx[2] : c->src_mn[2]};
        for (int a = 0; a < 3; ++a) {
            const double w = R[a * 3] * p[0] + R[a * 3 + 1] * p[1] + R[a * 3 + 2] * p[2] + t[a];
            lo[a] = std::min(lo[a], w); hi[a] = std::max(hi[a], w);
        }
    }
}
// every query of a linearisation at this pose, and the ball of the search radius around it, lies inside the window's box
static bool roi_covers(const dcreg_ctx *c, const double *R, const double *t, double pad) {
    double lo[3], hi[3];
    source_box_at(c, R, t, lo, hi);
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] - pad >= c->roi_lo[a] && hi[a] + pad <= c->roi_hi[a])) return false;      // (also false for a NaN pose)
    }
    return true;
}
static bool roi_wanted(const dcreg_ctx *c) { return c->opt_roi_index == 2 || (c->opt_roi_index == 1 && c->whole_capped); }

static int build_gap_field(dcreg_ctx *c, double radius_hint);
static int build_row_words(dcreg_ctx *c);
// A single-pose linearisation at (R, t) with this search radius is about to be queued: make the index it should search the active one -
// the window if the map wants one (building it around the pose when there is none that covers it), the whole map otherwise.
static int roi_ensure(dcreg_ctx *c, const double *R, const double *t, double search_radius) {
    if (!roi_wanted(c)) return roi_deactivate(c);
    const double pad = std::max(search_radius, c->radius_hint) * (1.0 + c->opt_cert_margin) * 1.001 + 1e-3;
    if (c->roi_built && pad <= c->roi_pad && roi_covers(c, R, t, pad)) {
        if (c->roi_empty) return roi_deactivate(c);
        if (!c->roi_active) swap_index(c);
        return DCREG_OK;
    }
    // ---- a new window around this pose
    (void)roi_deactivate(c);                                        // the members are the whole map's
    HIP_TRY(c, hipStreamSynchronize(c->stream));                    // (nothing in flight reads the buffers that are about to be replaced)
    double lo[3], hi[3];
    source_box_at(c, R, t, lo, hi);
    for (int a = 0; a < 3; ++a) {
        if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) return DCREG_OK;          // a pose that is not a pose: the whole map, the kernels decide
        c->roi_lo[a] = lo[a] - pad - c->opt_roi_margin; c->roi_hi[a] = hi[a] + pad + c->opt_roi_margin;
    }
    c->roi_pad = pad;
    c->roi_built = false; c->roi_empty = false;
    // the map's points in the box: the x-runs of the whole map's (y,z) rows that cross it - contiguous ranges of its cell-sorted points (every cell
    // the box touches is taken whole: a few points more than the box holds, all of them the map's), read through the whole map's OWN index,
    // i.e. a pass over the window and not over the map
    const int64_t n = c->n_tgt;
    const GridDev &g = c->grid;
    int c0[3], c1[3];                                   // cell ranges [c0, c1) of the box in the whole map's grid
    const double org[3] = {g.ox, g.oy, g.oz};
    const int dims[3] = {g.nx, g.ny, g.nz};
    for (int a = 0; a < 3; ++a) {
        const double f0 = std::floor((c->roi_lo[a] - org[a]) * g.inv_h) - 1.0, f1 = std::floor((c->roi_hi[a] - org[a]) * g.inv_h) + 2.0;    // (one cell of slack: float cell keys)
        c0[a] = (int)std::min(std::max(f0, 0.0), (double)dims[a]); c1[a] = (int)std::min(std::max(f1, 0.0), (double)dims[a]);
    }
    const int ny_box = c1[1] - c0[1], nz_box = c1[2] - c0[2];
    const int64_t n_rows64 = (int64_t)ny_box * nz_box;
    uint32_t count = 0;
    c->roi_rebuilds += 1;
    if (c1[0] > c0[0] && n_rows64 > 0 && n_rows64 < ((int64_t)1 << 30)) {
        const int n_rows = (int)n_rows64, nxs = g.nx * g.sx, x0s = c0[0] * g.sx, x1s = c1[0] * g.sx;
        if (ensure(c, c->d_vals, c->vals_cap, (size_t)n_rows) || ensure(c, c->d_vals2, c->vals2_cap, (size_t)n_rows)) return DCREG_E_NOMEM;
        hipLaunchKernelGGL(k_roi_rows, dim3(blocks_for(n_rows, 256)), dim3(256), 0, c->stream, g.cell_start, nxs, g.ny, x0s, x1s, c0[1], ny_box, c0[2], n_rows, c->d_vals);
        size_t tmp = 0;
        HIP_TRY(c, rocprim::exclusive_scan(nullptr, tmp, c->d_vals, c->d_vals2, 0u, (size_t)n_rows, rocprim::plus<uint32_t>(), c->stream));
        if (ensure(c, c->sort_tmp, c->sort_tmp_cap, tmp) != DCREG_OK) return DCREG_E_NOMEM;
        HIP_TRY(c, rocprim::exclusive_scan(c->sort_tmp, tmp, c->d_vals, c->d_vals2, 0u, (size_t)n_rows, rocprim::plus<uint32_t>(), c->stream));
        uint32_t last[2] = {0u, 0u};
        HIP_TRY(c, hipMemcpyAsync(&last[0], c->d_vals + (n_rows - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&last[1], c->d_vals2 + (n_rows - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
        count = last[0] + last[1];
        if (count != 0u && (int64_t)count < n) {
            dcreg_ctx::IndexSet &s = c->roi_store;
            if (ensure(c, s.raw, s.raw_cap, (size_t)count)) return DCREG_E_NOMEM;
            hipLaunchKernelGGL(k_roi_copy, dim3((unsigned)n_rows), dim3(256), 0, c->stream, g.pts, g.cell_start, nxs, g.ny, x0s, x1s, c0[1], ny_box, c0[2], c->d_vals2, s.raw);
            HIP_TRY(c, hipGetLastError());
        }
    }
    if (count == 0u || (int64_t)count >= n) {        // nothing of the map in the box / all of it: the whole map's index serves inside this box
        c->roi_built = true; c->roi_empty = true;
        return DCREG_OK;
    }
    dcreg_ctx::IndexSet &s = c->roi_store;
    s.n = (int64_t)count;
    swap_index(c);                                                  // the members are the window's now (its buffers of the last build are reused)
    int rc = build_index(c, target_dst(c), c->radius_hint * (1.0 + c->opt_cert_margin), &c->occupied_cells);
    if (rc == DCREG_OK) rc = build_gap_field(c, c->radius_hint * (1.0 + c->opt_cert_margin));
    if (rc == DCREG_OK) rc = build_row_words(c);
    if (rc != DCREG_OK) { swap_index(c); return rc; }               // (the whole map stays usable)
    c->roi_built = true;
    return DCREG_OK;
}

// grid over the body-frame source cloud (backward pass of dcreg_p2p_error); built lazily, once per source
int build_aux_index(dcreg_ctx *c) {
    if (c->aux_valid) return DCREG_OK;
    GridDst d{c->d_src_raw, c->n_src, &c->d_aux, &c->aux_cap, &c->d_aux_cell_start, &c->aux_cell_cap, &c->aux_grid, &c->aux_n_cells};
    int rc = build_index(c, d, c->radius_hint, nullptr);
    if (rc) return rc;
    c->aux_valid = true;
    return DCREG_OK;
}

// a frame from a host buffer small enough for the registration path (icp_test_runner.cpp:442-461): staged through pinned memory, bounds on the
// host, no stream synchronise in dcreg_set_source
static bool small_host_frame(int64_t n, int64_t stride) { return n <= 65536 && n * stride <= (int64_t)1 << 20; }

static int upload_cloud(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride, bool on_device, float4 *&raw, size_t &raw_cap) {
    if (!xyz || n < 0 || stride < 3) { c->fail("invalid cloud arguments"); return DCREG_E_INVALID; }
    if (n >= ((int64_t)1 << 31)) { c->fail("cloud too large (%lld points)", (long long)n); return DCREG_E_INVALID; }
    if (ensure(c, raw, raw_cap, (size_t)n)) return DCREG_E_NOMEM;
    if (n == 0) return DCREG_OK;
    const float *src = xyz;
    if (!on_device) {
        if (ensure(c, c->d_stage, c->stage_cap, (size_t)(n * stride))) return DCREG_E_NOMEM;
        const float *from = xyz;
        const size_t words = (size_t)(n * stride);
        if (small_host_frame(n, stride)) {
            // through the context's own pinned block: safe for pageable, pinned and registered buffers alike, and the
            // copy engine reads pinned memory without the runtime's staging pass
            if (words > c->h_stage_cap) {
                if (c->h_stage) { HIP_TRY(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; c->h_stage_busy = false; }
                HIP_TRY(c, hipHostMalloc((void **)&c->h_stage, sizeof(float) * words, hipHostMallocDefault));
                c->h_stage_cap = words;
            }
            if (!c->h_stage_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->h_stage_ev, hipEventDisableTiming));
            if (c->h_stage_busy) { HIP_TRY(c, hipEventSynchronize(c->h_stage_ev)); c->h_stage_busy = false; }   // (the previous frame's upload: long done)
            std::memcpy(c->h_stage, xyz, sizeof(float) * words);
            from = c->h_stage;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_stage, from, sizeof(float) * words, hipMemcpyHostToDevice, c->stream));
        if (from == c->h_stage) { HIP_TRY(c, hipEventRecord(c->h_stage_ev, c->stream)); c->h_stage_busy = true; }
        src = c->d_stage;
    }
    hipLaunchKernelGGL(k_pack, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream, src, n, stride, raw);
    HIP_TRY(c, hipGetLastError());
    return DCREG_OK;
}

static int set_target(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride, double radius_hint, bool on_device) {
    if (!c) return DCREG_E_INVALID;
    if (n <= 0) { c->fail("target cloud is null or empty"); return DCREG_E_INVALID; }   // icp_test_runner.cpp:1643
    HIP_TRY(c, hipSetDevice(c->device));
    (void)roi_deactivate(c);             // the new map goes into the whole map's buffers; a window of the old one means nothing
    c->roi_built = false; c->whole_capped = false;
    int rc = upload_cloud(c, xyz, n, stride, on_device, c->d_tgt_raw, c->tgt_raw_cap);
    if (rc) return rc;
    c->n_tgt = n;
    c->radius_hint = radius_hint;
    // the cells are sized for the SEARCH radius (make_lin_args): one ring must cover it
    rc = build_index(c, target_dst(c), radius_hint * (1.0 + c->opt_cert_margin), &c->occupied_cells);
    if (rc) { c->n_tgt = 0; return rc; }
    c->whole_capped = c->last_build_capped;
    rc = build_gap_field(c, radius_hint * (1.0 + c->opt_cert_margin));
    if (rc) { c->n_tgt = 0; return rc; }
    rc = build_row_words(c);
    if (rc) { c->n_tgt = 0; return rc; }
    drop_warm(c);            // positions and certificates refer to the old target
    c->order_valid = false;  // ... and the cost estimate of the query groups to the old map
    c->last_pose_valid = false;
    c->n_batch_states = 0;
    return DCREG_OK;
}

static int set_source(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride, bool on_device) {
    if (!c) return DCREG_E_INVALID;
    if (n <= 0) { c->fail("measure cloud is null or empty"); return DCREG_E_INVALID; }  // icp_test_runner.cpp:1635
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = upload_cloud(c, xyz, n, stride, on_device, c->d_src_raw, c->src_raw_cap);
    if (rc) return rc;
    // Hilbert-curve order in the body frame (pose independent: a rigid motion keeps neighbours neighbours)
    double mn[3], mx[3];
    const bool small_host = !on_device && small_host_frame(n, stride);   // a frame from a host buffer: the registration path (icp_test_runner.cpp:442-461)
    if (small_host) {
        host_bounds(xyz, n, stride, mn, mx);
    } else {
        rc = device_bounds(c, c->d_src_raw, n, mn, mx);
        if (rc) return rc;
    }
    for (int a = 0; a < 3; ++a) if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) { c->fail("source cloud has non-finite coordinates"); return DCREG_E_INVALID; }
    for (int a = 0; a < 3; ++a) { c->src_mn[a] = mn[a]; c->src_mx[a] = mx[a]; }
    const double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-6});
    const double inv_q = 2097151.0 / ext * 0.999999;
    {   // farthest corner of the bounding box: no point is farther from the body-frame origin
        double r2 = 0.0;
        for (int k = 0; k < 3; ++k) { const double m = std::max(std::fabs(mn[k]), std::fabs(mx[k])); r2 += m * m; }
        c->src_radius = std::sqrt(r2);
    }
    if (ensure(c, c->d_mkeys, c->mkeys_cap, (size_t)n) || ensure(c, c->d_mkeys2, c->mkeys2_cap, (size_t)n) ||
        ensure(c, c->d_vals, c->vals_cap, (size_t)n) || ensure(c, c->d_vals2, c->vals2_cap, (size_t)n) ||
        ensure(c, c->d_src, c->src_cap, (size_t)n))
        return DCREG_E_NOMEM;
    if (c->opt_keep_source_order) {      // experiments: the caller supplies the processing order
        HIP_TRY(c, hipMemcpyAsync(c->d_src, c->d_src_raw, sizeof(float4) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    } else {
        // the curve is resolved as far as the cloud can tell cells apart: log8(n) levels + 4 (a 4096-fold finer grid than one point
        // per cell) - 27 key bits for an 8 k-point frame instead of 63, i.e. half the radix passes; points that share the prefix keep
        // their input order (the sort is stable)
        int levels = 4;
        while (levels < 21 && ((int64_t)1 << (3 * (levels - 4))) < n) ++levels;
        // (round 6: ONE workgroup ordering a frame of <= 8192 points in LDS - keys, bitonic sort, gather, a single launch - was built and
        //  measured: 35 us of host time instead of 39, and 120 us of DEVICE time in front of the first linearisation instead of 15; removed)
        hipLaunchKernelGGL(k_curve_keys, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream, c->d_src_raw, n, mn[0], mn[1], mn[2], inv_q, c->opt_curve_x_scale, c->d_mkeys, c->d_vals);
        rc = sort_pairs_u64(c, c->d_mkeys, c->d_mkeys2, c->d_vals, c->d_vals2, (size_t)n, 63 - 3 * levels);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather4, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream, c->d_src_raw, c->d_vals2, n, c->d_src);
    }
    // A small frame from a host buffer went through the context's pinned block (upload_cloud): the caller's buffer is consumed, whatever
    // kind of memory it is, and nothing has to be waited for - the first linearisation queues behind the sort, and a device fault surfaces
    // at that linearisation (include/dcreg.h says so).  Everything else is waited for here.
    const bool must_wait = !small_host;
    if (must_wait) HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->n_src = n;
    {   // dispatch groups of the single-pose launches (kernels.hpp k_group_cost): multiples of 16 query blocks, at most kMaxGroups of them
        const uint32_t nb = blocks_for(n, kLinBlock);
        const uint32_t gq = 16u;                  // one XCD's run of query blocks (opt_xcd_chunk)
        c->group_blocks = gq * std::max<uint32_t>(1u, (nb + gq * kMaxGroups - 1) / (gq * kMaxGroups));
        c->n_groups = std::min<uint32_t>(nb / c->group_blocks, (uint32_t)kMaxGroups);
        c->order_valid = false;                   // estimated at the pose of the next single-pose linearisation
    }
    c->aux_valid = false;
    drop_warm(c);
    c->n_batch_states = 0;
    c->last_pose_valid = false;
    return DCREG_OK;
}

// ------------------------------------------------------------------------------------------ linearise
static int make_lin_args(dcreg_ctx *c, const dcreg_lin_params *p, LinArgs &a) {
    if (!p || !(p->search_radius > 0.0)) { c->fail("invalid linearisation parameters"); return DCREG_E_INVALID; }
    if (p->k != 5 && p->k != 0) { c->fail("only k = 5 is supported (icp_test_runner.cpp:1722)"); return DCREG_E_INVALID; }
    a.radius_sq = p->search_radius * p->search_radius;
    {   // searches cover a little more than the gate radius (certificates of "5th neighbour beyond R" spend the difference)
        const double rs = p->search_radius * (1.0 + c->opt_cert_margin), r2 = rs * rs;
        float rf = (float)r2;
        if ((double)rf < r2) rf = std::nextafterf(rf, INFINITY);
        a.radius_sq_f = std::nextafterf(rf, INFINITY);   // candidates with d2 < this are kept
        float ro = (float)(p->search_radius * (1.0 + 1e-5));
        if ((double)ro < p->search_radius * (1.0 + 1e-5)) ro = std::nextafterf(ro, INFINITY);
        a.cert_r_out = ro;
        float ri = (float)(p->search_radius * (1.0 - 1e-5));
        if ((double)ri > p->search_radius * (1.0 - 1e-5)) ri = std::nextafterf(ri, 0.0f);
        a.cert_r_in = ri;
    }
    a.max_thick_sq = p->max_plane_thickness_sq; a.min_norm = p->min_normal_norm;
    a.w_slope = p->weight_slope; a.w_min = p->weight_min; a.use_wd = p->use_weight_derivative;
    a.warm = c->opt_warm ? 1 : 0;
    a.team_max = std::min(std::max(c->opt_team_max, 0), kTeamMax);
    a.far_loose = (float)c->opt_far_loose;
    a.prune_infl = (float)((1.0 + c->opt_cert_inflate) * (1.0 + c->opt_cert_inflate));
    a.infl_max_d2 = (float)(4.0 * c->grid.h * c->grid.h);
    int k = 1;
    while (k < 100000) {
        const double safe = (double)k * c->grid.h * (1.0 - 1e-9);
        if (safe * safe * (1.0 - 1e-6) >= (double)a.radius_sq_f) break;
        ++k;
    }
    a.max_ring = k;
    c->last_max_ring = k;
    a.count_scale = c->n_src < ((int64_t)1 << 26) ? kCountScale : 0.0;      // (strictly below: a count of 2^26 would read as one more of the number riding above it)
    a.euler = p->parameterization != DCREG_PARAM_SO3 ? 1 : 0;
    if (p->parameterization != DCREG_PARAM_SO3 && p->parameterization != DCREG_PARAM_EULER && p->parameterization != DCREG_PARAM_EULER_EXACT) { c->fail("unknown parameterization"); return DCREG_E_INVALID; }
    a.dR = nullptr;
    if (a.euler) {
        // The kernel's Euler branch (search.hpp row_of_plane) evaluates, for k = 0 (roll column) / 1 (pitch) / 2 (yaw),
        //     A[k] = c.x (D[9k+0] p.y + D[9k+1] p.z + D[9k+2] p.x) + c.y (D[9k+3] p.y + ...) + c.z (D[9k+6] p.y + ...)
        // - the shape of icp_test_runner.cpp:2323-2335 with its axis relabelling (pointOri = (p.y, p.z, p.x), coeff = (c.y, c.z, c.x)) undone for c
        // and kept for p, so that every product and every sum is taken in the order of the source text.  The 27 coefficients come from here.
        if (!c->h_euler) {
            HIP_TRY(c, hipHostMalloc((void **)&c->h_euler, 27 * sizeof(double), hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void **)&c->d_euler, 27 * sizeof(double)));
        }
        // (the staging block is reused: the launches of the Euler engine are blocking calls, one at a time)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        double *D = c->h_euler;
        if (p->parameterization == DCREG_PARAM_EULER) {
            // the reference's row, literally (:2299-2346): srx / crx of PITCH, sry / cry of YAW, srz / crz of ROLL; the brackets of arx times coeff.z,
            // coeff.x, coeff.y (LOAM: x, y, z), of ary times coeff.z, coeff.y, of arz times coeff.z, coeff.x, coeff.y; matA row = [arz, arx, ary, ...]
            const double srx = std::sin(p->euler_rpy[1]), crx = std::cos(p->euler_rpy[1]);
            const double sry = std::sin(p->euler_rpy[2]), cry = std::cos(p->euler_rpy[2]);
            const double srz = std::sin(p->euler_rpy[0]), crz = std::cos(p->euler_rpy[0]);
            const double crx_sry = crx * sry, crz_sry = crz * sry, srx_sry = srx * sry, srx_srz = srx * srz;   // :2319-2322
            // arz (:2331-2335) -> column 0
            D[0] = crz * srx_sry - cry * srz;   D[1] = -cry * crz - srx_sry * srz;   D[2] = 0.0;      // x coeff.z (= c.x)
            D[3] = crx * crz;                   D[4] = -(crx * srz);                 D[5] = 0.0;      // x coeff.x (= c.y)
            D[6] = sry * srz + cry * crz * srx; D[7] = crz_sry - cry * srx_srz;      D[8] = 0.0;      // x coeff.y (= c.z)
            // arx (:2323-2326) -> column 1
            D[9] = crx_sry * srz;               D[10] = crx * crz_sry;               D[11] = -srx_sry;        // x coeff.z
            D[12] = -srx_srz;                   D[13] = -(crz * srx);                D[14] = -crx;            // x coeff.x
            D[15] = crx * cry * srz;            D[16] = crx * cry * crz;             D[17] = -(cry * srx);    // x coeff.y
            // ary (:2327-2330) -> column 2 (no coeff.x bracket)
            D[18] = cry * srx_srz - crz_sry;    D[19] = sry * srz + cry * crz * srx; D[20] = crx * cry;       // x coeff.z
            D[21] = 0.0;                        D[22] = 0.0;                         D[23] = 0.0;
            D[24] = -cry * crz - srx_sry * srz; D[25] = cry * srz - crz * srx_sry;   D[26] = -crx_sry;        // x coeff.y
        } else {
            // exact derivatives of R = Rz(yaw) Ry(pitch) Rx(roll) (Pose6D2Matrix, utils.hpp:452-460), columns in the kernel's (p.y, p.z, p.x) order
            const double cr = std::cos(p->euler_rpy[0]), sr = std::sin(p->euler_rpy[0]);
            const double cp = std::cos(p->euler_rpy[1]), sp = std::sin(p->euler_rpy[1]);
            const double cy = std::cos(p->euler_rpy[2]), sy = std::sin(p->euler_rpy[2]);
            const double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr}, dRx[9] = {0, 0, 0, 0, -sr, -cr, 0, cr, -sr};
            const double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp}, dRy[9] = {-sp, 0, cp, 0, 0, 0, -cp, 0, -sp};
            const double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1}, dRz[9] = {-sy, -cy, 0, cy, -sy, 0, 0, 0, 0};
            auto mul3 = [](const double *A, const double *B, const double *C3, double *out) {
                double T[9], O[9];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j]; T[i * 3 + j] = s; }
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += T[i * 3 + k] * C3[k * 3 + j]; O[i * 3 + j] = s; }
                for (int i = 0; i < 3; ++i) { out[i * 3 + 0] = O[i * 3 + 1]; out[i * 3 + 1] = O[i * 3 + 2]; out[i * 3 + 2] = O[i * 3 + 0]; }
            };
            mul3(Rz, Ry, dRx, D); mul3(Rz, dRy, Rx, D + 9); mul3(dRz, Ry, Rx, D + 18);
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_euler, c->h_euler, 27 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        a.dR = c->d_euler;
    }
    return DCREG_OK;
}

// One linearisation = begin (argument checks, buffers, launches: returns as soon as the work is queued) + end (wait for
// the pinned result rows, unpack).  Two slots with their own pose / partial / result buffers let a caller keep one batch
// on the device while the host works on the other (dcreg_linearize_batch_begin / _end); the blocking entry points use
// slot 0.  Debug dumps are synchronous and only exist on slot 0.
static void free_tmp(LinSlot &S) {
    for (void *p2 : S.tmp_dev) (void)hipFree(p2);
    S.tmp_dev.clear();
}

// after a launch that may not have run: what the states hold is unknown
static void drop_warm(dcreg_ctx *c) {
    c->state_valid = false;
    c->adv_counts_dirty = true;        // (a pass may have run without the k_lin that takes its counts)
    std::fill(c->batch_state_valid.begin(), c->batch_state_valid.end(), (uint8_t)0);
}
// publish a gate record (kernels.hpp GateHost): pose words and checksum first, the number last.  R == null: an abort, the pose words
// stay whatever they were.
static void gate_publish(dcreg_ctx *c, unsigned long long seq_word, const double *R, const double *t) {
    unsigned long long w[kGateWords];
    w[0] = seq_word;
    for (int k = 1; k < kGateWords - 1; ++k) w[k] = c->h_gate->w[k];
    if (R) {
        std::memcpy(&w[1], R, 9 * sizeof(double));
        std::memcpy(&w[10], t, 3 * sizeof(double));
    }
    unsigned long long x = kGateSalt;
    for (int k = 0; k < kGateWords - 1; ++k) x ^= w[k];
    w[kGateWords - 1] = x;
    volatile unsigned long long *dst = c->h_gate->w;
    for (int k = 1; k < kGateWords; ++k) dst[k] = w[k];
    __atomic_store_n(&c->h_gate->w[0], w[0], __ATOMIC_RELEASE);
}
static void gate_call_off(dcreg_ctx *c) { gate_publish(c, (c->gate_seq << 1) | 1ull, nullptr, nullptr); }

// dispatch order of the query-block groups (kernels.hpp k_group_cost): estimated cost of every group at this pose, heaviest first
static int estimate_dispatch_order(dcreg_ctx *c, const double *R9, const double *t3, int max_ring) {
    if (!c->d_group_est) { HIP_TRY(c, hipMalloc((void **)&c->d_group_est, sizeof(float) * kMaxGroups)); }
    PoseArg P{};
    for (int k = 0; k < 9; ++k) P.R[k] = R9[k];
    for (int k = 0; k < 3; ++k) P.t[k] = t3[k];
    const uint32_t ng = c->n_groups;
    const uint32_t lead = (blocks_for(c->n_src, kLinBlock) - ng * c->group_blocks) * (uint32_t)kLinBlock;      // points in front of the first group
    hipLaunchKernelGGL(k_group_cost, dim3(ng), dim3(256), 0, c->stream, c->d_src + lead, (uint32_t)c->n_src - lead, c->grid, P, c->group_blocks * (uint32_t)kLinBlock,
                       max_ring, c->d_group_est);
    float est[kMaxGroups];
    HIP_TRY(c, hipMemcpyAsync(est, c->d_group_est, sizeof(float) * ng, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t order[kMaxGroups];
    for (uint32_t k = 0; k < (uint32_t)kMaxGroups; ++k) order[k] = k;
    std::stable_sort(order, order + ng, [&](uint32_t x, uint32_t y) { return est[x] > est[y]; });
    for (uint32_t k = 0; k < (uint32_t)kMaxGroups; ++k) c->group_order[k] = (uint8_t)order[k];
    double sum = 0.0, mx = 0.0;
    for (uint32_t k = 0; k < ng; ++k) { sum += est[k]; mx = std::max(mx, (double)est[k]); }
    c->order_uneven = mx * ng > 1.3 * sum;                       // some group costs well above the mean
    std::memcpy(c->est_R, R9, sizeof(c->est_R)); std::memcpy(c->est_t, t3, sizeof(c->est_t));
    c->order_valid = true; c->est_launch = (int64_t)c->seq;
    return DCREG_OK;
}

// gated = true: a single-pose launch whose pose arrives later through the gate (R9, t3 ignored; dcreg_linearize_gate_open /
// _gate_abort decide its fate)
static int linearize_begin(dcreg_ctx *c, int slot, int n_poses, const double *R9, const double *t3, const int32_t *state_ids,
                           const dcreg_lin_params *p, dcreg_lin_debug *dbg_host, bool gated = false) {
    if (!c) return DCREG_E_INVALID;
    // a timing probe, not a dump (dcreg_debug.h dcreg_lin_debug::stamps): certificates in use, only the stamps come back
    const bool stamps_only = dbg_host && dbg_host->stamps && !dbg_host->nn_idx && !dbg_host->nn_d2 && !dbg_host->flag && !dbg_host->normal &&
                             !dbg_host->r && !dbg_host->s && !dbg_host->stats;
    // nothing may be queued behind a gate that still waits: it would wait with it
    if (c->gate_slot >= 0) { c->fail("a gated linearisation still waits for its pose (dcreg_linearize_gate_open / _gate_abort first)"); return DCREG_E_STATE; }
    if (gated) {
        static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
        R9 = eye; t3 = zero; n_poses = 1; state_ids = nullptr; dbg_host = nullptr;
        // results must arrive through the pinned flags: a stream synchronise would wait for the gate, i.e. for its own caller
        if (!c->opt_spin) { c->fail("gated launches need the \"spin\" option (results through pinned flags)"); return DCREG_E_STATE; }
        // A launch that is to be TIMED (option "time_kernels": HIP events around it) is not gated - the caller starts it the plain way
        // once its pose exists.  Measured (profiles/r04_events_vs_trace.md): the event in front of a gated linearisation is stamped when
        // the gate kernel STARTS, and that can be long before the previous linearisation ends (its last blocks leave most of the device
        // idle): events reported 324 us for a launch the kernel trace gives 228 us.
        if (c->opt_time_kernels > 0 && (c->launch_counter % (uint64_t)c->opt_time_kernels) == 0) {
            c->fail("the next launch is a timed one: not gated"); return DCREG_E_STATE;
        }
    }
    if (slot < 0 || slot >= dcreg_ctx::kLinSlots) { c->fail("invalid slot"); return DCREG_E_INVALID; }
    LinSlot &S = c->slots[slot];
    if (S.pending) { c->fail("slot %d still has a linearisation in flight", slot); return DCREG_E_STATE; }
    if (!R9 || !t3 || n_poses < 1) { c->fail("null argument"); return DCREG_E_INVALID; }
    if (n_poses > 65535) { c->fail("at most 65535 poses per batched launch (grid.y limit), got %d", n_poses); return DCREG_E_INVALID; }
    if (c->n_tgt <= 0) { c->fail("KdTree/target index is not set up in context"); return DCREG_E_STATE; }   // :1639
    if (c->n_src <= 0) { c->fail("measure cloud is not set"); return DCREG_E_STATE; }
    if (c->need_set_device) { HIP_TRY(c, hipSetDevice(c->device)); }     // (before make_lin_args: the Euler branch allocates and copies)
    // which index the launch searches (context.hpp, the window of a large map): single-pose product launches the window around their pose, a
    // gated launch whatever is active (its pose is checked when the gate opens), everything else the whole map
    if (!gated && (c->roi_active || roi_wanted(c))) {
        if (!p) { c->fail("null argument"); return DCREG_E_INVALID; }
        const bool product = n_poses == 1 && !state_ids && (!dbg_host || stamps_only);
        const int rr = product ? roi_ensure(c, R9, t3, p->search_radius) : roi_deactivate(c);
        if (rr) return rr;
    }
    LinArgs a;
    int rc = make_lin_args(c, p, a);
    if (rc) return rc;
    // the parameters the stored certificates, gate bits and planes depend on (context.hpp StateKey): a launch with other values
    // starts from empty states
    dcreg_ctx::StateKey key;
    key.radius_sq = a.radius_sq; key.max_thick_sq = a.max_thick_sq; key.min_norm = a.min_norm;
    key.radius_sq_f = a.radius_sq_f; key.cert_r_out = a.cert_r_out; key.cert_r_in = a.cert_r_in; key.fast_plane = c->opt_fast_plane ? 1 : 0;
    const uint32_t nbx = blocks_for(c->n_src, kLinBlock);
    if (ensure(c, S.d_partials, S.partials_cap, (size_t)n_poses * nbx * kSlots)) return DCREG_E_NOMEM;
    // one pose: the kernels finish the reduction themselves (chunk rows -> pinned memory); many poses: k_finalize
    const uint32_t n_chunks = (nbx + kChunk - 1) / kChunk;
    // ... and so do batches whose poses are one chunk each (the Monte-Carlo batches: 30 blocks per pose): the last block of a pose sums
    // the pose's rows - block_sum_rows, as k_finalize would - and publishes the pose's result row
    const bool fused = (n_poses == 1) || (n_chunks == 1 && c->opt_fused_batches);
    // a launch of one chunk's worth of blocks: the blocks publish their rows themselves and the host adds them (kernels.hpp FinArgs)
    const bool direct = fused && n_poses == 1 && nbx <= (uint32_t)kChunk && c->opt_direct_rows && c->opt_one_wave < 2;     // (one_wave = 2: tile rows even there - tests)
    const size_t n_rows = direct ? (size_t)nbx : (fused ? (size_t)n_chunks * (size_t)n_poses : (size_t)n_poses);      // result rows the host waits for
    if (fused) {   // tickets: zero when (re)allocated, afterwards every completed launch leaves them zero
        const size_t had = S.tickets_cap;
        if (ensure(c, S.d_tickets, S.tickets_cap, (size_t)n_chunks * (size_t)n_poses * kCounterStride)) return DCREG_E_NOMEM;
        if (S.tickets_cap != had || S.tickets_dirty) {
            HIP_TRY(c, hipMemsetAsync(S.d_tickets, 0, sizeof(unsigned int) * S.tickets_cap, c->stream));
            S.tickets_dirty = false;
        }
    }
    if (n_rows > S.out_cap) {
        if (S.h_out) (void)hipHostFree(S.h_out);
        S.h_out = nullptr; S.out_cap = 0;
        const size_t cap = std::max<size_t>(n_rows, 64);
        HIP_TRY(c, hipHostMalloc((void **)&S.h_out, cap * kSlots * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(S.h_out, 0, cap * kSlots * sizeof(double));
        std::fill(S.row_chk.begin(), S.row_chk.end(), 0ull);
        HIP_TRY(c, hipHostGetDevicePointer((void **)&S.d_out, S.h_out, 0));
        S.out_cap = cap;
    }
    const int64_t n = c->n_src;
    a.state = nullptr; a.state_stride = 0;
    a.xcd_chunk = (uint32_t)std::max(c->opt_xcd_chunk, 0);
    if (n_poses == 1 && !state_ids && !gated) {
        // how far this pose may carry a point from where the last linearised pose had it
        double jump = 1e300;
        if (c->last_pose_valid) {
            double dr = 0.0, dt = 0.0;
            for (int k = 0; k < 9; ++k) dr += (R9[k] - c->last_R[k]) * (R9[k] - c->last_R[k]);
            for (int k = 0; k < 3; ++k) dt += (t3[k] - c->last_t[k]) * (t3[k] - c->last_t[k]);
            jump = std::sqrt(dr) * c->src_radius + std::sqrt(dt);
        }
        if (c->hint_unknown && jump <= 0.5 * c->grid.h) c->hint_misalign = c->hint_last;      // the same trajectory, continued
        // ... and what the last completed launch searched says nothing about a launch half a cell or more away from it (the first
        // launch of a new run): expect a search of everything - for this launch and for the one that may be queued behind it before
        // its counts are known (scheduling only: which kernels carry the launch out)
        if (jump > 0.5 * c->grid.h) { c->last_searched = n_poses * c->n_src; c->last_points = c->last_searched; }
        c->hint_unknown = false;
        std::memcpy(c->last_R, R9, sizeof(c->last_R)); std::memcpy(c->last_t, t3, sizeof(c->last_t));
        c->last_pose_valid = true;
    }
    {   // heavy groups first: single-pose launches of at least two groups, once the order has been estimated (a gated launch does not
        // know its pose yet: it uses the order there is)
        // (a launch whose blocks are all resident at once - 4 per CU - has no order to speak of)
        const bool wanted = c->opt_dispatch_order && n_poses == 1 && !state_ids && c->n_groups >= 2 && nbx > 4u * (uint32_t)c->n_cus &&
                            c->hint_misalign > 0.5 * c->grid.h;
        if (wanted && !gated) {
            // the estimate holds for poses near the one it was made at (within a cell for every point)
            bool stale = !c->order_valid;
            // (counted in c->seq, the launch number that is never reset: round 6 found this test dead - it compared with n_launches, which
            //  dcreg_launch_stats_get(reset) zeroes, so that after one reset of the statistics the order of the moment was kept for good; with
            //  `bench.py --steps 20 --warmup 5` that was the order estimated at an ALIGNED pose, i.e. none: 355 / 314 / 260 us for the first
            //  launches of every later run instead of 264 / 216 / 205)
            if (!stale && (int64_t)c->seq - c->est_launch >= 16) {       // (a loop of ungated launches does not re-estimate at every step)
                double dr = 0.0, dt = 0.0;
                for (int k = 0; k < 9; ++k) dr += (R9[k] - c->est_R[k]) * (R9[k] - c->est_R[k]);
                for (int k = 0; k < 3; ++k) dt += (t3[k] - c->est_t[k]) * (t3[k] - c->est_t[k]);
                stale = std::sqrt(dr) * c->src_radius + std::sqrt(dt) > c->grid.h;
            }
            if (stale) {
                rc = estimate_dispatch_order(c, R9, t3, a.max_ring);
                if (rc) return rc;
            }
        }
        const bool ordered = wanted && c->order_valid && c->order_uneven;
        a.group_blocks = std::max<uint32_t>(c->group_blocks, 1u); a.n_groups = ordered ? c->n_groups : 0u;
        std::memcpy(a.group_order, c->group_order, sizeof(a.group_order));
    }
    PoseArg one{};
    one.state = kNoIdx; one.fresh = 1;
    const PoseArg *d_poses = nullptr;
    bool uses_state = false;
    if (n_poses == 1 && !state_ids && c->opt_warm && !(key == c->state_key)) { c->state_valid = false; c->state_key = key; }
    const bool state_was_valid = c->state_valid;
    if (n_poses == 1 && !state_ids) {
        std::memcpy(one.R, R9, sizeof(one.R)); std::memcpy(one.t, t3, sizeof(one.t));
        if (c->opt_warm) {      // single pose: the ctx's own state
            const size_t stride = ((size_t)n + 63) & ~(size_t)63;
            if (!c->state_valid || c->state_stride != stride) {
                if (ensure(c, c->d_state, c->state_cap, kStateRows * stride)) return DCREG_E_NOMEM;
                c->state_stride = stride; c->state_valid = false;
            }
            uses_state = true;
            one.state = 0; one.fresh = c->state_valid ? 0u : 1u;
            a.state = c->d_state; a.state_stride = (uint32_t)c->state_stride;
        }
        if (gated) {
            if (!c->h_gate) {     // all or nothing: a half-built gate would be taken for a whole one by the next call
                GateHost *hg = nullptr; PoseArg *dp = nullptr; uint32_t *da = nullptr; GateDev *gd = nullptr;
                bool ok = hipHostMalloc((void **)&hg, sizeof(GateHost), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
                ok = ok && hipMalloc((void **)&dp, sizeof(PoseArg)) == hipSuccess && hipMalloc((void **)&da, sizeof(uint32_t)) == hipSuccess;
                // (the device copy of the record, for launches gated in their first kernel: cleared IN THE STREAM - the ctx stream does not
                //  synchronise with the null stream)
                ok = ok && hipMalloc((void **)&gd, sizeof(GateDev)) == hipSuccess && hipMemsetAsync(gd, 0, sizeof(GateDev), c->stream) == hipSuccess;
                void *dgh = nullptr;
                ok = ok && hipHostGetDevicePointer(&dgh, hg, 0) == hipSuccess;
                if (!ok) {
                    if (hg) (void)hipHostFree(hg);
                    if (dp) (void)hipFree(dp);
                    if (da) (void)hipFree(da);
                    if (gd) (void)hipFree(gd);
                    c->state_valid = state_was_valid;
                    c->fail("allocating the launch gate failed");
                    return DCREG_E_NOMEM;
                }
                std::memset(hg, 0, sizeof(GateHost));
                c->h_gate = hg; c->d_gate_host = (GateHost *)dgh; c->d_gate_pose = dp; c->d_gate_abort = da; c->d_gate_dev = gd;
            }
            d_poses = c->d_gate_pose;
        }
    } else {
        // batched poses: each may own one of the reserved states (dcreg_reserve_warm_states); -1 = search cold, keep nothing
        const bool use_states = state_ids && c->opt_warm && c->n_batch_states > 0;
        if (use_states && !(key == c->batch_state_key)) {
            std::fill(c->batch_state_valid.begin(), c->batch_state_valid.end(), (uint8_t)0);
            c->batch_state_key = key;
        }
        if (state_ids && c->opt_warm) {
            std::vector<uint8_t> seen((size_t)std::max<int64_t>(c->n_batch_states, 1), 0);
            for (int i = 0; i < n_poses; ++i) {
                const int32_t sid = state_ids[i];
                if (sid < 0) continue;
                if ((int64_t)sid >= c->n_batch_states) { c->fail("warm state %d was not reserved (dcreg_reserve_warm_states: %lld)", sid, (long long)c->n_batch_states); return DCREG_E_INVALID; }
                if (seen[(size_t)sid]) { c->fail("warm state %d is used by two poses of one launch", sid); return DCREG_E_INVALID; }
                seen[(size_t)sid] = 1;
            }
        }
        const size_t bytes = (size_t)n_poses * sizeof(PoseArg);
        if (bytes > S.poses_cap) {      // the pinned block stays alive until end(): source of the asynchronous copy
            if (S.h_poses) (void)hipHostFree(S.h_poses);
            if (S.d_poses) (void)hipFree(S.d_poses);
            S.h_poses = nullptr; S.d_poses = nullptr; S.poses_cap = 0;
            const size_t cap = std::max<size_t>(bytes, 256 * sizeof(PoseArg));
            HIP_TRY(c, hipHostMalloc((void **)&S.h_poses, cap, hipHostMallocDefault));
            if (hipMalloc((void **)&S.d_poses, cap) != hipSuccess) { c->fail("hipMalloc(%zu B) failed", cap); return DCREG_E_NOMEM; }
            S.poses_cap = cap;
        }
        PoseArg *hp = (PoseArg *)S.h_poses;
        for (int i = 0; i < n_poses; ++i) {
            std::memcpy(hp[i].R, R9 + 9 * i, sizeof(one.R)); std::memcpy(hp[i].t, t3 + 3 * i, sizeof(one.t));
            const bool has = use_states && state_ids[i] >= 0;
            hp[i].state = has ? (uint32_t)state_ids[i] : kNoIdx;
            hp[i].fresh = (has && c->batch_state_valid[(size_t)state_ids[i]] != 0) ? 0u : 1u;
            if (has) c->batch_state_valid[(size_t)state_ids[i]] = 1;
        }
        // (measured and dropped, profiles/r05_ablation.md: the upload on a copy stream behind an event - the cross-stream dependency costs
        //  more than the 4.6 us copy it hides, C5 1.56 M against 1.64 M it/s - and no upload at all, the kernel reading the pinned block:
        //  +2.7 us per kernel, nothing gained - the experiment is bound by its kernels, two groups of trials alternating on the device)
        HIP_TRY(c, hipMemcpyAsync(S.d_poses, S.h_poses, bytes, hipMemcpyHostToDevice, c->stream));
        d_poses = (const PoseArg *)S.d_poses;
        if (use_states) { a.state = c->d_state_batch; a.state_stride = (uint32_t)c->state_batch_stride; }
    }
    a.use_cert = ((dbg_host && !stamps_only) || !c->opt_use_cert) ? 0 : 1;   // a debug dump searches every point (its statistics are those of the searches)
    a.search_count = c->opt_count_searches ? c->d_search_count : nullptr;
    // The advance pass (kernels.hpp k_advance) in front of this launch?  Only where it can pay: a single pose on the ctx's own state,
    // certificates in use, the state filled by an earlier launch; by default ("advance" = 1) a cloud whose query blocks exceed what the
    // device holds at once (below that a launch lasts as long as one search whether its searches are dense or not) in a launch expected
    // to search a few per cent of its points - expected = what the last completed launch reported (with the engines' pipeline that is
    // the launch two back).  Scheduling only: the sums are the same with and without the pass.
    bool adv = false, team = false;
    a.adv_counts = nullptr;
    if (n_poses == 1 && !state_ids && uses_state && one.fresh == 0u && a.use_cert && !dbg_host && a.count_scale != 0.0 && a.warm) {
        // what the launch is expected to search: what the last completed launch reported (with the engines' pipeline: the launch two back)
        const bool known = c->last_searched >= 0 && c->last_points == n;
        const double f = known ? (double)c->last_searched / (double)n : -1.0;
        adv = c->opt_advance >= 2 || (c->opt_advance == 1 && nbx >= (uint32_t)c->opt_advance_min_blocks && known && f >= c->opt_advance_lo && f <= c->opt_advance_hi);
        // ... its small-frame form (k_advance_team: sixteen lanes per query, one wave per kTeamTile points): a frame of a few thousand
        // points, whose query blocks leave most of the device idle, in a launch expected to search most of its points (the pass costs
        // ~12 us whatever it finds to do; k_lin alone gets cheaper as its searching waves thin out, so below half it wins again)
        // ... and only against a map whose occupied cells hold several points each: there a query with a loose bound (an OUT point, a
        // jump) has a hundred and more candidates in its 27-cell block and the lock-step search, two trips in flight, is at its slowest;
        // against a sparse map (the 7.5 k-point fixture: 1.6 points per occupied cell) it is as fast as the teams
        const double per_cell = (double)c->n_tgt / (double)std::max<uint32_t>(c->occupied_cells, 1u);
        team = c->opt_team_pass >= 2 || (c->opt_team_pass == 1 && n <= (int64_t)c->opt_team_pass_max_points && known && f >= c->opt_team_pass_min_frac &&
                                         per_cell >= c->opt_team_pass_min_cell_pts);
    }
    if (team) adv = false;
    // the one-wave instantiation of k_lin (kernels.hpp): fused single-pose launches of many blocks in which most waves search
    bool one_wave = false;
    if (n_poses > 1) {      // batched launches of one-chunk poses (the Monte-Carlo batches): trials at every stage of their runs side by side
        one_wave = fused && !dbg_host && (c->opt_one_wave >= 2 || (c->opt_one_wave_batches && (uint64_t)n_poses * nbx >= (uint64_t)c->opt_one_wave_min_blocks));
    } else
    if (n_poses == 1 && fused && !direct && !dbg_host && !stamps_only && !team) {
        // by the rule: a launch of many blocks whose searches are long - the queries more than a cell and a half from the surface (the
        // engines' hint, as for the dispatch order above; unknown at the start of a run = far) - and most of whose points search
        const bool known = c->last_searched >= 0 && c->last_points == n;
        const bool most = !uses_state || one.fresh != 0u || !a.use_cert || (known && (double)c->last_searched >= c->opt_one_wave_min_frac * (double)n);
        one_wave = c->opt_one_wave >= 2 || (c->opt_one_wave == 1 && nbx >= (uint32_t)c->opt_one_wave_min_blocks && !adv && most &&
                                            c->hint_misalign > c->opt_one_wave_min_cells * c->grid.h);
    }
    // (a row per TILE then: grown here, before any kernel of this launch is queued)
    if (one_wave && ensure(c, S.d_partials, S.partials_cap, (size_t)n_poses * nbx * kSlots * (kLinBlock / kWave))) return DCREG_E_NOMEM;
    const uint32_t n_tiles = team ? blocks_for(n, kTeamTile) : blocks_for(n, kAdvTile);
    if (adv || team) {      // the passes' counts, per query block of k_lin: zero between launches (k_lin takes them and zeroes them again)
        const size_t had = c->adv_counts_cap;
        if (ensure(c, c->d_adv_counts, c->adv_counts_cap, (size_t)kCounterStride * nbx)) return DCREG_E_NOMEM;      // (a 128-byte line per query block)
        if (c->adv_counts_cap != had || c->adv_counts_dirty) {
            HIP_TRY(c, hipMemsetAsync(c->d_adv_counts, 0, sizeof(uint32_t) * c->adv_counts_cap, c->stream));
            c->adv_counts_dirty = false;
        }
        a.adv_counts = c->d_adv_counts;
    }
    unsigned long long *team_stamps = nullptr;
    if (team && c->opt_team_stamps) {          // timing probe (dcreg_debug.h dcreg_team_pass_stamps): eight clock words per block of the pass.  Grown HERE,
        // before any kernel of this launch is queued: a hipFree behind a gate kernel that waits for the host would wait for it
        if (ensure(c, c->d_team_stamps, c->team_stamps_cap, (size_t)8 * (n_tiles + 1))) return DCREG_E_NOMEM;
        HIP_TRY(c, hipMemsetAsync(c->d_team_stamps, 0, sizeof(unsigned long long) * 8 * (n_tiles + 1), c->stream));
        team_stamps = c->d_team_stamps; c->team_stamps_n = n_tiles;
    }
    DebugDev dd{};
    free_tmp(S);
    if (dbg_host) {
        bool oom = false;
        auto alloc = [&](size_t bytes, int fill) -> void * {
            void *p2 = nullptr;
            if (hipMalloc(&p2, bytes) != hipSuccess) { oom = true; return nullptr; }
            (void)hipMemsetAsync(p2, fill, bytes, c->stream);
            S.tmp_dev.push_back(p2);
            return p2;
        };
        if (dbg_host->nn_idx) dd.nn_idx = (int32_t *)alloc(sizeof(int32_t) * 5 * n, 0xFF);
        if (dbg_host->nn_d2) dd.nn_d2 = (float *)alloc(sizeof(float) * 5 * n, 0);
        if (dbg_host->flag) dd.flag = (uint8_t *)alloc(n, 0);
        if (dbg_host->normal) dd.normal = (double *)alloc(sizeof(double) * 3 * n, 0);
        if (dbg_host->r) dd.r = (double *)alloc(sizeof(double) * n, 0);
        if (dbg_host->s) dd.s = (double *)alloc(sizeof(double) * n, 0);
        if (dbg_host->stats) dd.stats = (uint32_t *)alloc(sizeof(uint32_t) * n, 0);
        if (dbg_host->stamps) dd.stamps = (unsigned long long *)alloc(sizeof(unsigned long long) * 8 * (kLinBlock / 64) * (size_t)nbx, 0);
        if (oom) { free_tmp(S); drop_warm(c); c->fail("hipMalloc of the debug dump buffers failed"); return DCREG_E_NOMEM; }
    }
    const unsigned long long seq = ++c->seq;
    FinArgs fin{S.d_tickets, S.d_out, seq, direct ? 1u : 0u, n_chunks};
    if (fused) S.tickets_dirty = true;    // cleared again once this launch is known to have completed
    // kernel timing: HIP events around every opt_time_kernels-th linearisation (each timed launch costs ~10 us of host time)
    bool timed = c->opt_time_kernels > 0 && (c->launch_counter++ % (uint64_t)c->opt_time_kernels) == 0;
    if (timed && !S.ev0) timed = false;               // (the slot's events are created with the context)
    const uint32_t *abort_flag = nullptr;
    // a launch that was queued and must not run after all (errors below): call the gate off, forget what the states were about to hold
    auto bail = [&](const char *what, hipError_t e) {
        free_tmp(S);
        if (gated) gate_call_off(c);
        drop_warm(c);
        c->fail("%s failed: %s", what, hipGetErrorString(e));
        return DCREG_E_DEVICE;
    };
    // A gated launch the device holds at once (at most kChunk query blocks) waits for its pose in its FIRST kernel instead of behind a
    // gate kernel (kernels.hpp gate_wait: one kernel boundary less on a launch that lasts a few microseconds); larger launches keep
    // k_gate (every wave would pay for the wait: profiles/r04_gate_in_kernel.txt).  What this relies on and costs: every
    // wave of the gated kernel - at most 64 x 4 of k_lin, at most 4096 one-wave blocks of the teams - is resident and polls the DEVICE copy
    // of the record while the host takes its step; only the first wave of block 0 polls the host record and fills the copy, so progress
    // rests on block 0 being dispatched before the device is full, which holds while blocks are dispatched in index order and the launch
    // fits the device (nbx <= kChunk: it does).  A context that shares its device with other busy contexts holds wave slots idle that way;
    // option "gate_in_kernel" = 0 puts the one-wave k_gate back in front.
    GateArgs gt{nullptr, nullptr, 0ull, nullptr, nullptr, 0u};
    const bool gate_inside = gated && c->opt_gate_in_kernel && nbx <= (uint32_t)kChunk && !adv && !dbg_host;
    const PoseArg *lin_poses = d_poses;            // what k_lin reads its pose from (null: pose1)
    if (gated) {
        const unsigned long long want = ++c->gate_seq;
        if (gate_inside) {
            gt = GateArgs{c->d_gate_host, c->d_gate_dev, want, team ? c->d_gate_pose : nullptr, team ? c->d_gate_abort : nullptr, one.fresh};
            if (team) abort_flag = c->d_gate_abort;        // (k_lin behind the teams reads what their polling wave left, as behind k_gate)
            else lin_poses = nullptr;                      // (k_lin itself is the gated kernel: pose1 carries state / fresh)
        } else {
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream, c->d_gate_host, want, c->d_gate_pose, one.fresh, c->d_gate_abort);
            abort_flag = c->d_gate_abort;
        }
    }
    if (timed) {                           // after the gate: the events bracket the linearisation, not the wait for the pose
        const hipError_t ee = hipEventRecord(S.ev0, c->stream);
        if (ee != hipSuccess) return bail("hipEventRecord", ee);
    }
    const bool fast = c->opt_fast_plane;
    if (adv) {
        if (fast) hipLaunchKernelGGL((k_advance<true>), dim3(n_tiles), dim3(kLinBlock), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, d_poses, a, c->d_adv_counts, abort_flag);
        else hipLaunchKernelGGL((k_advance<false>), dim3(n_tiles), dim3(kLinBlock), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, d_poses, a, c->d_adv_counts, abort_flag);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return bail("advance pass launch", le);
        c->n_advance_launches += 1;
    }
    if (team) {
        unsigned long long *stamps = team_stamps;
        if (gate_inside) {      // the teams are the gated kernel
            if (fast) hipLaunchKernelGGL((k_advance_team<true, true>), dim3(n_tiles), dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, (const PoseArg *)nullptr, a, c->d_adv_counts, (const uint32_t *)nullptr, stamps, gt);
            else hipLaunchKernelGGL((k_advance_team<false, true>), dim3(n_tiles), dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, (const PoseArg *)nullptr, a, c->d_adv_counts, (const uint32_t *)nullptr, stamps, gt);
        } else {
            if (fast) hipLaunchKernelGGL((k_advance_team<true>), dim3(n_tiles), dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, d_poses, a, c->d_adv_counts, abort_flag, stamps, gt);
            else hipLaunchKernelGGL((k_advance_team<false>), dim3(n_tiles), dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, d_poses, a, c->d_adv_counts, abort_flag, stamps, gt);
        }
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return bail("team advance pass launch", le);
        c->n_advance_launches += 1;
    }
    {
        const dim3 grid(nbx, (unsigned)n_poses);
#define DCREG_LAUNCH_LIN(MODE, FUSED, FAST)                                                                                              \
    hipLaunchKernelGGL((k_lin<MODE, FUSED, FAST>), grid, dim3(kLinBlock), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, lin_poses, a,  \
                       S.d_partials, nbx, fin, dd, abort_flag, gt)
        if (gate_inside && !team) {        // k_lin is the gated kernel (fused, MODE 0: a gated launch is never a dump)
            if (fast) hipLaunchKernelGGL((k_lin<0, true, true, true>), grid, dim3(kLinBlock), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, lin_poses, a, S.d_partials, nbx, fin, dd, abort_flag, gt);
            else hipLaunchKernelGGL((k_lin<0, true, false, true>), grid, dim3(kLinBlock), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, lin_poses, a, S.d_partials, nbx, fin, dd, abort_flag, gt);
        } else
        if (stamps_only) { if (fast) DCREG_LAUNCH_LIN(2, true, true); else DCREG_LAUNCH_LIN(2, true, false); }     // (the probe writes the shared state: same fit as the plain launches)
        else if (dbg_host) { if (fast) DCREG_LAUNCH_LIN(1, true, true); else DCREG_LAUNCH_LIN(1, true, false); }
        else if (one_wave) {               // one-wave blocks: a grid of tiles
            const dim3 tiles(nbx * (kLinBlock / kWave), (unsigned)n_poses);
            if (fast) hipLaunchKernelGGL((k_lin<0, true, true, false, true>), tiles, dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, lin_poses, a, S.d_partials, nbx, fin, dd, abort_flag, gt);
            else hipLaunchKernelGGL((k_lin<0, true, false, false, true>), tiles, dim3(kWave), 0, c->stream, c->d_src, (uint32_t)n, c->grid, one, lin_poses, a, S.d_partials, nbx, fin, dd, abort_flag, gt);
            hipLaunchKernelGGL(k_sum_tiles, dim3(n_chunks, (unsigned)n_poses), dim3(kLinBlock), 0, c->stream, S.d_partials, nbx, S.d_out, seq, abort_flag);
        }
        else if (fused) { if (fast) DCREG_LAUNCH_LIN(0, true, true); else DCREG_LAUNCH_LIN(0, true, false); }
        else { if (fast) DCREG_LAUNCH_LIN(0, false, true); else DCREG_LAUNCH_LIN(0, false, false); }
#undef DCREG_LAUNCH_LIN
    }
    {   // an invalid launch (bad grid, too many resources) must surface here, not as a spin timeout in end()
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return bail("linearisation kernel launch", le);
    }
    if (!fused) {
        hipLaunchKernelGGL(k_finalize, dim3((unsigned)n_poses), dim3(kLinBlock), 0, c->stream, S.d_partials, nbx, S.d_out, seq);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return bail("k_finalize launch", le);
    }
    if (timed) {                           // brackets the linearisation's kernels
        const hipError_t ee = hipEventRecord(S.ev1, c->stream);
        if (ee != hipSuccess) return bail("hipEventRecord", ee);
    }
    if (dbg_host) {
        hipError_t ce = hipSuccess;
        auto back = [&](void *dst, const void *srcp, size_t bytes) { if (ce == hipSuccess && srcp) ce = hipMemcpyAsync(dst, srcp, bytes, hipMemcpyDeviceToHost, c->stream); };
        back(dbg_host->nn_idx, dd.nn_idx, sizeof(int32_t) * 5 * n);
        back(dbg_host->nn_d2, dd.nn_d2, sizeof(float) * 5 * n);
        back(dbg_host->flag, dd.flag, (size_t)n);
        back(dbg_host->normal, dd.normal, sizeof(double) * 3 * n);
        back(dbg_host->r, dd.r, sizeof(double) * n);
        back(dbg_host->s, dd.s, sizeof(double) * n);
        back(dbg_host->stats, dd.stats, sizeof(uint32_t) * n);
        back(dbg_host->stamps, dd.stamps, sizeof(unsigned long long) * 8 * (kLinBlock / 64) * (size_t)nbx);
        if (ce != hipSuccess) {
            (void)hipStreamSynchronize(c->stream);
            return bail("copying the debug dump back", ce);
        }
    }
    c->n_launches += 1; c->n_poses_launched += n_poses; c->n_points_launched += (int64_t)n_poses * n;
    if (uses_state) c->state_valid = true;          // once this launch has run, the state holds a search of the current clouds
    S.pending = true; S.n_poses = n_poses; S.n_chunks = n_chunks; S.n_rows = n_rows; S.fused = fused; S.direct = direct; S.timed = timed;
    S.seq = seq; S.sync = dbg_host != nullptr;
    S.stamps_only = stamps_only;
    S.advanced = (adv ? 1 : (team ? 2 : 0)) | (one_wave ? 4 : 0);
    S.coded = a.count_scale != 0.0;        // how THIS launch's count slots are to be read (the source may be replaced while it is pending)
    if (gated) {
        c->gate_slot = slot;
        c->gate_uses_state = uses_state; c->gate_state_was_valid = state_was_valid;
    }
    return DCREG_OK;
}

// wait for the result rows of a launch.  Hot path: spin on the rows the kernels publish into pinned host memory (check word).  A
// launch may be slow but healthy (a GPU shared with other processes, huge clouds), and while it is awaited a gate for the NEXT launch
// may already sit in the stream: a stream synchronise would then wait for that gate, i.e. for this very thread.  So the wait is
// bounded by wall-clock time (far below the gate's own patience); only when it runs out is the stream drained - after calling the
// waiting gate off - to surface a device fault instead of hanging.
static int wait_rows(dcreg_ctx *c, LinSlot &S) {
    using Clk = std::chrono::steady_clock;
    Clk::time_point t0;
    bool clocked = false;
    const uint64_t every = c->opt_wait_seconds < 1.0 ? ((1ull << 6) - 1) : ((1ull << 20) - 1);      // (sub-second patience: tests of this path)
    // a row has arrived when its check word fits the 31 values next to it (kernels.hpp publish_row).  The row is copied out in one
    // piece (plain loads behind a compiler barrier: a torn snapshot fails the check and is read again) and the snapshot that was
    // checked is the one that is used.
    static const struct Mults { unsigned long long m[kSlots]; Mults() { for (int k = 0; k < kSlots; ++k) m[k] = row_check_mult(k); } } mults;
    auto arrived = [&](size_t i) {
        unsigned long long v[kSlots];
        asm volatile("" ::: "memory");
        std::memcpy(v, (const void *)(S.h_out + i * kSlots), sizeof(v));
        unsigned long long c0 = S.seq * mults.m[31], c1 = 0, c2 = 0, c3 = 0;
        for (int k = 0; k < 28; k += 4) { c0 += v[k] * mults.m[k]; c1 += v[k + 1] * mults.m[k + 1]; c2 += v[k + 2] * mults.m[k + 2]; c3 += v[k + 3] * mults.m[k + 3]; }
        c0 += v[28] * mults.m[28]; c1 += v[29] * mults.m[29]; c2 += v[30] * mults.m[30];
        if (c0 + c1 + c2 + c3 != v[31]) return false;
        std::memcpy(S.h_rows.data() + i * kSlots, v, sizeof(v));
        S.row_chk[i] = v[31];
        return true;
    };
    // (cheap look first: a row whose check word is still the one taken last time has not been written again - the word carries the
    // launch number)
    auto touched = [&](size_t i) {
        return __atomic_load_n((const unsigned long long *)(S.h_out + i * kSlots) + 31, __ATOMIC_RELAXED) != S.row_chk[i];
    };
    S.h_rows.resize(S.n_rows * kSlots);
    // rows are taken in the order they arrive (the last one to come is then the only one left to check), not in index order
    S.row_done.assign(S.n_rows, 0);
    if (S.row_chk.size() < S.n_rows) S.row_chk.resize(S.n_rows, 0ull);
    size_t pending = S.n_rows;
    uint64_t spins = 0;
    while (pending) {
        for (size_t i = 0; i < S.n_rows; ++i)
            if (!S.row_done[i] && touched(i) && arrived(i)) { S.row_done[i] = 1; --pending; }
        if (!pending) break;
        __builtin_ia32_pause();
        if ((++spins & every) != 0) continue;
        if (!clocked) { t0 = Clk::now(); clocked = true; continue; }
        if (std::chrono::duration<double>(Clk::now() - t0).count() < c->opt_wait_seconds) continue;
        if (c->gate_slot >= 0) (void)dcreg_linearize_gate_abort(c);      // nothing may wait behind us while we drain the stream
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->fail("device fault while waiting for a linearisation: %s", hipGetErrorString(e)); return DCREG_E_DEVICE; }
        for (size_t i = 0; i < S.n_rows; ++i)
            if (!S.row_done[i]) {
                if (!arrived(i)) { c->fail("linearisation result never arrived"); return DCREG_E_DEVICE; }
                S.row_done[i] = 1; --pending;
            }
    }
    return DCREG_OK;
}

static int linearize_end(dcreg_ctx *c, int slot, dcreg_lin_out *outs) {
    if (!c) return DCREG_E_INVALID;
    if (slot < 0 || slot >= dcreg_ctx::kLinSlots) { c->fail("invalid slot"); return DCREG_E_INVALID; }
    LinSlot &S = c->slots[slot];
    if (!S.pending) { c->fail("slot %d has no linearisation in flight", slot); return DCREG_E_STATE; }
    if (!outs) { c->fail("null argument"); return DCREG_E_INVALID; }
    S.pending = false;
    int rc = DCREG_OK;
    if (S.sync || !c->opt_spin) {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->fail("hipStreamSynchronize failed: %s", hipGetErrorString(e)); rc = DCREG_E_DEVICE; }
        if (rc == DCREG_OK && S.sync) { const hipError_t e2 = hipGetLastError(); if (e2 != hipSuccess) { c->fail("%s", hipGetErrorString(e2)); rc = DCREG_E_DEVICE; } }
        if (rc == DCREG_OK) rc = wait_rows(c, S);       // (the stream is drained: the rows are there; this checks and snapshots them)
    } else {
        rc = wait_rows(c, S);
    }
    free_tmp(S);
    if (rc != DCREG_OK) { drop_warm(c); return rc; }       // what the launch left in the states is unknown
    S.tickets_dirty = false;   // every chunk published its row: all tickets are back to zero
    float launch_ms = -1.f;
    if (S.timed) {
        float ms = 0.f;
        hipError_t te = hipEventElapsedTime(&ms, S.ev0, S.ev1);
        if (te == hipErrorNotReady) { HIP_TRY(c, hipEventSynchronize(S.ev1)); te = hipEventElapsedTime(&ms, S.ev0, S.ev1); }
        HIP_TRY(c, te);
        c->kernel_ms_total += ms; c->kernel_launches += 1;
        launch_ms = ms;
    }
    double total[kSlots];
    if (S.direct) {  // the rows of the blocks of ONE chunk, added exactly as block_sum_rows adds them on the device (so a pose gives
                     // bitwise the same sums alone and in a batch): lane group g takes rows g, g + G, ..., then the G sums in order
        constexpr int G = kLinBlock / 32;
        for (int k = 0; k < 31; ++k) {
            double tot = 0.0;
            for (int g = 0; g < G; ++g) {
                double t = 0.0;
                for (size_t r = (size_t)g; r < S.n_rows; r += G) t += S.h_rows[r * kSlots + k];
                tot += t;
            }
            total[k] = 0.0 + tot;                            // (the chunk total enters a sum that starts at zero, as everywhere)
        }
    } else if (S.fused && S.n_poses == 1) {   // add the chunk rows in index order (fixed order: deterministic)
        for (int k = 0; k < 31; ++k) total[k] = 0.0;
        for (uint32_t ch = 0; ch < S.n_chunks; ++ch) {
            const double *row = S.h_rows.data() + (size_t)ch * kSlots;
            for (int k = 0; k < 31; ++k) total[k] += row[k];
        }
    }
    // the count slots carry two numbers each when the launch was asked to report what it did (LinArgs::count_scale): exact integers
    const bool coded = S.coded;
    int64_t searched = coded ? 0 : -1, refitted = coded ? 0 : -1;
    for (int i = 0; i < S.n_poses; ++i) {
        const double *o = (S.fused && S.n_poses == 1) ? total : S.h_rows.data() + (size_t)i * kSlots;      // (batches: one row per pose)
        std::memcpy(outs[i].H_upper, o, 21 * sizeof(double));
        std::memcpy(outs[i].g, o + 21, 6 * sizeof(double));
        outs[i].sum_r2 = o[27]; outs[i].sum_b2 = o[28];
        int64_t c29 = (int64_t)std::llround(o[29]), c30 = (int64_t)std::llround(o[30]);
        if (coded) {
            const int64_t S26 = (int64_t)1 << 26;
            searched += c29 / S26; refitted += c30 / S26;
            c29 %= S26; c30 %= S26;
        }
        outs[i].n_eff = c29; outs[i].n_pt = c30;
    }
    c->last_points = (int64_t)S.n_poses * c->n_src;
    // (a point whose new certificate has no slack at all - distance ties - is served by an advance pass and searched again by k_lin: it is
    // counted twice.  The scheduling rules and the launch log read fractions: never more than every point.  ADVICE round 5)
    if (searched > c->last_points) searched = c->last_points;
    if (refitted > c->last_points) refitted = c->last_points;
    c->last_searched = searched; c->last_refitted = refitted;
    if (c->opt_record_launches && c->launch_series.size() < ((size_t)1 << 20))
        c->launch_series.push_back(dcreg_ctx::LaunchRec{(double)launch_ms, searched, refitted, c->last_points, S.advanced});
    return DCREG_OK;
}

int launch_linearize(dcreg_ctx *c, int n_poses, const double *R9, const double *t3, const dcreg_lin_params *p,
                     dcreg_lin_out *outs, dcreg_lin_debug *dbg_host) {
    if (c && !outs) { c->fail("null argument"); return DCREG_E_INVALID; }
    static const bool timing = std::getenv("DCREG_HOST_TIMING") != nullptr;     // diagnostic: where a blocking call's host time goes
    using Clk = std::chrono::steady_clock;
    const auto t0 = Clk::now();
    int rc = linearize_begin(c, 0, n_poses, R9, t3, nullptr, p, dbg_host);
    if (rc) return rc;
    const auto t1 = Clk::now();
    rc = linearize_end(c, 0, outs);
    if (rc && c) c->slots[0].pending = false;
    if (timing && c) {
        static double s_begin = 0.0, s_end = 0.0, s_between = 0.0;
        static long n = 0;
        static Clk::time_point last;
        const auto t2 = Clk::now();
        if (n > 0) s_between += std::chrono::duration<double, std::micro>(t0 - last).count();
        s_begin += std::chrono::duration<double, std::micro>(t1 - t0).count();
        s_end += std::chrono::duration<double, std::micro>(t2 - t1).count();
        last = t2;
        if (++n % 100 == 0)
            std::fprintf(stderr, "[dcreg host timing] %ld blocking linearisations: enqueue %.2f us, wait for the result %.2f us, caller between calls %.2f us (means)\n",
                         n, s_begin / n, s_end / n, s_between / (n - 1));
    }
    return rc;
}

// k-NN of arbitrary queries (host pointer) or of the transformed source cloud (q == nullptr)
int launch_knn(dcreg_ctx *c, const GridDev &grid, const float4 *d_q, int64_t n, int k, double max_radius, const PoseArg *pose,
               int32_t *d_idx, float *d_d2, bool sweep) {
    float bound = INFINITY;
    int max_ring;
    if (max_radius > 0.0 && std::isfinite(max_radius)) {
        const double r2 = max_radius * max_radius;
        float rf = (float)r2; if ((double)rf < r2) rf = std::nextafterf(rf, INFINITY);
        bound = std::nextafterf(rf, INFINITY);
        int kk = 1;
        while (kk < 100000) { const double s = (double)kk * grid.h * (1.0 - 1e-9); if (s * s * (1.0 - 1e-6) >= (double)bound) break; ++kk; }
        max_ring = kk;
    } else {
        bound = 3.0e38f;
        max_ring = -1;   // unbounded: the kernel sweeps as many rings as the grid needs
    }
    PoseArg P{};
    if (pose) P = *pose;
    if (sweep) {       // the linearisation's way of covering what lies beyond the 27-cell block, in the plain kernel (k = 5, bounded)
        if (k != 5 || max_ring < 0 || !grid.ymask) { c->fail("the row sweep needs k = 5, a radius and the target grid"); return DCREG_E_INVALID; }
        hipLaunchKernelGGL((k_knn<5, true>), dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, d_q, (uint32_t)n, grid, bound, max_ring, P, pose ? 1 : 0, d_idx, d_d2);
        HIP_TRY(c, hipGetLastError());
        return DCREG_OK;
    }
    if (k == 1)
        hipLaunchKernelGGL(k_knn<1>, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, d_q, (uint32_t)n, grid, bound, max_ring, P, pose ? 1 : 0, d_idx, d_d2);
    else
        hipLaunchKernelGGL(k_knn<5>, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, d_q, (uint32_t)n, grid, bound, max_ring, P, pose ? 1 : 0, d_idx, d_d2);
    HIP_TRY(c, hipGetLastError());
    return DCREG_OK;
}

}  // namespace dcreg

using namespace dcreg;

void dcreg_ctx::fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, sizeof(err), fmt, ap);
    va_end(ap);
}

extern "C" {

int dcreg_backend_create(dcreg_ctx **out, int device) {
    if (!out) return DCREG_E_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        std::fprintf(stderr, "[dcreg] no usable HIP device (%s); this library has no CPU fallback\n",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return DCREG_E_DEVICE;
    }
    if (device < 0 || device >= count) return DCREG_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return DCREG_E_DEVICE;
    dcreg_ctx *c = new (std::nothrow) dcreg_ctx();
    if (!c) return DCREG_E_NOMEM;
    c->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = cus;
    }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->d_scratch, 256) != hipSuccess) {
        delete c;
        return DCREG_E_DEVICE;
    }
    c->stream = c->own_stream;
    for (LinSlot &S : c->slots)       // "time_kernels": the events of each slot's launch (created here, not in a timed region)
        if (hipEventCreate(&S.ev0) != hipSuccess || hipEventCreate(&S.ev1) != hipSuccess) { S.ev0 = S.ev1 = nullptr; }
    *out = c;
    return DCREG_OK;
}

void dcreg_backend_destroy(dcreg_ctx *c) {
    if (!c) return;
    (void)dcreg_comm_destroy(c);
    (void)hipSetDevice(c->device);
    (void)dcreg_linearize_gate_abort(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->h_gate) (void)hipHostFree(c->h_gate);
    if (c->h_euler) (void)hipHostFree(c->h_euler);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_stage_ev) (void)hipEventDestroy(c->h_stage_ev);
    if (c->d_euler) (void)hipFree(c->d_euler);
    if (c->d_gate_pose) (void)hipFree(c->d_gate_pose);
    if (c->d_gate_abort) (void)hipFree(c->d_gate_abort);
    if (c->d_gate_dev) (void)hipFree(c->d_gate_dev);
    if (c->d_group_est) (void)hipFree(c->d_group_est);
    kdtree_free(c->kd); c->kd = nullptr;
    void *bufs[] = {c->d_tgt_raw, c->d_tgt, c->d_src_raw, c->d_src, c->d_stage, c->d_keys, c->d_keys2, c->d_vals, c->d_vals2,
                    c->d_mkeys, c->d_mkeys2, c->d_cell_start, c->d_scratch, c->sort_tmp,
                    c->d_nn_idx, c->d_nn_d2, c->d_p2p_part, c->d_aligned, c->d_aux, c->d_aux_cell_start, c->d_state, c->d_state_batch, c->d_search_count, c->d_gap, c->d_ymask, c->d_owner,
                    c->d_adv_counts, c->d_team_stamps, c->roi_store.raw, c->roi_store.sorted, c->roi_store.cell_start, c->roi_store.gap,
                    c->roi_store.owner, c->roi_store.ymask};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (LinSlot &S : c->slots) {
        for (void *b : {(void *)S.d_partials, (void *)S.d_poses, (void *)S.d_tickets}) if (b) (void)hipFree(b);
        for (void *b : S.tmp_dev) (void)hipFree(b);
        if (S.ev0) (void)hipEventDestroy(S.ev0);
        if (S.ev1) (void)hipEventDestroy(S.ev1);
        if (S.h_out) (void)hipHostFree(S.h_out);
        if (S.h_poses) (void)hipHostFree(S.h_poses);
    }
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char *dcreg_last_error(const dcreg_ctx *c) { return c ? c->err : "null context"; }
// (dcreg_debug.h) the host translation units above the device seam leave their error text here
void dcreg_set_error_message(dcreg_ctx *c, const char *msg) { if (c && msg) c->fail("%s", msg); }

int dcreg_set_stream(dcreg_ctx *c, void *s) {
    if (!c) return DCREG_E_INVALID;
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return DCREG_OK;
}

int dcreg_set_option(dcreg_ctx *c, const char *key, double v) {
    if (!c || !key) return DCREG_E_INVALID;
    const std::string k(key);
    if (k == "cell") c->opt_cell = v;
    else if (k == "cell_factor") c->opt_cell_factor = v > 0.1 ? v : 2.0;
    else if (k == "use_certificates") c->opt_use_cert = v != 0.0;
    else if (k == "count_searches") {
        if (v != 0.0 && !c->d_search_count) {
            HIP_TRY(c, hipMalloc((void **)&c->d_search_count, kSearchCountBytes));
            HIP_TRY(c, hipMemsetAsync(c->d_search_count, 0, kSearchCountBytes, c->stream));
        }
        c->opt_count_searches = v != 0.0;
    }
    else if (k == "cert_margin" || k == "cert_inflate" || k == "warm_start" || k == "fast_plane_fit") {
        // options the neighbour states depend on: not while a queued launch may still write them (a later _gate_abort would also
        // restore state_valid, undoing the drop)
        if (c->gate_slot >= 0) { c->fail("a gated linearisation is queued: open or abort it before changing \"%s\"", key); return DCREG_E_STATE; }
        if (k == "cert_margin") c->opt_cert_margin = v >= 1e-4 ? std::min(v, 1.0) : 1e-4;    // the cells follow at the next dcreg_set_target
        else if (k == "cert_inflate") c->opt_cert_inflate = v >= 0.0 ? std::min(v, 1.0) : 0.0;
        else if (k == "warm_start") c->opt_warm = v != 0.0;
        else c->opt_fast_plane = v != 0.0;   // 1 (default) = plane_fit_qr_fast, 0 = the Eigen-shaped plane_fit_qr
        drop_warm(c);
    }
    else if (k == "wait_seconds") c->opt_wait_seconds = v > 0.0 ? v : 30.0;
    else if (k == "x_subdiv") { int sx = 1; while (sx < 16 && (double)(sx * 2) <= v) sx *= 2; c->opt_x_subdiv = sx; }
    else if (k == "time_kernels") { c->opt_time_kernels = v > 0.0 ? (int)v : 0; c->launch_counter = 0; }
    else if (k == "record_launches") { c->opt_record_launches = v != 0.0; if (v == 0.0) c->launch_series.clear(); }
    else if (k == "curve_x_scale") c->opt_curve_x_scale = (v > 0.0 && v <= 1.0) ? v : 1.0;   // next dcreg_set_source: patches of the curve order 1 / v times as long in x
    else if (k == "roi_index") { c->opt_roi_index = (int)std::min(std::max(v, 0.0), 2.0); c->roi_built = false; }   // 0 never, 1 auto, 2 always (next single-pose launch)
    else if (k == "roi_margin") { c->opt_roi_margin = std::min(std::max(v, 0.0), 1.0e6); c->roi_built = false; }
    else if (k == "max_table_entries") c->opt_max_table_entries = (int64_t)std::min(std::max(v, 1048576.0), 2147483648.0);   // next dcreg_set_target
    else if (k == "advance") c->opt_advance = (int)v;            // the advance pass in front of single-pose launches: 0 never, 1 (default) by the host's rule, 2 whenever possible
    else if (k == "one_wave") c->opt_one_wave = (int)v;                  // k_lin in one-wave blocks: 0 never, 1 by the rule (launches of many blocks in which most waves search), 2 wherever possible
    else if (k == "one_wave_batches") c->opt_one_wave_batches = v != 0.0;
    else if (k == "one_wave_min_frac") c->opt_one_wave_min_frac = v;
    else if (k == "one_wave_min_cells") c->opt_one_wave_min_cells = v;
    else if (k == "one_wave_min_blocks") c->opt_one_wave_min_blocks = (int)v;
    else if (k == "gate_in_kernel") c->opt_gate_in_kernel = v != 0.0;     // pipelined launches of at most 64 query blocks wait for their pose in their first kernel (1, default) or behind k_gate (0)
    else if (k == "team_stamps") c->opt_team_stamps = v != 0.0;   // timing probe of the small-frame pass (dcreg_team_pass_stamps)
    else if (k == "team_pass") c->opt_team_pass = (int)v;        // the small-frame advance pass: 0 never, 1 (default) by the host's rule, 2 whenever possible
    else if (k == "advance_min_blocks") c->opt_advance_min_blocks = (int)v;   // ... and the cloud has at least this many query blocks
    else if (k == "team_search") c->opt_team_max = (int)v;        // lanes a sparse wave serves cooperatively (0 = off, default 7)
    else if (k == "spin") c->opt_spin = v != 0.0;
    else if (k == "far_bound") c->opt_far_bound = v != 0.0;       // next dcreg_set_target: start bound of far queries from the nearest occupied cell
    else if (k == "dispatch_order") { c->opt_dispatch_order = v != 0.0; c->order_valid = false; }   // heavy query groups first (kernels.hpp k_group_cost)
    else if (k == "keep_source_order") c->opt_keep_source_order = v != 0.0;   // next dcreg_set_source: no Hilbert sort
    else if (k == "gap_field") c->opt_gap_field = v != 0.0;      // takes effect at the next dcreg_set_target
    else { c->fail("unknown option '%s'", key); return DCREG_E_INVALID; }
    return DCREG_OK;
}

int dcreg_set_target(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride, double r) { return set_target(c, xyz, n, stride, r, false); }
int dcreg_set_target_device(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride, double r) { return set_target(c, xyz, n, stride, r, true); }
int dcreg_set_source(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride) { return set_source(c, xyz, n, stride, false); }
int dcreg_set_source_device(dcreg_ctx *c, const float *xyz, int64_t n, int64_t stride) { return set_source(c, xyz, n, stride, true); }

int dcreg_default_lin_params(dcreg_lin_params *p, double radius) {
    if (!p) return DCREG_E_INVALID;
    std::memset(p, 0, sizeof(*p));            // parameterization = DCREG_PARAM_SO3
    p->search_radius = radius; p->max_plane_thickness_sq = 0.2 * 0.2; p->min_normal_norm = 1e-6;
    p->weight_slope = 0.9; p->weight_min = 0.1; p->use_weight_derivative = 0; p->k = 5;
    return DCREG_OK;
}

int dcreg_linearize(dcreg_ctx *c, const double R[9], const double t[3], const dcreg_lin_params *p, dcreg_lin_out *out) {
    return launch_linearize(c, 1, R, t, p, out, nullptr);
}
int dcreg_linearize_batch(dcreg_ctx *c, int n, const double *R9, const double *t3, const dcreg_lin_params *p, dcreg_lin_out *outs) {
    return launch_linearize(c, n, R9, t3, p, outs, nullptr);
}
int dcreg_linearize_batch_begin(dcreg_ctx *c, int slot, int n, const double *R9, const double *t3, const dcreg_lin_params *p) {
    return linearize_begin(c, slot, n, R9, t3, nullptr, p, nullptr);
}
int dcreg_linearize_batch_begin_warm(dcreg_ctx *c, int slot, int n, const double *R9, const double *t3, const int32_t *state_ids,
                                     const dcreg_lin_params *p) {
    return linearize_begin(c, slot, n, R9, t3, state_ids, p, nullptr);
}
int dcreg_reserve_warm_states(dcreg_ctx *c, int64_t n_states) {
    if (!c) return DCREG_E_INVALID;
    if (n_states < 0) { c->fail("negative state count"); return DCREG_E_INVALID; }
    for (const LinSlot &S : c->slots) if (S.pending) { c->fail("a linearisation is still in flight"); return DCREG_E_STATE; }
    c->n_batch_states = 0;
    c->batch_state_valid.clear();
    if (n_states == 0 || c->n_src <= 0) return DCREG_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t stride = ((size_t)c->n_src + 63) & ~(size_t)63;
    // nothing is cleared: a state is "fresh" (host-side flag) until its first launch has filled it
    if (ensure(c, c->d_state_batch, c->state_batch_cap, kStateRows * stride * (size_t)n_states)) return DCREG_E_NOMEM;
    c->state_batch_stride = stride;
    c->n_batch_states = n_states;
    c->batch_state_valid.assign((size_t)n_states, 0);
    return DCREG_OK;
}
int dcreg_hint_misalignment(dcreg_ctx *c, double metres) {
    if (!c) return DCREG_E_INVALID;
    if (metres >= 0.0) { c->hint_misalign = metres; c->hint_last = metres; c->hint_unknown = false; }
    else { c->hint_misalign = 1e300; c->hint_unknown = true; }            // (also NaN: no knowledge - resolved in linearize_begin)
    return DCREG_OK;
}
int dcreg_reset_warm_state(dcreg_ctx *c, int64_t state_id) {
    if (!c) return DCREG_E_INVALID;
    if (state_id == -1) {            // the context's own state (single-pose launches): as after dcreg_set_source - what a fresh ICPContext holds
        if (c->gate_slot >= 0) { c->fail("a gated linearisation is queued: open or abort it before dropping the neighbour state"); return DCREG_E_STATE; }
        for (const LinSlot &S : c->slots) if (S.pending) { c->fail("a linearisation is still in flight"); return DCREG_E_STATE; }
        c->state_valid = false;
        return DCREG_OK;
    }
    if (state_id < 0 || state_id >= c->n_batch_states) { c->fail("warm state %lld was not reserved", (long long)state_id); return DCREG_E_INVALID; }
    c->batch_state_valid[(size_t)state_id] = 0;
    return DCREG_OK;
}
int dcreg_linearize_batch_end(dcreg_ctx *c, int slot, dcreg_lin_out *outs) { return linearize_end(c, slot, outs); }
int dcreg_linearize_gated_begin(dcreg_ctx *c, int slot, const dcreg_lin_params *p) {
    return linearize_begin(c, slot, 1, nullptr, nullptr, nullptr, p, nullptr, true);
}
int dcreg_linearize_gate_open(dcreg_ctx *c, const double R[9], const double t[3]) {
    if (!c) return DCREG_E_INVALID;
    if (c->gate_slot < 0 || !R || !t) { c->fail("no gated linearisation waits for a pose"); return DCREG_E_STATE; }
    if (c->roi_active && !roi_covers(c, R, t, c->roi_pad)) {
        // the queued launch was built on a window this pose has left: it is called off; the caller starts the launch the plain way
        // (DCREG_E_STATE = "no launch waits", as after a timeout) and that builds the window around the new pose
        (void)dcreg_linearize_gate_abort(c);
        c->fail("the pose left the window of the map the queued launch was built on: launch called off");
        return DCREG_E_STATE;
    }
    gate_publish(c, c->gate_seq << 1, R, t);
    std::memcpy(c->last_R, R, sizeof(c->last_R)); std::memcpy(c->last_t, t, sizeof(c->last_t));
    c->last_pose_valid = true;
    c->gate_slot = -1;
    return DCREG_OK;
}
int dcreg_linearize_gate_abort(dcreg_ctx *c) {
    if (!c) return DCREG_E_INVALID;
    if (c->gate_slot < 0) return DCREG_OK;                                  // nothing queued
    gate_call_off(c);
    LinSlot &S = c->slots[c->gate_slot];
    S.pending = false;                 // no result will come; tickets_dirty stays set, so the next launch of the slot clears them
    free_tmp(S);
    // the kernels behind the gate return without touching anything: the state, its pose and the list counters are as before
    if (c->gate_uses_state) c->state_valid = c->gate_state_was_valid;
    c->gate_slot = -1;
    return DCREG_OK;
}
int dcreg_linearize_debug(dcreg_ctx *c, const double R[9], const double t[3], const dcreg_lin_params *p, dcreg_lin_out *out, dcreg_lin_debug *dbg) {
    return launch_linearize(c, 1, R, t, p, out, dbg);
}

int dcreg_knn(dcreg_ctx *c, const float *q, int64_t n, int64_t stride, int k, double max_radius, int32_t *idx, float *d2) {
    if (!c) return DCREG_E_INVALID;
    if (!q || !idx || !d2 || n < 0 || (k != 1 && k != 5)) { c->fail("invalid k-NN arguments (k must be 1 or 5)"); return DCREG_E_INVALID; }
    (void)roi_deactivate(c);
    if (c->n_tgt <= 0) { c->fail("target index is not set"); return DCREG_E_STATE; }
    if (n == 0) return DCREG_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = upload_cloud(c, q, n, stride, false, c->d_aligned, c->aligned_cap);
    if (rc) return rc;
    if (ensure(c, c->d_nn_idx, c->nn_idx_cap, (size_t)n * k) || ensure(c, c->d_nn_d2, c->nn_d2_cap, (size_t)n * k)) return DCREG_E_NOMEM;
    rc = launch_knn(c, c->grid, c->d_aligned, n, k, max_radius, nullptr, c->d_nn_idx, c->d_nn_d2);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(idx, c->d_nn_idx, sizeof(int32_t) * (size_t)n * k, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d2, c->d_nn_d2, sizeof(float) * (size_t)n * k, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DCREG_OK;
}

int dcreg_index_info_get(const dcreg_ctx *c, dcreg_index_info *info) {
    if (!c || !info) return DCREG_E_INVALID;
    const dcreg::GridDev &g = c->roi_active ? c->roi_store.grid : c->grid;         // the whole map's index, whichever is active
    info->cell = g.h;
    info->origin[0] = g.ox; info->origin[1] = g.oy; info->origin[2] = g.oz;
    info->dims[0] = g.nx; info->dims[1] = g.ny; info->dims[2] = g.nz;
    info->n_cells = c->roi_active ? c->roi_store.n_cells : c->n_cells; info->n_target = c->roi_active ? c->roi_store.n : c->n_tgt;
    info->n_source = c->n_src; info->max_ring = c->last_max_ring;
    return DCREG_OK;
}

int dcreg_roi_info(const dcreg_ctx *c, double info[11]) {
    if (!c || !info) return DCREG_E_INVALID;
    for (int a = 0; a < 3; ++a) { info[a] = c->roi_lo[a]; info[3 + a] = c->roi_hi[a]; }
    const bool have = c->roi_built && !c->roi_empty;
    info[6] = have ? (double)(c->roi_active ? c->n_tgt : c->roi_store.n) : 0.0;
    info[7] = have ? (c->roi_active ? c->grid.h : c->roi_store.grid.h) : 0.0;
    info[8] = (double)c->roi_rebuilds; info[9] = c->roi_active ? 1.0 : 0.0; info[10] = c->whole_capped ? 1.0 : 0.0;
    return DCREG_OK;
}

int dcreg_launch_stats_get(dcreg_ctx *c, dcreg_launch_stats *st, int reset) {
    if (!c || !st) return DCREG_E_INVALID;
    st->launches = c->n_launches; st->poses = c->n_poses_launched; st->points = c->n_points_launched;
    st->points_searched = -1; st->points_team = -1;
    if (c->opt_count_searches && c->d_search_count) {      // synchronous: every launch so far has finished when this returns
        std::vector<unsigned long long> v(kSearchCountBytes / sizeof(unsigned long long));
        HIP_TRY(c, hipMemcpyAsync(v.data(), c->d_search_count, kSearchCountBytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        unsigned long long tot = 0, team = 0;
        for (size_t k = 0; k < v.size(); k += kCounterStride / 2) { tot += v[k]; team += v[k + 1]; }
        st->points_searched = (int64_t)tot;
        st->points_team = (int64_t)team;
        if (reset) HIP_TRY(c, hipMemsetAsync(c->d_search_count, 0, kSearchCountBytes, c->stream));
    }
    if (reset) { c->n_launches = 0; c->n_poses_launched = 0; c->n_points_launched = 0; }
    return DCREG_OK;
}

int dcreg_launch_series(dcreg_ctx *c, double *ms, int64_t *searched, int64_t *refitted, int64_t *points, int64_t cap, int reset) {
    if (!c || cap < 0) return -1;
    const int64_t n = std::min<int64_t>(cap, (int64_t)c->launch_series.size());
    for (int64_t i = 0; i < n; ++i) {
        const dcreg_ctx::LaunchRec &r = c->launch_series[(size_t)i];
        if (ms) ms[i] = r.ms;
        if (searched) searched[i] = r.searched;
        if (refitted) refitted[i] = r.refitted;
        if (points) points[i] = r.points;
    }
    const int64_t total = (int64_t)c->launch_series.size();
    if (reset) c->launch_series.clear();
    return (int)std::min<int64_t>(total, 0x7FFFFFFF);
}

int dcreg_launch_series_passes(dcreg_ctx *c, uint8_t *advanced, int64_t cap) {
    if (!c || cap < 0) return -1;
    const int64_t n = std::min<int64_t>(cap, (int64_t)c->launch_series.size());
    for (int64_t i = 0; i < n; ++i) if (advanced) advanced[i] = (uint8_t)c->launch_series[(size_t)i].advanced;
    return (int)std::min<int64_t>((int64_t)c->launch_series.size(), 0x7FFFFFFF);
}

int dcreg_team_pass_stamps(dcreg_ctx *c, uint64_t *out, int64_t cap_blocks) {
    if (!c || cap_blocks < 0) return -1;
    if (!c->d_team_stamps || c->team_stamps_n == 0) return 0;
    const int64_t n = std::min<int64_t>(cap_blocks, (int64_t)c->team_stamps_n + 1);      // (+ 1: the outcome histogram behind the blocks)
    if (out && n > 0) {
        HIP_TRY(c, hipMemcpyAsync(out, c->d_team_stamps, sizeof(unsigned long long) * 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return (int)std::min<int64_t>((int64_t)c->team_stamps_n, 0x7FFFFFFF);
}

int dcreg_kernel_time(dcreg_ctx *c, double *ms_total, int64_t *launches, int reset) {
    if (!c) return DCREG_E_INVALID;
    if (ms_total) *ms_total = c->kernel_ms_total;
    if (launches) *launches = c->kernel_launches;
    if (reset) { c->kernel_ms_total = 0.0; c->kernel_launches = 0; }
    return DCREG_OK;
}

}  // extern "C"
