// TEST INFRASTRUCTURE ONLY: host replay of the device algorithm in dcreg_amd/csrc/device/search.hpp (see host_emul_shim.hpp).
// Builds the same grid index the device builds (cell choice restated from context.hip build_index), then runs the very same
// per-thread functions (lin_search / lin_row: warm bound, exact 5-NN with the ring walk, plane fit, gates, row) for every
// query on the CPU and sums the rows in fp64.  C-ABI for ctypes (tests/emul.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#define DCREG_HOST_EMUL 1
#include "host_emul_shim.hpp"
#include "../../dcreg_amd/csrc/device/search.hpp"

using namespace dcreg;

struct EmuIndex {
    std::vector<float4> pts;            // sorted by cell, +8 padding
    std::vector<uint32_t> cell_start;
    std::vector<uint8_t> gap;
    std::vector<uint32_t> owner;
    std::vector<uint32_t> ymask;
    bool sweep = true;
    bool team = false;                  // replay team_search6's algorithm for the queries it would take (tight warm bound)
    int64_t team_served = 0;
    GridDev g{};
    int64_t n_cells = 0;
    uint32_t occupied = 0;
};

static void grid_dims(double h, const double mn[3], const double mx[3], GridDev &g) {
    g.h = h; g.inv_h = 1.0 / h;
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
    g.nx = (int)std::floor((mx[0] - mn[0]) * g.inv_h) + 1;
    g.ny = (int)std::floor((mx[1] - mn[1]) * g.inv_h) + 1;
    g.nz = (int)std::floor((mx[2] - mn[2]) * g.inv_h) + 1;
}
static double cap_cell_for_budget(double h, const double mn[3], const double mx[3], double max_cells) {
    for (int it = 0; it < 64; ++it) {
        const double nx = std::floor((mx[0] - mn[0]) / h) + 1, ny = std::floor((mx[1] - mn[1]) / h) + 1, nz = std::floor((mx[2] - mn[2]) / h) + 1;
        if (nx * ny * nz <= max_cells && nx < 2e9 && ny < 2e9 && nz < 2e9) return h;
        h *= 1.26;
    }
    return h;
}
static inline int clampi_h(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static uint32_t build_at(EmuIndex &E, const float *xyz, int64_t n, double h, const double mn[3], const double mx[3], int sx = 1) {
    GridDev g{};
    grid_dims(h, mn, mx, g);
    g.sx = sx;
    const int nxf = g.nx * sx;
    const int64_t n_cells = (int64_t)nxf * g.ny * g.nz;       // table entries (x in sub-cells, as k_cell_keys)
    std::vector<uint32_t> keys((size_t)n), order((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int cx = clampi_h((int)std::floor(((double)xyz[3 * i] - g.ox) * g.inv_h * (double)sx), 0, nxf - 1);
        const int cy = clampi_h((int)std::floor(((double)xyz[3 * i + 1] - g.oy) * g.inv_h), 0, g.ny - 1);
        const int cz = clampi_h((int)std::floor(((double)xyz[3 * i + 2] - g.oz) * g.inv_h), 0, g.nz - 1);
        keys[(size_t)i] = (uint32_t)(((int64_t)cz * g.ny + cy) * nxf + cx);
    }
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });   // radix sort is stable too
    E.pts.assign((size_t)n + dcreg::kPtsPad, float4{0.f, 0.f, 0.f, 0.f});
    E.cell_start.assign((size_t)n_cells + 1, 0u);
    uint32_t occ = 0;
    std::vector<uint32_t> cnt((size_t)n_cells, 0u);
    for (int64_t i = 0; i < n; ++i) ++cnt[keys[(size_t)i]];
    uint32_t run = 0;
    for (int64_t c = 0; c < n_cells; ++c) { E.cell_start[(size_t)c] = run; run += cnt[(size_t)c]; occ += cnt[(size_t)c] ? 1u : 0u; }
    E.cell_start[(size_t)n_cells] = run;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t o = order[(size_t)i];
        E.pts[(size_t)i] = float4{xyz[3 * o], xyz[3 * o + 1], xyz[3 * o + 2], __uint_as_float(o)};
    }
    g.n_pts = (uint32_t)n;
    E.g = g; E.n_cells = n_cells / sx; E.occupied = occ;
    return occ;
}

// Scalar replay of search.hpp team_search6 for ONE query (the device spreads this over the 64 lanes of a wave): the nine rows of the
// 27-cell block cut by team_row, every point of them with a float distance below the bound, ranked by (distance bits, original
// index); the first six + the seventh's distance -> positions and certificate.  false: more than 64 points inside the bound (the
// device leaves such a query to the lock-step search).
static bool team_search_host(const GridDev &g, const LinArgs &a, float qx, float qy, float qz, float bound, uint32_t (&pos_out)[6], uint32_t &cert_out) {
    struct Ent { uint64_t key; uint32_t pos, d2; };
    std::vector<Ent> list;
    for (int r = 0; r < 9; ++r) {
        uint32_t s_ = 0, e_ = 0;
        team_row(g, qx, qy, qz, bound, r % 3 - 1, r / 3 - 1, s_, e_);
        for (uint32_t p = s_; p < e_; ++p) {
            const float4 c = g.pts[p];
            const float d2 = dist2_nofma(qx, qy, qz, c);
            if (d2 < bound) list.push_back(Ent{((uint64_t)__float_as_uint(d2) << 32) | __float_as_uint(c.w), p, __float_as_uint(d2)});
        }
    }
    if (list.size() > 64) return false;
    std::sort(list.begin(), list.end(), [](const Ent &x, const Ent &y) { return x.key < y.key; });
    Set6 out{};
    const uint32_t n = (uint32_t)list.size();
    for (int j = 0; j < 6; ++j) {
        const bool got = (uint32_t)j < n;
        out.pos[j] = got ? list[(size_t)j].pos : kNoIdx;
        out.d2[j] = got ? __uint_as_float(list[(size_t)j].d2) : bound;
    }
    out.lb7 = n > 6u ? fminf(__uint_as_float(list[6].d2), bound) : bound;
    cert_out = make_cert(out, a);
    for (int j = 0; j < 6; ++j) pos_out[j] = out.pos[j];
    return true;
}

extern "C" {

void *emu_index_build(const float *xyz, int64_t n, double radius_hint, double opt_cell, double cell_factor, int gap_field, int x_subdiv) {
    EmuIndex *E = new EmuIndex();
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], (double)xyz[3 * i + a]); mx[a] = std::max(mx[a], (double)xyz[3 * i + a]); }
    const double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-6});
    const double max_cells = (double)((int64_t)1 << 27);
    const double h_cap = radius_hint > 0.0 ? radius_hint * 1.00001 : ext / std::cbrt((double)n) * 4.0;
    double h = opt_cell > 0.0 ? opt_cell : h_cap;
    h = cap_cell_for_budget(h, mn, mx, max_cells);
    uint32_t occ = build_at(*E, xyz, n, h, mn, mx);
    if (opt_cell <= 0.0) {     // density-adaptive cell, as context.hip build_index
        const double target_occ = 1.59 * cell_factor * cell_factor;
        double m1 = (double)n / std::max<uint32_t>(occ, 1);
        double h1 = h, expo = 2.0;
        for (int pass = 0; pass < 2 && m1 > target_occ * 1.3; ++pass) {
            double h2 = h1 * std::pow(target_occ / m1, 1.0 / expo);
            h2 = std::max(h2, h_cap / 64.0);
            h2 = cap_cell_for_budget(h2, mn, mx, max_cells);
            if (h2 >= h1 * 0.95) break;
            occ = build_at(*E, xyz, n, h2, mn, mx);
            const double m2 = (double)n / std::max<uint32_t>(occ, 1);
            if (m2 < m1 && h2 < h1) expo = std::min(3.0, std::max(1.0, std::log(m1 / m2) / std::log(h1 / h2)));
            h1 = h2; m1 = m2;
        }
    }
    {   // x sub-cells once the cell edge is settled, as context.hip build_index
        int sx = 1;
        while (sx < 16 && sx * 2 <= x_subdiv) sx *= 2;
        while (sx > 1 && (double)E->g.nx * sx * E->g.ny * E->g.nz > max_cells) sx >>= 1;
        if (sx > 1) build_at(*E, xyz, n, E->g.h, mn, mx, sx);
    }
    GridDev &g = E->g;
    g.cell_start = E->cell_start.data();
    g.pts = E->pts.data();
    g.gap = nullptr; g.gap_cap = 0; g.owner = nullptr;
    if (gap_field) {           // empty-space field, as build_gap_field / k_gap_dilate
        int rings = 1;
        while (rings < 12 && (double)rings * g.h < (radius_hint > 0.0 ? radius_hint : 4.0 * g.h)) ++rings;
        if (rings >= 2) {
            const int nx = g.nx, ny = g.ny, nz = g.nz;
            // one field: seeds -> `rings` dilation passes (as k_gap_dilate: a cell takes the nearest of the owners of its neighbours of the ring before)
            auto dilate = [&](std::vector<uint8_t> &gap, std::vector<uint32_t> &owner) {
                // (only the cells next to the ring before can join a ring: they are collected from that ring's cells instead of scanning the grid)
                std::vector<int64_t> front, cand;
                for (int64_t c = 0; c < E->n_cells; ++c) if (gap[(size_t)c] == 0) front.push_back(c);
                std::vector<uint8_t> mark((size_t)E->n_cells, 0);
                for (int r = 1; r <= rings && !front.empty(); ++r) {
                    cand.clear();
                    for (int64_t f : front) {
                        const int x = (int)(f % nx), y = (int)((f / nx) % ny), z = (int)(f / ((int64_t)nx * ny));
                        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            const int xx = x + dx, yy = y + dy, zz = z + dz;
                            if (xx < 0 || yy < 0 || zz < 0 || xx >= nx || yy >= ny || zz >= nz) continue;
                            const int64_t nb = ((int64_t)zz * ny + yy) * nx + xx;
                            if (gap[(size_t)nb] == 255 && !mark[(size_t)nb]) { mark[(size_t)nb] = 1; cand.push_back(nb); }
                        }
                    }
                    for (int64_t c : cand) {
                        const int x = (int)(c % nx), y = (int)((c / nx) % ny), z = (int)(c / ((int64_t)nx * ny));
                        int64_t best = INT64_MAX;
                        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                            const int xx = x + dx, yy = y + dy, zz = z + dz;
                            if (xx < 0 || yy < 0 || zz < 0 || xx >= nx || yy >= ny || zz >= nz) continue;
                            const size_t nb = (size_t)(((int64_t)zz * ny + yy) * nx + xx);
                            if (gap[nb] != (uint8_t)(r - 1)) continue;
                            const uint32_t o = owner[nb];
                            const int64_t ox = o % (uint32_t)nx, oy = (o / (uint32_t)nx) % (uint32_t)ny, oz = o / ((uint32_t)nx * (uint32_t)ny);
                            const int64_t d = (ox - x) * (ox - x) + (oy - y) * (oy - y) + (oz - z) * (oz - z);
                            if (d < best) { best = d; owner[(size_t)c] = o; }
                        }
                    }
                    for (int64_t c : cand) gap[(size_t)c] = (uint8_t)r;       // (after all of the ring's owners are chosen: a ring only reads the ring before)
                    front.swap(cand);
                }
            };
            E->gap.assign((size_t)E->n_cells, 255);
            E->owner.assign((size_t)E->n_cells, kNoIdx);
            for (int64_t c = 0; c < E->n_cells; ++c) if (E->cell_start[(size_t)(c + 1) * g.sx] > E->cell_start[(size_t)c * g.sx]) { E->gap[(size_t)c] = 0; E->owner[(size_t)c] = (uint32_t)c; }
            dilate(E->gap, E->owner);
            {   // the probe's owners: the nearest DENSE cell where there is one within the rings (kernels.hpp k_gap_init_dense / k_owner_merge)
                std::vector<uint8_t> gap2((size_t)E->n_cells, 255);
                std::vector<uint32_t> own2((size_t)E->n_cells, kNoIdx);
                for (int64_t c = 0; c < E->n_cells; ++c) {
                    const int x = (int)(c % nx);
                    const int64_t row = c - x;
                    const bool own = E->cell_start[(size_t)(c + 1) * g.sx] > E->cell_start[(size_t)c * g.sx];
                    const uint32_t run = E->cell_start[(size_t)(row + std::min(x + 2, nx)) * g.sx] - E->cell_start[(size_t)(row + std::max(x - 1, 0)) * g.sx];
                    if (own && run >= 6u) { gap2[(size_t)c] = 0; own2[(size_t)c] = (uint32_t)c; }
                }
                dilate(gap2, own2);
                for (int64_t c = 0; c < E->n_cells; ++c) if (own2[(size_t)c] != kNoIdx) E->owner[(size_t)c] = own2[(size_t)c];
            }
            g.gap = E->gap.data(); g.gap_cap = rings; g.owner = E->owner.data();
        }
    }
    {   // row occupancy words, as k_ymask / build_row_words
        const int nx = g.nx, ny = g.ny, nz = g.nz;
        g.nxb = (nx + 15) >> 4; g.nyw = (ny + 31) >> 5;
        E->ymask.assign((size_t)nz * g.nxb * g.nyw, 0u);
        const int64_t nxf = (int64_t)nx * g.sx;
        for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int xb = 0; xb < g.nxb; ++xb) {
            const int64_t row = ((int64_t)z * ny + y) * nxf;
            const int xa = xb * 16, xe = std::min(xa + 16, nx);
            if (E->cell_start[(size_t)(row + (int64_t)xe * g.sx)] > E->cell_start[(size_t)(row + (int64_t)xa * g.sx)])
                E->ymask[((size_t)z * g.nxb + xb) * g.nyw + (y >> 5)] |= 1u << (y & 31);
        }
        g.ymask = E->ymask.data();

    }
    return E;
}
void emu_index_free(void *p) { delete (EmuIndex *)p; }
// the searches of emu_linearize: 1 = row sweep (as k_lin), 0 = ring walk (as -DDCREG_RING_WALK; the two must agree bit for bit)
void emu_index_set_sweep(void *p, int32_t on) { ((EmuIndex *)p)->sweep = on != 0; }
void emu_index_set_team(void *p, int32_t on) { ((EmuIndex *)p)->team = on != 0; }
int64_t emu_index_team_served(void *p) { return ((EmuIndex *)p)->team_served; }
void emu_index_info(void *p, double *h, int32_t dims[3], int64_t *n_cells, int32_t *gap_cap) {
    EmuIndex *E = (EmuIndex *)p;
    *h = E->g.h; dims[0] = E->g.nx; dims[1] = E->g.ny; dims[2] = E->g.nz; *n_cells = E->n_cells; *gap_cap = E->g.gap_cap;
}

// Hilbert-curve order of a cloud, as k_curve_keys + radix sort on the device
static uint64_t spread21_h(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
void emu_hilbert_order(const float *xyz, int64_t n, uint32_t *order) {
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], (double)xyz[3 * i + a]); mx[a] = std::max(mx[a], (double)xyz[3 * i + a]); }
    const double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2], 1e-6});
    const double inv_q = 2097151.0 / ext * 0.999999;
    std::vector<uint64_t> keys((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t X[3];
        for (int a = 0; a < 3; ++a) X[a] = (uint32_t)std::fmin(std::fmax(((double)xyz[3 * i + a] - mn[a]) * inv_q, 0.0), 2097151.0);
        const uint32_t M = 1u << 20;
        for (uint32_t Q = M; Q > 1; Q >>= 1) {
            const uint32_t P = Q - 1;
            for (int a = 0; a < 3; ++a) {
                if (X[a] & Q) X[0] ^= P;
                else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
            }
        }
        X[1] ^= X[0]; X[2] ^= X[1];
        uint32_t t = 0;
        for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
        X[0] ^= t; X[1] ^= t; X[2] ^= t;
        keys[(size_t)i] = (spread21_h(X[0]) << 2) | (spread21_h(X[1]) << 1) | spread21_h(X[2]);
    }
    std::iota(order, order + n, 0u);
    std::stable_sort(order, order + n, [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
}

struct EmuLinParams {
    double search_radius, max_plane_thickness_sq, min_normal_norm, weight_slope, weight_min;
    int32_t use_weight_derivative, fast_plane_fit;
    double cert_margin, cert_inflate;
};

// One linearisation of `n` queries (src_xyz in the order given; order[i] = original index written into the per-point
// outputs, or null for identity), replaying what the kernels of kernels.hpp do per query.
//   state   : [kStateRows][stride] neighbour state, read and updated (null = keep nothing); fresh != 0: it holds nothing yet
//   certify : 0 = k_full (search every query, bounded by the old neighbours when the state has some);
//             1 = k_rows / k_search_list / k_rows<LISTED>: test every certificate at the query's new position, search only the
//                 queries whose certificate does not hold there
// out32: 21 H + 6 g + sum r^2 + sum b^2 + n_eff + n_pt.  Per-point outputs may be null.  stats: [n][8] counters
// {candidates, outermost shell, table loads, rows, runs, trips, faces, face skips}; counts[0] = queries searched, [1] = queries refitted.
// nn_idx of a query that passed the radius gate on its stored plane (no list rebuilt in this launch) is -2.
int emu_linearize(void *idx, const float *src_xyz, const uint32_t *order, int64_t n, const double R[9], const double t[3],
                  const EmuLinParams *p, uint32_t *state, int64_t stride, int fresh, int certify, int warm,
                  double *out32, int32_t *nn_idx, float *nn_d2,
                  uint8_t *flag_out, double *normal, double *r_out, double *s_out, uint32_t *stats, uint32_t *trace, int64_t trace_cap_per_query,
                  int64_t *counts) {
    EmuIndex *E = (EmuIndex *)idx;
    const GridDev &g = E->g;
    LinArgs a{};
    a.radius_sq = p->search_radius * p->search_radius;
    {   // as context.hip make_lin_args
        const double rs = p->search_radius * (1.0 + p->cert_margin), r2 = rs * rs;
        float rf = (float)r2;
        if ((double)rf < r2) rf = std::nextafterf(rf, INFINITY);
        a.radius_sq_f = std::nextafterf(rf, INFINITY);
        float ro = (float)(p->search_radius * (1.0 + 1e-5));
        if ((double)ro < p->search_radius * (1.0 + 1e-5)) ro = std::nextafterf(ro, INFINITY);
        a.cert_r_out = ro;
        float ri = (float)(p->search_radius * (1.0 - 1e-5));
        if ((double)ri > p->search_radius * (1.0 - 1e-5)) ri = std::nextafterf(ri, 0.0f);
        a.cert_r_in = ri;
    }
    a.max_thick_sq = p->max_plane_thickness_sq; a.min_norm = p->min_normal_norm; a.w_slope = p->weight_slope; a.w_min = p->weight_min;
    a.use_wd = p->use_weight_derivative;
    int k = 1;
    while (k < 100000) { const double safe = (double)k * g.h * (1.0 - 1e-9); if (safe * safe * (1.0 - 1e-6) >= (double)a.radius_sq_f) break; ++k; }
    a.max_ring = k;
    a.warm = warm;
    a.far_loose = 1.5f;
    a.prune_infl = (float)((1.0 + p->cert_inflate) * (1.0 + p->cert_inflate));
    a.infl_max_d2 = (float)(4.0 * g.h * g.h);
    a.state = state; a.state_stride = (uint32_t)stride; a.euler = 0; a.dR = nullptr;
    PoseArg P{};
    std::memcpy(P.R, R, sizeof(P.R)); std::memcpy(P.t, t, sizeof(P.t));
    P.state = state ? 0u : kNoIdx; P.fresh = fresh ? 1u : 0u;
    if (certify && (!state || fresh)) return -1;
    static thread_local RunList runs;
    double tot[31];
    for (double &v : tot) v = 0.0;
    threadIdx.x = 0;
    int64_t n_searched = 0, n_fitted = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t oi = order ? order[i] : (uint32_t)i;
        const float4 s4{src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2], __uint_as_float(oi)};
        emu_stats = EmuStats{};
        if (trace) { emu_trace.buf = trace + (size_t)i * (size_t)trace_cap_per_query; emu_trace.cap = (uint32_t)trace_cap_per_query - 1; emu_trace.n = 0; }
        float qx, qy, qz;
        body_to_global(P, (double)s4.x, (double)s4.y, (double)s4.z, qx, qy, qz);
        uint32_t cert = kCertSearch, fitw = kFitNone, pos6[6];
        for (int j = 0; j < 6; ++j) pos6[j] = kNoIdx;
        const bool old = state && !fresh;
        uint32_t *st = state;                    // word `row` of this query: ST(row) (search.hpp state_word_index)
        const size_t ss = (size_t)stride;
#define ST(row) (st[state_word_index((row), (size_t)i, ss)])
        float q0x = 0.f, q0y = 0.f, q0z = 0.f;
        if (certify) {                           // what the fast path reads
            cert = ST(6); fitw = ST(10);
            q0x = __uint_as_float(ST(7)); q0y = __uint_as_float(ST(8)); q0z = __uint_as_float(ST(9));
        }
        const bool need = !(certify && cert_holds(cert, q0x, q0y, q0z, qx, qy, qz));                 // level 1: search
        const bool refit = !need && !cert_is_out(cert) && !fit_holds(fitw, q0x, q0y, q0z, qx, qy, qz);   // level 2: gather, order, fit
        Set6 s6{};
        if (need) {
            if (old && warm) for (int j = 0; j < 6; ++j) pos6[j] = ST(j);
            uint32_t c2;
            bool by_team = false;
            if (E->team && old && warm && pos6[5] != kNoIdx) {      // the queries k_lin hands to team_search6 in a sparse wave
                bool tight = false;
                const float tb = team_bound(g, a, pos6, qx, qy, qz, tight);
                uint32_t tpos[6];
                if (tight && team_search_host(g, a, qx, qy, qz, tb, tpos, c2)) {
                    by_team = true;
                    for (int j = 0; j < 6; ++j) s6.pos[j] = tpos[j];
                    s6.n_eval = 0; s6.n_shell = 1;
                    ++E->team_served;
                }
            }
            if (!by_team) {
                if (E->sweep) lin_search6<true>(g, runs, a, true, warm && old, pos6, qx, qy, qz, s6, c2);
                else lin_search6<false>(g, runs, a, true, warm && old, pos6, qx, qy, qz, s6, c2);
            }
            cert = c2;
            for (int j = 0; j < 6; ++j) pos6[j] = s6.pos[j];
            if (state) for (int j = 0; j < 6; ++j) ST(j) = s6.pos[j];
            ++n_searched;
        }
        if (trace) { emu_trace.buf[trace_cap_per_query - 1] = emu_trace.n; emu_trace.buf = nullptr; }
        double row[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[31], nrm[3] = {0, 0, 0}, rr = 0.0, ss_ = 0.0;
        uint8_t fl = 0, gate = 255;
        KnnResult<5> nn{};
        Fit fit{};
        const bool set = !cert_is_out(cert);
        const bool fitnow = set && (need || refit);
        if (fitnow) {
            const bool six = cert_is_set6(cert);      // (then the fit certificate has to cover the 5th / 6th gap itself)
            if (!need) for (int j = 0; j < 6; ++j) pos6[j] = ST(j);
            if (!six) pos6[5] = kNoIdx;
            const bool presorted = need;                   // (k_lin: uniform over the wave; a wave of one lane here)
            const uint8_t in_r = p->fast_plane_fit ? fit_from_set<true>(g, a, qx, qy, qz, pos6, six, nn, fit, presorted) : fit_from_set<false>(g, a, qx, qy, qz, pos6, six, nn, fit, presorted);
            gate = in_r ? (uint8_t)(fit.word & 3u) : (uint8_t)255;
            if (state) {
                if (!need) cert = cert_rebased(cert, q0x, q0y, q0z, qx, qy, qz);
                ST(6) = cert; ST(7) = __float_as_uint(qx); ST(8) = __float_as_uint(qy); ST(9) = __float_as_uint(qz);
                ST(10) = fit.word;
                for (int k = 0; k < 4; ++k) { uint64_t b; std::memcpy(&b, &fit.plane[k], 8); ST(11 + 2 * k) = (uint32_t)b; ST(12 + 2 * k) = (uint32_t)(b >> 32); }
            }
            ++n_fitted;
        } else if (set) {                        // level 3: the stored plane
            gate = (uint8_t)(fitw & 3u);
            for (int k = 0; k < 4; ++k) { const uint64_t b = ((uint64_t)ST(12 + 2 * k) << 32) | ST(11 + 2 * k); std::memcpy(&fit.plane[k], &b, 8); }
        } else if (need && state) {
            ST(6) = cert; ST(7) = __float_as_uint(qx); ST(8) = __float_as_uint(qy); ST(9) = __float_as_uint(qz);
            ST(10) = kFitNone;
        }
        if (gate == 0) fl = p->fast_plane_fit ? row_of_plane<true>(P, a, s4, qx, qy, qz, fit.plane, row, nrm, rr, ss_) : row_of_plane<false>(P, a, s4, qx, qy, qz, fit.plane, row, nrm, rr, ss_);
        else fl = gate == 255 ? 0 : gate;
        const bool listed = fitnow;              // the neighbour list exists only where it was rebuilt in this launch
        row_products(row, fl, acc);
        for (int j = 0; j < 31; ++j) tot[j] += acc[j];
        if (nn_idx) for (int j = 0; j < 5; ++j) nn_idx[5 * (size_t)oi + j] = (fl != 0 && listed) ? (int32_t)nn.idx[j] : (fl != 0 ? -2 : -1);
        if (nn_d2) for (int j = 0; j < 5; ++j) nn_d2[5 * (size_t)oi + j] = (fl != 0 && listed) ? nn.d2[j] : INFINITY;
        if (flag_out) flag_out[oi] = fl;
        if (fl == 1 || fl == 4) {
            if (normal) { normal[3 * (size_t)oi] = nrm[0]; normal[3 * (size_t)oi + 1] = nrm[1]; normal[3 * (size_t)oi + 2] = nrm[2]; }
            if (r_out) r_out[oi] = rr;
            if (s_out) s_out[oi] = ss_;
        }
        if (stats) {
            uint32_t *s = stats + 8 * (size_t)i;      // in processing order (the wave model groups consecutive queries)
            s[0] = s6.n_eval; s[1] = need ? s6.n_shell : 0; s[2] = emu_stats.table_loads; s[3] = emu_stats.rows; s[4] = emu_stats.runs;
            s[5] = emu_stats.trips; s[6] = emu_stats.faces; s[7] = emu_stats.face_skips;
        }
    }
#undef ST
    for (int j = 0; j < 31; ++j) out32[j] = tot[j];
    out32[31] = 0.0;
    if (counts) { counts[0] = n_searched; counts[1] = n_fitted; }
    return 0;
}

// plain exact k-NN (k = 1 or 5) of host queries, as k_knn / dcreg_knn: max_radius <= 0 -> unbounded
int emu_knn(void *idx, const float *q_xyz, int64_t n, int k, double max_radius, int32_t *out_idx, float *out_d2) {
    EmuIndex *E = (EmuIndex *)idx;
    const GridDev &g = E->g;
    float bound = 3.0e38f;
    int max_ring = -1;
    if (max_radius > 0.0 && std::isfinite(max_radius)) {
        const double r2 = max_radius * max_radius;
        float rf = (float)r2; if ((double)rf < r2) rf = std::nextafterf(rf, INFINITY);
        bound = std::nextafterf(rf, INFINITY);
        int kk = 1;
        while (kk < 100000) { const double s = (double)kk * g.h * (1.0 - 1e-9); if (s * s * (1.0 - 1e-6) >= (double)bound) break; ++kk; }
        max_ring = kk;
    }
    static thread_local RunList runs;
    threadIdx.x = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float qx = q_xyz[3 * i], qy = q_xyz[3 * i + 1], qz = q_xyz[3 * i + 2];
        if (k == 1) {
            KnnResult<1> nn;
            knn_exact<1>(g, runs, qx, qy, qz, bound, max_ring, nn);
            out_idx[i] = nn.idx[0] != kNoIdx ? (int32_t)nn.idx[0] : -1;
            out_d2[i] = nn.idx[0] != kNoIdx ? nn.d2[0] : INFINITY;
        } else {
            KnnResult<5> nn;
            knn_exact<5>(g, runs, qx, qy, qz, bound, max_ring, nn);
            for (int j = 0; j < 5; ++j) {
                out_idx[5 * i + j] = nn.idx[j] != kNoIdx ? (int32_t)nn.idx[j] : -1;
                out_d2[5 * i + j] = nn.idx[j] != kNoIdx ? nn.d2[j] : INFINITY;
            }
        }
    }
    return 0;
}

// the nine cell-table intervals [s, e) of each query's 27-cell block for a given bound (team_row = knn_search's phase A): design data
// for wave-level candidate tiles (scripts/tile_union_model.py)
void emu_block_rows(void *idx, const float *q_xyz, const float *bounds, int64_t n, uint32_t *out) {
    EmuIndex *E = (EmuIndex *)idx;
    for (int64_t i = 0; i < n; ++i)
        for (int r = 0; r < 9; ++r)
            team_row(E->g, q_xyz[3 * i], q_xyz[3 * i + 1], q_xyz[3 * i + 2], bounds[i], r % 3 - 1, r / 3 - 1, out[(i * 9 + r) * 2], out[(i * 9 + r) * 2 + 1]);
}

// Scalar replay of the row enumeration of kernels.hpp k_advance_team for ONE query and bound: the rows of the ball (every row of its
// bounding square when that is at most sixteen rows, else the rows whose occupancy bit is set in the words of the ball's z layers - one
// lane per layer on the device, two words x two 16-cell x blocks each), every row cut to the ball by ball_row; all points of those runs
// with a float distance below the bound, ranked by (distance bits, original index).  out_idx / out_d2: the first seven (-1 / inf where
// fewer); returns the number of points inside the bound, or -1 - k where the device would leave the query to k_lin (k = 1: more layers
// than lanes, 2: a layer wider than one lane reads, 3: more than 64 occupied rows).
int64_t emu_ball_query(void *idx, const float *q, float bound, int32_t *out_idx, float *out_d2) {
    EmuIndex *E = (EmuIndex *)idx;
    const GridDev &g = E->g;
    const float qx = q[0], qy = q[1], qz = q[2];
    const BallCells bc = ball_cells(g, qx, qy, qz, bound);
    const int ny_r = bc.yhi - bc.ylo + 1, nz_r = bc.zhi - bc.zlo + 1;
    std::vector<std::pair<int, int>> rows;
    if (ny_r * nz_r <= 16) {
        for (int k = 0; k < ny_r * nz_r; ++k) rows.push_back({bc.ylo + k % ny_r, bc.zlo + k / ny_r});
    } else {
        if (nz_r > 16 || !g.ymask) return -2;
        for (int gl = 0; gl < nz_r; ++gl) {
            const int dz = bc.zlo + gl, z = bc.cz + dz;
            if (z < 0 || z >= g.nz) continue;
            const float hf = (float)g.h;
            const float gz = dz < 0 ? ((float)(-dz - 1) + bc.frz) * hf : (dz > 0 ? ((float)dz - bc.frz) * hf : 0.f);
            const float rem = bound - gz * gz * 0.99999f;
            if (rem < 0.f) continue;
            const float rc = fminf(sqrt_approx(rem) * 1.00001f * (float)g.inv_h + 1e-4f, 1.0e6f);
            const int cap = 1 << 24;
            const int ylo = -std::min(cap, (int)floorf(rc + 1.f - bc.fry)), yhi = std::min(cap, (int)floorf(rc + bc.fry));
            const int xlo = -std::min(cap, (int)floorf(rc + 1.f - bc.frx)), xhi = std::min(cap, (int)floorf(rc + bc.frx));
            const int y0 = std::max(bc.cy + ylo, 0), y1 = std::min(bc.cy + yhi, g.ny - 1);
            const int b0 = std::max(bc.cx + xlo, 0) >> 4, b1 = std::min(bc.cx + xhi, g.nx - 1) >> 4;
            if (y1 < y0 || b1 < b0) continue;
            if (((y1 >> 5) - (y0 >> 5)) > 1 || b1 - b0 > 1) return -3;
            const int yw0 = y0 >> 5, yw1 = y1 >> 5, bb = std::min(b0 + 1, b1);
            const uint32_t *mw = g.ymask + ((int64_t)z * g.nxb + b0) * g.nyw, *mv = g.ymask + ((int64_t)z * g.nxb + bb) * g.nyw;
            uint32_t m0 = mw[yw0] | mv[yw0], m1 = yw1 > yw0 ? (mw[yw1] | mv[yw1]) : 0u;
            const int base0 = yw0 << 5;
            { const int lo = std::max(y0 - base0, 0), hi = std::min(y1 - base0, 31); m0 &= (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo); }
            if (yw1 > yw0) { const int hi = std::min(y1 - (base0 + 32), 31); m1 &= (0xFFFFFFFFu >> (31 - hi)); }
            for (int b = 0; b < 32; ++b) if (m0 & (1u << b)) rows.push_back({base0 + b - bc.cy, dz});
            for (int b = 0; b < 32; ++b) if (m1 & (1u << b)) rows.push_back({base0 + 32 + b - bc.cy, dz});
        }
        if (rows.size() > 64) return -4;
    }
    struct Ent { uint64_t key; float d2; uint32_t idx; };
    std::vector<Ent> list;
    for (const auto &r : rows) {
        uint32_t s_ = 0, e_ = 0;
        ball_row(g, bc, bound, r.first, r.second, s_, e_);
        for (uint32_t p = s_; p < e_; ++p) {
            const float4 c = g.pts[p];
            const float d2 = dist2_nofma(qx, qy, qz, c);
            if (d2 < bound) list.push_back(Ent{((uint64_t)__float_as_uint(d2) << 32) | __float_as_uint(c.w), d2, __float_as_uint(c.w)});
        }
    }
    std::sort(list.begin(), list.end(), [](const Ent &x, const Ent &y) { return x.key < y.key; });
    for (int j = 0; j < 7; ++j) {
        const bool got = (size_t)j < list.size();
        out_idx[j] = got ? (int32_t)list[(size_t)j].idx : -1;
        out_d2[j] = got ? list[(size_t)j].d2 : INFINITY;
    }
    return (int64_t)list.size();
}

// plane fit alone: Q = 5 neighbours (row-major 5x3); fast = 1 -> plane_fit_qr_fast
void emu_plane_fit(const double *Q, int fast, double x[3]) {
    double qx[5], qy[5], qz[5];
    for (int j = 0; j < 5; ++j) { qx[j] = Q[3 * j]; qy[j] = Q[3 * j + 1]; qz[j] = Q[3 * j + 2]; }
    double y[3];
    if (fast) plane_fit_qr_fast(qx, qy, qz, y); else plane_fit_qr(qx, qy, qz, y);
    x[0] = y[0]; x[1] = y[1]; x[2] = y[2];
}

}  // extern "C"
