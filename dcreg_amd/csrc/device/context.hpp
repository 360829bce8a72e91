// dcreg_ctx: device state behind the C-ABI.  Stands for ICPContext (DCReg/include/utils.hpp:340-425):
// the kd-tree becomes a cell-sorted target + cell table in HBM, the per-point scratch vectors become
// nothing at all (the row of every point lives in registers and is reduced on the fly).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <vector>

#include "kernels.hpp"

struct dcreg_lin_params;
struct dcreg_lin_out;
struct dcreg_lin_debug;

// buffers and in-flight state of one linearisation slot

struct LinSlot {
    double *d_partials = nullptr; size_t partials_cap = 0;
    // batched poses: ONE pinned staging block [PoseArg x n | pose ids x n] and its device copy (one plain DMA per launch; a pageable
    // source is staged by the runtime, and beyond 16 KB that cost 14 us per launch)
    unsigned char *h_poses = nullptr, *d_poses = nullptr; size_t poses_cap = 0;      // bytes
    double *h_out = nullptr, *d_out = nullptr; size_t out_cap = 0;   // pinned, device-mapped result rows
    std::vector<double> h_rows;    // ... and the checked snapshot of them the sums are taken from
    unsigned int *d_tickets = nullptr; size_t tickets_cap = 0;
    bool tickets_dirty = false;    // a launch may have died half-way: clear the tickets before the next one
    std::vector<void *> tmp_dev;   // debug dump buffers of the launch in flight
    bool pending = false, fused = false, timed = false, sync = false;
    int n_poses = 0;
    uint32_t n_chunks = 0;
    bool direct = false;           // the rows are block rows of one chunk (kernels.hpp FinArgs::direct)
    std::vector<uint8_t> row_done; // wait_rows: rows taken so far
    std::vector<unsigned long long> row_chk;   // ... and the check word each row carried when it was last taken
    size_t n_rows = 0;
    unsigned long long seq = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // "time_kernels": the events of this slot's launch (the two slots alternate in a pipelined run)
    bool stamps_only = false;
    int advanced = 0;              // an advance pass ran in front of the launch in flight: 1 = k_advance, 2 = k_advance_team (kernels.hpp); + 4: k_lin ran in one-wave blocks
    bool coded = false;            // the launch in flight reports searched / refitted counts above its count slots (LinArgs::count_scale)
};

struct dcreg_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char err[512] = {0};

    // target
    int64_t n_tgt = 0;
    float4 *d_tgt_raw = nullptr; size_t tgt_raw_cap = 0;   // original order
    float4 *d_tgt = nullptr; size_t tgt_cap = 0;           // cell-sorted
    uint32_t *d_cell_start = nullptr; size_t cell_cap = 0;
    dcreg::GridDev grid{};
    int64_t n_cells = 0;
    uint32_t occupied_cells = 0;
    double radius_hint = 0.0;
    int last_max_ring = 0;

    // ---- the WINDOW index of a large map (context.hip roi_ensure).  A prior map whose dense cell table would exceed "max_table_entries" gets
    // coarser cells the larger its extent - a local search then pays for the size of the map.  Single-pose linearisations of such a map
    // therefore search a second index built over the points of a BOX around the transformed source (its bounding box at the pose, the search
    // radius, "roi_margin" metres on top): the same points as the whole map holds there, in cells sized for their density alone.  Every
    // neighbour of every query lies inside the box, so the searches return what they would on the whole map (exact: the sums are bitwise the
    // same, tests/test_gpu_round6.py); a pose that leaves the box rebuilds the window around itself.  One of the two indices is ACTIVE (the
    // members above: what every kernel launch uses), the other is kept in roi_store; swapping them drops the neighbour states (their
    // positions refer to one index's sorted order).  Everything but the single-pose product launches (k-NN, metrics, batches, dumps, the
    // kd-tree comparator) runs on the whole map.
    struct IndexSet {
        float4 *raw = nullptr; size_t raw_cap = 0; int64_t n = 0;
        float4 *sorted = nullptr; size_t sorted_cap = 0;
        uint32_t *cell_start = nullptr; size_t cell_cap = 0;
        dcreg::GridDev grid{}; int64_t n_cells = 0; uint32_t occupied = 0;
        uint8_t *gap = nullptr; size_t gap_cap = 0;
        uint32_t *owner = nullptr; size_t owner_cap = 0;
        uint32_t *ymask = nullptr; size_t ymask_cap = 0;
    };
    IndexSet roi_store;            // the index that is NOT active
    bool roi_active = false;       // the members above hold the window, roi_store the whole map
    bool roi_built = false;        // a window exists for the box roi_lo .. roi_hi
    bool roi_empty = false;        // ... but the map has no point in it: the whole map serves inside this box
    int opt_roi_index = 1;         // 0 never, 1 when the whole map's build ran into the table budget, 2 always
    double opt_roi_margin = 20.0;  // metres of the box beyond what the first pose needs
    bool whole_capped = false;     // the whole map's build enlarged its cells or dropped x sub-cells for the table budget
    bool last_build_capped = false;
    double roi_lo[3] = {}, roi_hi[3] = {}, roi_pad = 0.0;
    double src_mn[3] = {}, src_mx[3] = {};      // bounding box of the source in the body frame (dcreg_set_source)
    int64_t roi_rebuilds = 0;

    // auxiliary grid over the body-frame source (backward pass of dcreg_p2p_error)
    float4 *d_aux = nullptr; size_t aux_cap = 0;
    uint32_t *d_aux_cell_start = nullptr; size_t aux_cell_cap = 0;
    dcreg::GridDev aux_grid{};
    int64_t aux_n_cells = 0;
    bool aux_valid = false;

    // source
    int64_t n_src = 0;
    float4 *d_src_raw = nullptr; size_t src_raw_cap = 0;
    float4 *d_src = nullptr; size_t src_cap = 0;           // Hilbert-sorted
    // neighbour state of the ctx's own single-pose launches (search.hpp kStateRows): [kStateRows][state_stride]
    uint32_t *d_state = nullptr; size_t state_cap = 0;
    size_t state_stride = 0;
    bool state_valid = false;      // the state holds the results of a search of the current clouds
    // What the states hold was measured against the parameters of the launch that wrote it: certificates against the search / gate
    // radii, the stored gate bits against the plane thresholds, the stored plane by one of the two fits.  A launch with another key
    // finds the states empty (linearize_begin); the weights, the weight derivative and the parameterisation are not part of it
    // (they enter after the stored plane).  One key for the ctx's own state and one for the batch states.
    struct StateKey {
        double radius_sq = -1.0, max_thick_sq = 0.0, min_norm = 0.0;
        float radius_sq_f = 0.f, cert_r_out = 0.f, cert_r_in = 0.f;
        int fast_plane = -1;
        bool operator==(const StateKey &o) const {
            return radius_sq == o.radius_sq && max_thick_sq == o.max_thick_sq && min_norm == o.min_norm && radius_sq_f == o.radius_sq_f &&
                   cert_r_out == o.cert_r_out && cert_r_in == o.cert_r_in && fast_plane == o.fast_plane;
        }
    };
    StateKey state_key, batch_state_key;
    double src_radius = 0.0;       // largest distance of a source point from the body-frame origin (bounds a pose change's effect)
    // batched launches: n_batch_states states of the same layout, [state][kStateRows][state_batch_stride] (dcreg_reserve_warm_states),
    // and whether each holds anything yet
    uint32_t *d_state_batch = nullptr; size_t state_batch_cap = 0;
    size_t state_batch_stride = 0;
    int64_t n_batch_states = 0;
    std::vector<uint8_t> batch_state_valid;
    double *h_euler = nullptr, *d_euler = nullptr;     // Euler engine: the 27 derivative entries of a launch (LinArgs::dR)
    unsigned long long *d_search_count = nullptr;      // option "count_searches": points searched since the last reset

    // build scratch
    float *d_stage = nullptr; size_t stage_cap = 0;
    // small frames from host buffers (the registration path): the caller's floats are copied into this pinned block with a plain memcpy and
    // uploaded from there - the caller's buffer is consumed when dcreg_set_source returns whatever kind of memory it is, without a
    // stream synchronise; h_stage_ev = the upload behind the last use of the block
    float *h_stage = nullptr; size_t h_stage_cap = 0; hipEvent_t h_stage_ev = nullptr; bool h_stage_busy = false;
    uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
    size_t keys_cap = 0, keys2_cap = 0, vals_cap = 0, vals2_cap = 0;
    uint64_t *d_mkeys = nullptr, *d_mkeys2 = nullptr; size_t mkeys_cap = 0, mkeys2_cap = 0;
    uint32_t *d_scratch = nullptr;
    char *sort_tmp = nullptr; size_t sort_tmp_cap = 0;

    // linearisation: per-slot buffers (see linearize_begin / linearize_end)
    // gate of pipelined launches (kernels.hpp k_gate): pinned sequence number + pose, the device-resident pose it fills, abort word
    dcreg::GateHost *h_gate = nullptr, *d_gate_host = nullptr;
    dcreg::PoseArg *d_gate_pose = nullptr;
    uint32_t *d_gate_abort = nullptr;
    dcreg::GateDev *d_gate_dev = nullptr;  // device copy of the gate record: launches gated in their first kernel (kernels.hpp gate_wait)
    bool opt_gate_in_kernel = true;
    int opt_one_wave = 1;                 // k_lin<.., ONE> (one-wave blocks): 0 never, 1 by the rule, 2 wherever possible
    bool opt_one_wave_batches = true;     // ... for batched launches of one-chunk poses with at least opt_one_wave_min_blocks blocks in all
    double opt_one_wave_min_frac = 0.5;
    double opt_one_wave_min_cells = 1.5;
    int opt_one_wave_min_blocks = 1024;
    unsigned long long gate_seq = 0;       // number of the gated launch last queued
    int gate_slot = -1;                    // slot of the gated launch that still waits for its pose (-1: none)
    bool gate_uses_state = false;          // what the queued launch was built with: it reads / writes the ctx's own state,
    bool gate_state_was_valid = false;     //   and what state_valid was before it was queued (restored if it is called off)
    static constexpr int kLinSlots = 2;
    LinSlot slots[kLinSlots];

    // k-NN / p2p
    float4 *d_aligned = nullptr; size_t aligned_cap = 0;
    int32_t *d_nn_idx = nullptr; size_t nn_idx_cap = 0;
    float *d_nn_d2 = nullptr; size_t nn_d2_cap = 0;
    double *d_p2p_part = nullptr; size_t p2p_part_cap = 0;

    // native exchange of point-sharded runs (exchange.hip): an ncclComm_t on this ctx's device + staging rows
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    double *d_xrow = nullptr, *d_xall = nullptr, *h_xrow = nullptr, *h_xall = nullptr;

    // options / timing
    double opt_cell = 0.0, opt_cell_factor = 2.0;
    int opt_x_subdiv = 8;          // x sub-cells per grid cell (1, 2, 4, 8, 16)
    bool opt_use_cert = true;      // skip the search of every point whose certificate still holds (0: only bound the searches)
    bool opt_count_searches = false;
    double opt_cert_inflate = 0.005; // searches prune at (1 + inflate) x the 6th best distance: the 7th neighbour's lower bound (SET6 certificates)
    double opt_cert_margin = 0.05; // searches cover R (1 + margin): what "5th neighbour beyond R" certificates can spend
    int opt_time_kernels = 0;      // N > 0: bracket every N-th linearisation with HIP events
    uint64_t launch_counter = 0;
    double opt_wait_seconds = 30.0; // how long a result is awaited before the stream is drained to look for a device fault
    bool opt_spin = true;          // wait for results by spinning on pinned memory instead of hipStreamSynchronize
    bool need_set_device = true;
    unsigned long long seq = 0;
    int opt_xcd_chunk = 16;        // query-block -> XCD mapping (kernels.hpp xcd_remap): runs of 16 blocks round-robin (measured: C4 -13 %)
    bool opt_fast_plane = true;    // plane_fit_qr_fast (search.hpp) instead of the Eigen-shaped plane_fit_qr
    bool opt_gap_field = true;     // build the empty-space distance field of the target grid
    uint8_t *d_gap = nullptr; size_t gap_cap = 0;
    bool opt_direct_rows = true;   // single-pose launches of at most kChunk blocks publish block rows; the host adds them
    bool opt_far_bound = true;     // far queries with a loose bound start from the points around the nearest occupied cell (search.hpp lin_search6)
    uint32_t *d_owner = nullptr; size_t owner_cap = 0;
    uint32_t *d_ymask = nullptr; size_t ymask_cap = 0;
    bool opt_keep_source_order = false;   // experiments only
    // heavy groups first (kernels.hpp k_group_cost): the dispatch order of the query-block groups, estimated once per cloud pair
    bool opt_dispatch_order = true;
    uint8_t group_order[256] = {}; float *d_group_est = nullptr;
    uint32_t group_blocks = 0, n_groups = 0;
    int n_cus = 256;                // compute units of the device
    void *kd = nullptr;             // kd-tree comparator (kdtree.hip), built on request
    bool order_valid = false;       // group_order is the estimate for est_R / est_t; order_uneven: its costs differ enough to matter
    bool order_uneven = false;
    double est_R[9] = {}, est_t[3] = {};
    int64_t est_launch = 0;         // the launch number (seq) at which the estimate was made
    double hint_misalign = 1e300;   // dcreg_hint_misalignment
    // "nothing known" (a negative hint: what the engines say at the start of a run) is resolved at the next launch whose pose is known up
    // front: a pose within half a cell of the last linearised one continues that trajectory - the last hint still describes it (a run
    // that is stepped through in several engine calls does not pay a cost estimate at the start of each)
    bool hint_unknown = false;
    double hint_last = 1e300, last_R[9] = {}, last_t[3] = {};
    bool last_pose_valid = false;
    bool opt_fused_batches = true; // batched launches of one-chunk poses sum and publish per pose inside k_lin (kernels.hpp FinArgs::chunks_per_pose)
    double opt_curve_x_scale = 1.0;  // kernels.hpp k_curve_keys: < 1 stretches the patches of the source's curve order along x
    int64_t opt_max_table_entries = (int64_t)1 << 30;   // entries of the dense cell table (x sub-cells of the bounding box) before the cell edge grows
    double opt_far_loose = 1.5;    // search.hpp lin_search6: when a start bound is loose enough to be worth a probe (cells)
    // the advance pass (kernels.hpp k_advance): 0 never, 1 by the rule below, 2 whenever a launch can take it (tests)
    int opt_advance = 1;
    double opt_advance_lo = 0.01, opt_advance_hi = 0.45;     // ... the last completed launch searched between these fractions of its points
    int opt_advance_min_blocks = 2048;                        // ... and the cloud has at least this many query blocks (twice what the device holds)
    // its small-frame form (k_advance_team): same switch values; rule: at most max_points source points, the last completed launch
    // searched at least min_frac of them, the map has at least min_cell_pts points per occupied cell
    int opt_team_pass = 1;
    double opt_team_pass_max_points = 16384.0, opt_team_pass_min_frac = 0.5, opt_team_pass_min_cell_pts = 3.0;
    bool opt_team_stamps = false;
    unsigned long long *d_team_stamps = nullptr; size_t team_stamps_cap = 0; uint32_t team_stamps_n = 0;
    uint32_t *d_adv_counts = nullptr; size_t adv_counts_cap = 0; bool adv_counts_dirty = true;
    int64_t n_advance_launches = 0;
    int opt_team_max = 7;          // search.hpp team_search6: waves with at most this many lanes to search serve them cooperatively
    bool opt_warm = true;          // bound each search by the previous neighbour set (same exact result, fewer cells)
    int64_t n_launches = 0, n_poses_launched = 0, n_points_launched = 0;    // dcreg_launch_stats
    double kernel_ms_total = 0.0;
    int64_t kernel_launches = 0;
    // what the last completed launch did (decoded from the count slots of its result rows, search.hpp LinArgs::count_scale): points
    // searched / refitted, -1 = not reported.  Scheduling input of the next launches; "record_launches": every launch is also logged
    int64_t last_searched = -1, last_refitted = -1, last_points = 0;
    struct LaunchRec { double ms; int64_t searched, refitted, points; int advanced; };
    bool opt_record_launches = false;
    std::vector<LaunchRec> launch_series;

    void fail(const char *fmt, ...);
};

namespace dcreg {
int launch_linearize(dcreg_ctx *c, int n_poses, const double *R9, const double *t3, const dcreg_lin_params *p,
                     dcreg_lin_out *outs, dcreg_lin_debug *dbg_host);
void kdtree_free(void *kd);      // kdtree.hip (the comparator index of dcreg_debug.h)
int roi_deactivate(dcreg_ctx *c);      // context.hip: make the whole map's index the active one (entry points that are not single-pose linearisations)
int launch_knn(dcreg_ctx *c, const GridDev &grid, const float4 *d_q, int64_t n, int k, double max_radius, const PoseArg *pose,
               int32_t *d_idx, float *d_d2, bool sweep = false);
}  // namespace dcreg
