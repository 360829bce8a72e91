/*
 * dcreg_debug.h -- test, profiling and experiment hooks of libdcreg_hip.so.  NOT part of the drop-in boundary (dcreg.h is; see
 * INTEGRATION.md): nothing a maintainer of the reference binds lives here.  Used by tests/, bench.py and scripts/.
 */
#ifndef DCREG_DEBUG_H
#define DCREG_DEBUG_H

#include "dcreg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per-point dump for parity tests (original source order; any pointer may be NULL).
 * flag: 1 valid, 0 radius/knn gate, 2 |x|<min_normal_norm, 3 plane thickness, 4 weight<=weight_min.
 * nn_idx / nn_d2 are the reference's result list of nearestKSearch (icp_test_runner.cpp:1722) for the points that pass its radius
 * gate (:1726, flag != 0); for the others (flag 0) the reference never looks at the list and the dump holds -1 / +inf. */
typedef struct dcreg_lin_debug {
    int32_t *nn_idx; /* [5*n] original target indices, ascending (d2, idx); -1 = none */
    float *nn_d2;    /* [5*n] */
    uint8_t *flag;   /* [n] */
    double *normal;  /* [3*n] */
    double *r;       /* [n] */
    double *s;       /* [n] */
    uint32_t *stats; /* [n] search statistics: candidates evaluated | outermost shell << 16 */
    uint64_t *stamps; /* [8 * 4 * ceil(n / 256)] timing probe: per wave (query block x 4 + wave) shader-clock stamps at the phase boundaries
                         of the linearisation kernel + its searched / refitted lane counts.  With ONLY this pointer set the call is not a
                         dump: certificates are used as in a plain call and nothing else is copied back (scripts/wave_phases.py) */
} dcreg_lin_debug;

/* dcreg_linearize with the dump; always searches every point (k_full), shares the ctx's neighbour state with the plain calls */
int dcreg_linearize_debug(dcreg_ctx *, const double R[9], const double t[3], const dcreg_lin_params *,
                          dcreg_lin_out *, dcreg_lin_debug *);

/* total duration (ms, HIP events on the ctx stream around ALL kernels of a linearisation) of the timed linearisations since the
 * last reset, and their number; option "time_kernels" = N > 0 brackets every N-th linearisation of slot 0 (an event pair costs ~10 us
 * of host time), 0 = off */
int dcreg_kernel_time(dcreg_ctx *, double *ms_total, int64_t *launches, int reset);

/* what the linearisations since the last reset did.  points_searched needs the option "count_searches" = 1 (one atomic per searching
 * wave: off by default) and makes this call wait for the launches queued so far; -1 when the option is off. */
typedef struct dcreg_launch_stats {
    int64_t launches;          /* kernel launches (a batched launch counts once) */
    int64_t poses;             /* poses linearised */
    int64_t points;            /* source points those poses had, all told */
    int64_t points_searched;   /* ... of which went through the 6-NN search (the others' certificates held) */
    int64_t points_team;       /* ... of which were searched by a whole wave at a time (search.hpp team_search6: waves with a few
                                  lanes to search), the rest in lock-step; -1 like points_searched */
} dcreg_launch_stats;
int dcreg_launch_stats_get(dcreg_ctx *, dcreg_launch_stats *, int reset);

/* Log of the launches completed since the option "record_launches" = 1 was set (or since the last reset), oldest first: duration
 * (ms, HIP events; -1 for a launch that was not timed - see "time_kernels"), points that went through the 6-NN search (level 1),
 * points whose known neighbours were only re-ordered and refitted (level 2), points the launch linearised in all (poses x source
 * points).  The two counts travel in the count slots of the launch's own result rows (no atomics, no extra transfer); -1 for clouds of
 * more than 2^26 points.  Any array may be NULL; at most `cap` entries are written; returns the number of entries logged (which may
 * exceed cap), < 0 on invalid arguments.  bench.py's per-regime roofline is built from this. */
int dcreg_launch_series(dcreg_ctx *, double *ms, int64_t *searched, int64_t *refitted, int64_t *points, int64_t cap, int reset);
/* ... and, for the same log (call it BEFORE the resetting dcreg_launch_series): 1 where the advance pass (kernels.hpp k_advance: the
 * searches and refits of a launch in dense waves, in front of the linearisation kernel) ran, 2 where its small-frame form did
 * (k_advance_team: sixteen lanes per query), 0 where neither did; the duration and the counts of such a launch cover both kernels.
 * Bit 2 (+ 4): the linearisation kernel ran in one-wave blocks with k_sum_tiles behind it (option "one_wave"; its duration covers
 * both).  Returns the number of entries logged. */
int dcreg_launch_series_passes(dcreg_ctx *, uint8_t *advanced, int64_t cap);

/* Timing probe of the small-frame advance pass (option "team_stamps" = 1): of the LAST launch that ran the pass, per block (one wave,
 * kTeamTile points) eight shader-clock words - start, tests done, old neighbours gathered, rows listed, rows cut (table loads), candidates
 * taken, ranked, state written (first round of the block; 0 where a block had nothing to do); one more row of eight outcome counts
 * follows the blocks.  Copies at most cap_blocks x 8 words;
 * returns the number of blocks of that launch.  Waits for the stream. */
int dcreg_team_pass_stamps(dcreg_ctx *, uint64_t *out, int64_t cap_blocks);

/* The window index of a large map (options "roi_index" 0 never / 1 when "max_table_entries" enlarged the whole map's cell edge by more than one step or took its x sub-cells (default) /
 * 2 always, "roi_margin" metres, default 20): single-pose linearisations of such a map search an index over the map's points inside a box
 * around the transformed source - same neighbours, same sums (bitwise), cells sized for the local density instead of the map's extent;
 * everything else (dcreg_knn, dcreg_p2p_error, batches, dumps) runs on the whole map.  info[0..5] = the box (min xyz, max xyz),
 * info[6] = points in the window, info[7] = its cell edge, info[8] = windows built since the context was created, info[9] = 1 while the
 * window is the active index, info[10] = 1 when the whole map's build was cut by the table budget. */
int dcreg_roi_info(const dcreg_ctx *, double info[11]);

/* the analysis as the pipelined engine takes it: the part the step needs first, then what that left owed (*owed: 4 = the axis alignment of the Schur eigenvectors, 1 = the full
 * eigen-decomposition block, 2 = the diagonal blocks of the Schur analysis); the record must equal dcreg_analyze_degeneracy's */
int dcreg_analyze_degeneracy_two_part(const double H[36], int detection, int handling, const dcreg_config *, dcreg_analysis *, int *owed);

/* A kd-tree over the target cloud as a COMPARATOR of the grid index (SURVEY.md 7.1 "benchmark both"): median splits along the widest
 * axis, a complete implicit tree with at most leaf_size points per leaf, built on the host from the cloud of the last dcreg_set_target.
 * dcreg_knn_timed runs the exact k-NN (k = 1 or 5) of dcreg_knn on the grid (index 0: the ring walk of dcreg_knn; index 2: the row
 * sweep the linearisation uses, k = 5 with a radius only) or on the tree (index 1): one untimed launch, then `repeats` launches between
 * two HIP events; all return the same lists, bit for bit.  scripts/kdtree_compare.py. */
int dcreg_kdtree_build(dcreg_ctx *, int leaf_size);
int dcreg_kdtree_info(const dcreg_ctx *, int32_t *depth, int32_t *leaf_size, double *build_ms);
int dcreg_knn_timed(dcreg_ctx *, const float *q_xyz, int64_t n, int64_t stride_floats, int k, double max_radius, int index, int repeats,
                    int32_t *idx, float *d2, double *kernel_ms);

/* test and measurement knobs of dcreg_set_option (defaults are what the product runs with; none changes a result):
 *   "time_kernels"       see dcreg_kernel_time;
 *   "count_searches"     see dcreg_launch_stats;
 *   "record_launches"    see dcreg_launch_series;
 *   "use_certificates"   0 = search every point in every launch (the old neighbours still bound the searches), 1 = default;
 *   "keep_source_order"  1 = the next dcreg_set_source keeps the caller's point order instead of the Hilbert-curve sort;
 *   "advance"            the advance pass in front of single-pose launches: 0 = never, 1 (default) = when the last completed launch searched
 *                        between 1 % and 45 % of its points and the cloud has at least "advance_min_blocks" (2048) query blocks, 2 = whenever
 *                        the launch can take it (warm state, certificates in use): tests;
 *   "team_pass"          the small-frame advance pass (sixteen lanes per query) in front of single-pose launches: 0 = never, 1 (default) =
 *                        for clouds of at most 16384 points when the last completed launch searched at least half of them and the map holds
 *                        at least 3 points per occupied cell, 2 = whenever the launch can take it; "team_stamps": see dcreg_team_pass_stamps;
 *   "one_wave"           the linearisation kernel in one-wave blocks (a tile row per wave, k_sum_tiles behind it): 0 = never, 1 (default) =
 *                        single-pose launches of at least "one_wave_min_blocks" (1024) query blocks most of whose points are expected to
 *                        search ("one_wave_min_frac", 0.5) while the misalignment hint is above "one_wave_min_cells" (1.5) cells - the
 *                        first launches of a run; 2 = every fused launch, small ones included: tests; "one_wave_batches" (1): batched
 *                        launches of one-chunk poses (the Monte-Carlo batches: trials at every stage of their runs side by side) with
 *                        at least "one_wave_min_blocks" query blocks in all run that way too (+ 6 % on the experiment);
 *   "gate_in_kernel"     1 (default) = a pipelined launch of at most 64 query blocks waits for its pose in its first kernel (one kernel boundary
 *                        less); 0 = behind the one-wave gate kernel, like larger launches;
 *   "team_search"        lanes a wave serves one query at a time with all 64 lanes instead of searching in lock-step (0 = never, 7 = default);
 *   "curve_x_scale"      next dcreg_set_source: the cells of the source's Hilbert-curve order are 1 / v times as long in x as in y and z
 *                        (v <= 1; default 1 = cubes).
 * Settled and no longer options (rounds 3-5, profiles/r0?_ablation.md): query blocks are dealt to the XCDs in runs of 16;
 * launches of at most 64 query blocks publish their block rows straight to pinned memory; batched launches of one-chunk poses finish inside
 * the kernel; a start bound counts as loose 1.5 cells beyond the nearest occupied cell; the windows of the two advance passes as above. */

/* internal: the host-only translation units above the device seam (engine.cpp) store their error text where dcreg_last_error finds it */
void dcreg_set_error_message(dcreg_ctx *, const char *msg);

#ifdef __cplusplus
}
#endif
#endif /* DCREG_DEBUG_H */
