// Engine seam: the SO(3) point-to-plane ICP loop of TestRunner::Point2PlaneICP_SO3_OpenMP
// (DCReg/src/icp_test_runner.cpp:1611-2060) with steps 1-5 of every iteration replaced by ONE call into
// the device seam (dcreg_linearize), and the num_runs loop of TestRunner::runMethod (:331-390) as a
// lock-step batch over independent trials (dcreg_icp_run_trials).
#include <omp.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../../include/dcreg_debug.h"
#include "linalg.hpp"
#include "se3.hpp"

namespace dcreg {
bool invertSpd6(const double H[36], double inv[36]);
int analyzeStep(const double H[36], int detection, int handling, const dcreg_config &cfg, dcreg_analysis &res);    // solver.cpp
void analyzeFinish(const double H[36], dcreg_analysis &res, int owed);
}

namespace {

// The CPUs this process can keep busy: min(affinity mask, cgroup quota).  A container shows the MACHINE's hardware threads to OpenMP
// (omp_get_max_threads() = 256 on a box whose cgroup grants 16): a team of that size on a sixteenth of the CPUs made the Monte-Carlo
// experiment seven times slower for a caller that had not called dcreg_set_host_threads (measured).  The engines never use more.
inline int usable_cpus() {
    static const int n = [] {
        int cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (int)std::max(1L, sysconf(_SC_NPROCESSORS_ONLN));
        double quota = 0.0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
            char q[32] = {0}; double per = 0.0;
            if (std::fscanf(f, "%31s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0.0) quota = std::atof(q) / per;
            std::fclose(f);
        } else {
            double q = -1.0, per = 0.0;                                                   // cgroup v1
            if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lf", &q) != 1) q = -1.0; std::fclose(fq); }
            if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lf", &per) != 1) per = 0.0; std::fclose(fp); }
            if (q > 0.0 && per > 0.0) quota = q / per;
        }
        if (quota >= 1.0) cpus = std::min(cpus, (int)quota);
        return std::max(1, cpus);
    }();
    return n;
}
inline int host_team() { return std::max(1, std::min(omp_get_max_threads(), usable_cpus())); }

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

inline dcreg_lin_params lin_params_of(const dcreg_config &cfg) {
    dcreg_lin_params p;
    dcreg_default_lin_params(&p, cfg.search_radius);
    p.use_weight_derivative = cfg.use_weight_derivative;
    return p;
}

// steps 6-9 of one iteration for one state; returns 0 continue, 1 converged, 2 abort (non-finite)
// defer = true: when nothing of the step depends on the full eigen-decomposition of H (a diagnostic of the log for "Ours"), it is
// left to host_step_finish() - the caller lets the device start on the new pose in between
struct StepOut { dcreg_analysis an; double dx[6]; double H[36]; int evd_owed = 0; };
inline int host_step(const dcreg_lin_out &lo, int detection, int handling, const dcreg_config &cfg, double R[9], double t[3], StepOut &so,
                     bool defer = false) {
    dcreg_unpack_hessian(lo.H_upper, so.H);
    if (defer) so.evd_owed = dcreg::analyzeStep(so.H, detection, handling, cfg, so.an);
    else dcreg_analyze_degeneracy(so.H, detection, handling, &cfg, &so.an);           // :1922-1923
    dcreg_solve_degenerate_system(so.H, lo.g, handling, &cfg, &so.an, so.dx);          // :1940
    for (double v : so.dx) if (!std::isfinite(v)) return 2;                            // :1942-1950
    dcreg::boxplus(R, t, so.dx, R, t);                                                 // :1953
    const double dr = std::sqrt(so.dx[0] * so.dx[0] + so.dx[1] * so.dx[1] + so.dx[2] * so.dx[2]);
    const double dt = std::sqrt(so.dx[3] * so.dx[3] + so.dx[4] * so.dx[4] + so.dx[5] * so.dx[5]);
    return (dr < cfg.CONVERGENCE_THRESH_ROT && dt < cfg.CONVERGENCE_THRESH_TRANS) ? 1 : 0;   // :1998
}

inline void host_step_finish(StepOut &so) {
    if (so.evd_owed) { dcreg::analyzeFinish(so.H, so.an, so.evd_owed); so.evd_owed = 0; }
}

// eigenvalue clamp of a symmetric 6x6 (:2020-2029): only when the smallest eigenvalue is <= 1e-12 (or `always`), to 1e-9
inline void clamp_psd6(dcreg::Mat6 &M, bool always) {
    dcreg::Mat6 Ms;                                                             // SelfAdjointEigenSolver reads one triangle
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ms.v[i * 6 + j] = M.v[(i > j ? i : j) * 6 + (i > j ? j : i)];   // lower triangle
    dcreg::Vec<6> w; dcreg::Mat6 V;
    const bool ok = dcreg::symEig<6>(Ms, w, V);
    double mn = w[0];
    for (double x : w) mn = std::min(mn, x);
    if (!always && ok && mn > 1e-12) return;
    for (double &x : w) x = std::max(x, 1e-9);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
        double s2 = 0.0;
        for (int k = 0; k < 6; ++k) s2 += V.v[i * 6 + k] * w[k] * V.v[j * 6 + k];
        M.v[i * 6 + j] = s2;
    }
}

inline void covariance_of(bool converged, const double Hlast[36], double cov[36]) {   // :2014-2037
    for (int i = 0; i < 36; ++i) cov[i] = (i % 7 == 0) ? 1e6 : 0.0;
    if (!converged) return;
    double inv[36];
    if (!dcreg::invertSpd6(Hlast, inv)) return;
    dcreg::Mat6 C;
    std::memcpy(C.v, inv, sizeof(inv));
    clamp_psd6(C, false);
    std::memcpy(cov, C.v, sizeof(C.v));
}

}  // namespace

extern "C" {

int dcreg_icp_run_sharded(dcreg_ctx *ctx, const double R0[9], const double t0[3], int detection, int handling,
                          const dcreg_config *cfg, int64_t n_source_total, dcreg_reduce_fn reduce, void *reduce_user,
                          dcreg_iter_log *log, int log_capacity, dcreg_icp_result *res) {
    if (!ctx || !R0 || !t0 || !cfg || !res) return DCREG_E_INVALID;
    std::memset(res, 0, sizeof(*res));
    // every rank passes the same total, so a bad value makes all of them return here together (no collective is pending)
    if (reduce && n_source_total <= 0) return DCREG_E_INVALID;
    const auto t_total = Clock::now();
    double R[9], t[3], Hlast[36];
    std::memcpy(R, R0, sizeof(R)); std::memcpy(t, t0, sizeof(t));
    for (int i = 0; i < 36; ++i) Hlast[i] = (i % 7 == 0) ? 1.0 : 0.0;
    const dcreg_lin_params prm = lin_params_of(*cfg);
    dcreg_index_info info;
    dcreg_index_info_get(ctx, &info);
    const double n_src_all = reduce ? (double)n_source_total : (double)info.n_source;
    // Sharded runs: a rank whose slice is empty, or whose target is missing, still takes part in EVERY exchange (it
    // contributes a zero row, or a poisoned one) - leaving the loop alone would block the other ranks in the collective.
    const bool local_empty = info.n_source <= 0;
    const bool local_bad = info.n_target <= 0;
    if (!reduce && (local_empty || local_bad)) {   // :1635-1646
        res->status = 3;
        std::memcpy(res->R, R, sizeof(R)); std::memcpy(res->t, t, sizeof(t));
        covariance_of(false, Hlast, res->icp_cov);
        return DCREG_OK;
    }
    int rc_all = DCREG_OK;
    // One pair, no exchange: the launches are pipelined - while linearisation `it` runs, linearisation `it + 1` is queued behind
    // a gate (dcreg_linearize_gated_begin); the host step then only publishes the pose.  slot = where linearisation `it` lives.
    const bool piped = !reduce;
    int slot = 0;
    bool queued = false;                        // a gated launch waits for its pose
    (void)dcreg_hint_misalignment(ctx, -1.0);   // nothing known about the start pose (scheduling only: include/dcreg.h)
    for (int it = 0; it < cfg->max_iterations; ++it) {
        const auto t_iter = Clock::now();
        dcreg_lin_out lo;
        std::memset(&lo, 0, sizeof(lo));
        int rc = DCREG_OK;
        if (piped) {
            if (it == 0) rc = dcreg_linearize_batch_begin(ctx, slot, 1, R, t, &prm);
            if (rc != DCREG_OK) return rc;
            queued = it + 1 < cfg->max_iterations && dcreg_linearize_gated_begin(ctx, slot ^ 1, &prm) == DCREG_OK;
            rc = dcreg_linearize_batch_end(ctx, slot, &lo);
            if (rc != DCREG_OK) { if (queued) dcreg_linearize_gate_abort(ctx); return rc; }
        } else if (!local_bad && !local_empty) {
            rc = dcreg_linearize(ctx, R, t, &prm, &lo);
        }
        if (!reduce && rc != DCREG_OK) return rc;
        if (reduce) {   // point sharding: this rank linearised its slice; the sums of all slices, added in rank order
            double row[32];
            std::memcpy(row, lo.H_upper, 21 * sizeof(double)); std::memcpy(row + 21, lo.g, 6 * sizeof(double));
            row[27] = lo.sum_r2; row[28] = lo.sum_b2; row[29] = (double)lo.n_eff; row[30] = (double)lo.n_pt;
            row[31] = (rc != DCREG_OK || local_bad) ? 1.0 : 0.0;          // poison flag: summed like the rest
            if (rc != DCREG_OK || local_bad) for (int k = 0; k < 31; ++k) row[k] = 0.0;
            if (reduce(row, reduce_user) != 0) return DCREG_E_DEVICE;   // the exchange itself failed: nothing left to wait for
            if (row[31] != 0.0) {                                        // some rank failed: every rank stops here, together
                rc_all = rc != DCREG_OK ? rc : (local_bad ? DCREG_E_STATE : DCREG_E_DEVICE);
                break;
            }
            std::memcpy(lo.H_upper, row, 21 * sizeof(double)); std::memcpy(lo.g, row + 21, 6 * sizeof(double));
            lo.sum_r2 = row[27]; lo.sum_b2 = row[28]; lo.n_eff = (int64_t)std::llround(row[29]); lo.n_pt = (int64_t)std::llround(row[30]);
        }
        if (lo.n_eff < 10) {                            // :1847-1854
            res->iterations = it + 1; res->converged = 0; res->status = 1;
            break;
        }
        (void)dcreg_hint_misalignment(ctx, std::sqrt(lo.sum_r2 / (double)lo.n_eff));     // for the launches queued from here on
        StepOut so;
        const int st = host_step(lo, detection, handling, *cfg, R, t, so, piped);
        if (st == 2) { res->iterations = it; res->converged = 0; res->status = 2; break; }
        if (piped && st != 1 && it + 1 < cfg->max_iterations) {
            // the pose of the next linearisation exists: let the device go before the bookkeeping below
            if (queued) rc = dcreg_linearize_gate_open(ctx, R, t);                       // the queued launch starts now
            // (no launch waits: queueing ahead had failed, or the wait for this iteration's result ran out of patience and called the
            // queued launch off before draining the stream - context.hip wait_rows)
            if (!queued || rc == DCREG_E_STATE) rc = dcreg_linearize_batch_begin(ctx, slot ^ 1, 1, R, t, &prm);
            queued = false;
            if (rc != DCREG_OK) return rc;
            slot ^= 1;
        }
        host_step_finish(so);                                                 // the part of the analysis only the log reads
        std::memcpy(Hlast, so.H, sizeof(Hlast));
        if (log && it < log_capacity) {
            dcreg_iter_log &L = log[it];
            std::memset(&L, 0, sizeof(L));
            L.iter_count = it;
            L.effective_points = lo.n_eff; L.corr_pt_count = lo.n_pt;
            L.fitness = (double)lo.n_pt / n_src_all;                      // :1856
            L.rmse = std::sqrt(lo.sum_r2 / (double)lo.n_eff);             // :1858
            L.objective_value = 0.5 * lo.sum_b2;                          // :1919
            for (int i = 0; i < 6; ++i) { L.gradient[i] = -lo.g[i]; L.update_dx[i] = so.dx[i]; }   // :1918
            dcreg::stateToMatrix(R, t, L.transform_matrix);               // :1954
            dcreg::poseError(cfg->gt_matrix, L.transform_matrix, &L.trans_error_vs_gt, &L.rot_error_vs_gt);   // :1976
            std::memcpy(L.H_upper, lo.H_upper, sizeof(L.H_upper));
            L.analysis = so.an;
            L.iter_time_ms = ms_since(t_iter);                            // :1973
        }
        res->iterations = it + 1;
        if (st == 1) { res->converged = 1; break; }
    }
    if (queued) dcreg_linearize_gate_abort(ctx);     // left the loop early: the launch queued ahead is called off
    std::memcpy(res->R, R, sizeof(R)); std::memcpy(res->t, t, sizeof(t));
    covariance_of(res->converged != 0, Hlast, res->icp_cov);
    res->time_ms = ms_since(t_total);
    return rc_all;
}

static int rccl_reduce(double row[32], void *user) { return dcreg_comm_allgather_sum((dcreg_ctx *)user, row) == DCREG_OK ? 0 : 1; }

int dcreg_icp_run_sharded_rccl(dcreg_ctx *ctx, const double R0[9], const double t0[3], int detection, int handling,
                               const dcreg_config *cfg, int64_t n_source_total, dcreg_iter_log *log, int log_capacity,
                               dcreg_icp_result *res) {
    return dcreg_icp_run_sharded(ctx, R0, t0, detection, handling, cfg, n_source_total, rccl_reduce, ctx, log, log_capacity, res);
}

int dcreg_icp_run(dcreg_ctx *ctx, const double R0[9], const double t0[3], int detection, int handling,
                  const dcreg_config *cfg, dcreg_iter_log *log, int log_capacity, dcreg_icp_result *res) {
    return dcreg_icp_run_sharded(ctx, R0, t0, detection, handling, cfg, 0, nullptr, nullptr, log, log_capacity, res);
}

int dcreg_icp_run_many(int n, dcreg_ctx *const *ctxs, const double *R0, const double *t0, int detection, int handling,
                       const dcreg_config *cfg, dcreg_icp_result *results) {
    if (n < 0 || (n > 0 && (!ctxs || !R0 || !t0 || !cfg || !results))) return DCREG_E_INVALID;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return DCREG_E_INVALID;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (ctxs[i] == ctxs[j]) return DCREG_E_INVALID;   // a ctx is single-threaded
    std::vector<int> rc((size_t)std::max(n, 0), DCREG_OK);
    std::vector<std::thread> th;
    th.reserve((size_t)std::max(n - 1, 0));
    auto body = [&](int i) { rc[(size_t)i] = dcreg_icp_run(ctxs[i], R0 + 9 * (size_t)i, t0 + 3 * (size_t)i, detection, handling, cfg, nullptr, 0, &results[i]); };
    for (int i = 1; i < n; ++i) th.emplace_back(body, i);
    if (n > 0) body(0);
    for (auto &t_ : th) t_.join();
    for (int i = 0; i < n; ++i) if (rc[(size_t)i] != DCREG_OK) return rc[(size_t)i];
    return DCREG_OK;
}

// The num_runs loop of TestRunner::runMethod (:331-390) over independent trials of one cloud pair, as a continuously refilled batch:
// `slots` trials are in flight at a time, split into two groups that alternate on the device - while the kernel of one group runs
// the host takes steps 6-9 of the other (the two linearisation slots of the ctx), so the 6x6 solves cost no wall time.  Every
// group iteration is ONE batched launch over its live trials.  A trial that ends (converged, aborted, out of iterations) hands
// its slot - and the slot's neighbour state, marked empty - to the next trial in line at once, so the batch stays full until the
// queue runs dry instead of thinning out while its stragglers finish.  Each trial is bitwise the single run of its initial pose.
static int run_trials_core(dcreg_ctx *ctx, int64_t n_trials, const double *R0, const double *t0, int detection, int handling,
                           const dcreg_config *cfg, dcreg_trial_result *results, int slots_wanted) {
    const auto t_total = Clock::now();
    const dcreg_lin_params prm = lin_params_of(*cfg);
    dcreg_index_info info;
    dcreg_index_info_get(ctx, &info);
    for (int64_t i = 0; i < n_trials; ++i) std::memset(&results[i], 0, sizeof(results[i]));
    if (info.n_source <= 0 || info.n_target <= 0) {
        for (int64_t i = 0; i < n_trials; ++i) results[i].status = 3;
        return DCREG_OK;
    }
    if (cfg->max_iterations <= 0) {
        for (int64_t i = 0; i < n_trials; ++i) {
            dcreg::stateToMatrix(R0 + 9 * i, t0 + 3 * i, results[i].final_transform);
            dcreg::poseError(cfg->gt_matrix, results[i].final_transform, &results[i].trans_error_m, &results[i].rot_error_deg);
        }
        return DCREG_OK;
    }
    int n_slots = (int)std::min<int64_t>(n_trials, slots_wanted > 0 ? slots_wanted : 256);
    n_slots = std::min(n_slots, 2 * 65535);
    const int n_groups = n_slots >= 64 ? 2 : 1;
    // one neighbour state per slot; without the memory for them the trials still run, every launch searching from scratch
    bool have_states = dcreg_reserve_warm_states(ctx, n_slots) == DCREG_OK;
    struct Slot { int64_t trial = -1; int it = 0; double R[9], t[3]; };
    std::vector<Slot> slot((size_t)n_slots);
    struct Group { std::vector<int> live; std::vector<int32_t> ids; std::vector<double> Rb, tb; std::vector<dcreg_lin_out> outs; bool in_flight = false; };
    Group grp[2];
    int64_t next_trial = 0;
    auto load = [&](int si) -> bool {                      // next trial in line -> slot si
        if (next_trial >= n_trials) { slot[(size_t)si].trial = -1; return false; }
        Slot &S = slot[(size_t)si];
        S.trial = next_trial++; S.it = 0;
        std::memcpy(S.R, R0 + 9 * S.trial, sizeof(S.R)); std::memcpy(S.t, t0 + 3 * S.trial, sizeof(S.t));
        if (have_states) dcreg_reset_warm_state(ctx, si);
        return true;
    };
    for (int si = 0; si < n_slots; ++si) load(si);
    double t_lin_ms = 0.0, t_host_ms = 0.0;
    int64_t n_steps = 0;
    auto begin = [&](int gi) -> int {
        Group &G = grp[gi];
        G.live.clear();
        for (int si = gi; si < n_slots; si += n_groups) if (slot[(size_t)si].trial >= 0) G.live.push_back(si);
        if (G.live.empty()) return DCREG_OK;
        const int nl = (int)G.live.size();
        G.Rb.resize((size_t)nl * 9); G.tb.resize((size_t)nl * 3); G.outs.resize((size_t)nl); G.ids.resize((size_t)nl);
        for (int j = 0; j < nl; ++j) {
            const Slot &S = slot[(size_t)G.live[(size_t)j]];
            G.ids[(size_t)j] = have_states ? (int32_t)G.live[(size_t)j] : -1;
            std::memcpy(&G.Rb[(size_t)j * 9], S.R, sizeof(S.R)); std::memcpy(&G.tb[(size_t)j * 3], S.t, sizeof(S.t));
        }
        const int rc = dcreg_linearize_batch_begin_warm(ctx, gi, nl, G.Rb.data(), G.tb.data(), G.ids.data(), &prm);
        G.in_flight = rc == DCREG_OK;
        return rc;
    };
    auto finish = [&](int gi) -> int {          // wait for the group's results, take the host steps, refill the slots that ended
        Group &G = grp[gi];
        if (!G.in_flight) return DCREG_OK;
        const auto t_a = Clock::now();
        const int rc = dcreg_linearize_batch_end(ctx, gi, G.outs.data());
        G.in_flight = false;
        if (rc != DCREG_OK) return rc;
        t_lin_ms += ms_since(t_a);
        const auto t_b = Clock::now();
        const int nl = (int)G.live.size();
        // host steps 6-9 of every live trial are independent: spread them over host threads
        std::vector<uint8_t> ended((size_t)nl, 0);
        const int nthreads = std::max(1, std::min({host_team(), 32, nl / 8}));
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int j = 0; j < nl; ++j) {
            Slot &S = slot[(size_t)G.live[(size_t)j]];
            dcreg_trial_result &tr = results[S.trial];
            const dcreg_lin_out &lo = G.outs[(size_t)j];
            const int it = S.it;
            ended[(size_t)j] = 1;
            if (lo.n_eff < 10) { tr.iterations = it + 1; tr.status = 1; continue; }          // :1847-1854
            StepOut so;
            // (defer = true and the owed part never paid: a trial record holds nothing of the full eigen-decomposition of H or the
            // diagonal blocks - diagnostics of the per-iteration log, which a trial does not keep; mask, update and pose are the same)
            const int st = host_step(lo, detection, handling, *cfg, S.R, S.t, so, true);
            if (st == 2) { tr.iterations = it; tr.status = 2; continue; }
            tr.iterations = it + 1;
            tr.final_rmse = std::sqrt(lo.sum_r2 / (double)lo.n_eff);
            tr.final_fitness = (double)lo.n_pt / (double)info.n_source;
            tr.corr_num = lo.n_eff;
            std::memcpy(tr.H_upper, lo.H_upper, sizeof(tr.H_upper));
            std::memcpy(tr.degenerate_mask, so.an.degenerate_mask, sizeof(tr.degenerate_mask));
            if (st == 1) { tr.converged = 1; continue; }
            S.it = it + 1;
            if (S.it < cfg->max_iterations) ended[(size_t)j] = 0;
        }
        for (int j = 0; j < nl; ++j) {
            if (!ended[(size_t)j]) continue;
            const int si = G.live[(size_t)j];
            Slot &S = slot[(size_t)si];
            dcreg_trial_result &tr = results[S.trial];
            dcreg::stateToMatrix(S.R, S.t, tr.final_transform);
            dcreg::poseError(cfg->gt_matrix, tr.final_transform, &tr.trans_error_m, &tr.rot_error_deg);   // :501-503
            load(si);
        }
        t_host_ms += ms_since(t_b);
        ++n_steps;
        return DCREG_OK;
    };
    auto has_work = [&](int gi) {
        if (grp[gi].in_flight) return true;
        for (int si = gi; si < n_slots; si += n_groups) if (slot[(size_t)si].trial >= 0) return true;
        return false;
    };
    int rc = begin(0);
    while (rc == DCREG_OK && (has_work(0) || (n_groups == 2 && has_work(1)))) {
        if (n_groups == 2 && !grp[1].in_flight && (rc = begin(1)) != DCREG_OK) break;   // queued behind group 0
        if ((rc = finish(0)) != DCREG_OK) break;                                         // host steps of group 0 overlap group 1's kernel
        if (!grp[0].in_flight && (rc = begin(0)) != DCREG_OK) break;                     // queued behind group 1
        if (n_groups == 2 && (rc = finish(1)) != DCREG_OK) break;                        // host steps of group 1 overlap group 0's kernel
    }
    if (rc != DCREG_OK) {                                            // drain whatever is still queued
        for (int gi = 0; gi < 2; ++gi) if (grp[gi].in_flight) { grp[gi].outs.resize(grp[gi].live.size()); (void)dcreg_linearize_batch_end(ctx, gi, grp[gi].outs.data()); }
        return rc;
    }
    const double total_ms = ms_since(t_total);
    if (std::getenv("DCREG_TRIALS_TIMING"))
        std::fprintf(stderr, "[dcreg_icp_run_trials] %lld trials in %d slots, %lld group steps: wait for results %.1f us/step, host %.1f us/step, wall %.1f us/step\n",
                     (long long)n_trials, n_slots, (long long)n_steps, 1e3 * t_lin_ms / std::max<int64_t>(n_steps, 1),
                     1e3 * t_host_ms / std::max<int64_t>(n_steps, 1), 1e3 * total_ms / std::max<int64_t>(n_steps, 1));
    for (int64_t i = 0; i < n_trials; ++i) results[i].time_ms = total_ms / (double)n_trials;   // amortised: trials advance together
    return DCREG_OK;
}

int dcreg_icp_run_trials(dcreg_ctx *ctx, int n_trials, const double *R0, const double *t0, int detection, int handling,
                         const dcreg_config *cfg, dcreg_trial_result *results) {
    if (!ctx || !R0 || !t0 || !cfg || !results || n_trials < 0) return DCREG_E_INVALID;
    if (n_trials == 0) return DCREG_OK;
    return run_trials_core(ctx, n_trials, R0, t0, detection, handling, cfg, results, 0);
}

int dcreg_icp_run_montecarlo(dcreg_ctx *ctx, const double base_xyzrpy[6], uint64_t seed, int64_t first_trial, int64_t trial_stride,
                             int64_t n_trials, double trans_amp, double rot_amp_rad, int detection, int handling,
                             const dcreg_config *cfg, int slots, dcreg_trial_result *results) {
    if (!ctx || !base_xyzrpy || !cfg || !results || n_trials < 0 || first_trial < 0 || trial_stride < 1) return DCREG_E_INVALID;
    if (n_trials == 0) return DCREG_OK;
    std::vector<double> R0((size_t)n_trials * 9), t0((size_t)n_trials * 3);
#pragma omp parallel for schedule(static) num_threads(host_team())
    for (int64_t j = 0; j < n_trials; ++j) {
        double T[16];
        dcreg_trial_pose(base_xyzrpy, seed, first_trial + j * trial_stride, trans_amp, rot_amp_rad, T, nullptr);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R0[(size_t)j * 9 + r * 3 + c] = T[r * 4 + c]; t0[(size_t)j * 3 + r] = T[r * 4 + 3]; }
    }
    return run_trials_core(ctx, n_trials, R0.data(), t0.data(), detection, handling, cfg, results, slots);
}

// The experiment as one job over the ranks of the ctx's communicator (include/dcreg.h): shard, run, ONE gather, statistics on every rank.
int dcreg_montecarlo_job(dcreg_ctx *ctx, const double base_xyzrpy[6], uint64_t seed, int64_t n_trials, double trans_amp, double rot_amp_rad,
                         int detection, int handling, const dcreg_config *cfg, int slots, double *records, dcreg_method_stats *stats) {
    if (!ctx || !base_xyzrpy || !cfg || n_trials < 0) return DCREG_E_INVALID;
    int rank = 0, world = 1;
    (void)dcreg_comm_info(ctx, &rank, &world);
    constexpr int REC = DCREG_TRIAL_RECORD_DOUBLES;
    const int64_t mine = n_trials > rank ? (n_trials - rank + world - 1) / world : 0;      // trials rank, rank + world, ...
    const int64_t per = (n_trials + world - 1) / world;                                    // fixed-size blocks: padded with trial index -1
    std::vector<dcreg_trial_result> res((size_t)std::max<int64_t>(mine, 1));
    int rc = dcreg_icp_run_montecarlo(ctx, base_xyzrpy, seed, rank, world, mine, trans_amp, rot_amp_rad, detection, handling, cfg, slots, res.data());
    // (a rank whose share failed still takes part in the collective - with an empty block - so that the others do not hang; it reports its error)
    std::vector<double> blk((size_t)per * REC, 0.0);
    for (int64_t j = 0; j < per; ++j) blk[(size_t)j * REC + 9] = -1.0;
    if (rc == DCREG_OK) {
        for (int64_t j = 0; j < mine; ++j) {
            const dcreg_trial_result &t = res[(size_t)j];
            double *r = &blk[(size_t)j * REC];
            r[0] = t.converged; r[1] = t.iterations; r[2] = t.time_ms; r[3] = t.trans_error_m; r[4] = t.rot_error_deg; r[5] = t.final_rmse;
            r[6] = t.final_fitness; r[7] = (double)t.corr_num; r[8] = t.status; r[9] = (double)(rank + j * world);
            for (int k = 0; k < 16; ++k) r[10 + k] = t.final_transform[k];
            for (int k = 0; k < 21; ++k) r[26 + k] = t.H_upper[k];
            for (int k = 0; k < 6; ++k) r[47 + k] = t.degenerate_mask[k];
        }
    }
    std::vector<double> all((size_t)per * REC * (size_t)world);
    const int gc = dcreg_comm_allgather(ctx, blk.data(), all.data(), per * REC);
    if (rc != DCREG_OK) return rc;
    if (gc != DCREG_OK) return gc;
    // order by trial: record k sits in block k % world at row k / world
    std::vector<uint8_t> seen((size_t)world, 0);
    std::vector<double> sorted((size_t)n_trials * REC);
    for (int64_t k = 0; k < n_trials; ++k) {
        const double *r = &all[((size_t)(k % world) * (size_t)per + (size_t)(k / world)) * REC];
        if ((int64_t)r[9] != k) {
            char msg[160];
            std::snprintf(msg, sizeof(msg), "the gathered record of trial %lld is missing (index %lld in its place): a rank's share failed", (long long)k, (long long)r[9]);
            dcreg_set_error_message(ctx, msg);
            return DCREG_E_STATE;
        }
        std::memcpy(&sorted[(size_t)k * REC], r, sizeof(double) * REC);
        seen[(size_t)(k % world)] = 1;
    }
    if (records) std::memcpy(records, sorted.data(), sizeof(double) * sorted.size());
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        dcreg_method_stats &s = *stats;
        s.total_runs = n_trials; s.world = world;
        for (int r = 0; r < world; ++r) s.ranks_seen += seen[(size_t)r];
        if (n_trials > 0) {
            const double n = (double)n_trials;
            double st = 0, sr = 0, sm = 0, si = 0, srm = 0, sf = 0;
            s.min_trans_error = s.min_rot_error = std::numeric_limits<double>::infinity();
            s.max_trans_error = s.max_rot_error = -std::numeric_limits<double>::infinity();
            for (int64_t k = 0; k < n_trials; ++k) {
                const double *r = &sorted[(size_t)k * REC];
                s.converged_runs += r[0] != 0.0;
                st += r[3]; sr += r[4]; sm += r[2]; si += r[1]; srm += r[5]; sf += r[6];
                s.corr_num += (int64_t)r[7]; s.iterations_total += (int64_t)r[1];
                s.min_trans_error = std::min(s.min_trans_error, r[3]); s.max_trans_error = std::max(s.max_trans_error, r[3]);
                s.min_rot_error = std::min(s.min_rot_error, r[4]); s.max_rot_error = std::max(s.max_rot_error, r[4]);
            }
            s.success_rate = (double)s.converged_runs / n;
            s.mean_trans_error = st / n; s.mean_rot_error = sr / n; s.mean_time_ms = sm / n;
            s.mean_iterations = si / n; s.mean_rmse = srm / n; s.mean_fitness = sf / n;
            double vt = 0, vr = 0, vm = 0;
            for (int64_t k = 0; k < n_trials; ++k) {
                const double *r = &sorted[(size_t)k * REC];
                vt += (r[3] - s.mean_trans_error) * (r[3] - s.mean_trans_error);
                vr += (r[4] - s.mean_rot_error) * (r[4] - s.mean_rot_error);
                vm += (r[2] - s.mean_time_ms) * (r[2] - s.mean_time_ms);
            }
            s.std_trans_error = std::sqrt(vt / n); s.std_rot_error = std::sqrt(vr / n); s.std_time_ms = std::sqrt(vm / n);   // population std (:660-662)
        }
    }
    return DCREG_OK;
}

// Second engine: TestRunner::Point2PlaneICP (icp_test_runner.cpp:2064-2830), Pose6D state, the Jacobian of :2299-2346 (LOAM's brackets,
// the reference's coefficient order; include/dcreg.h enum dcreg_parameterization).
int dcreg_icp_run_euler(dcreg_ctx *ctx, const double pose6d[6], int detection, int handling, const dcreg_config *cfg,
                        dcreg_iter_log *log, int log_capacity, dcreg_icp_result *res, double final_pose6d[6]) {
    if (!ctx || !pose6d || !cfg || !res) return DCREG_E_INVALID;
    std::memset(res, 0, sizeof(*res));
    const auto t_total = Clock::now();
    double pose[6];                                    // roll pitch yaw x y z  (:2086)
    std::memcpy(pose, pose6d, sizeof(pose));
    double Hlast[36];
    for (int i = 0; i < 36; ++i) Hlast[i] = (i % 7 == 0) ? 1.0 : 0.0;
    dcreg_lin_params prm = lin_params_of(*cfg);
    prm.parameterization = cfg->euler_exact_jacobian ? DCREG_PARAM_EULER_EXACT : DCREG_PARAM_EULER;   // default: the row of :2299-2346 as written
    prm.use_weight_derivative = 0;                     // this engine has no weight-derivative term (:2296-2347)
    dcreg_index_info info;
    dcreg_index_info_get(ctx, &info);
    double T[16], R[9], t[3];
    auto pose_matrix = [&]() {
        dcreg::pose6dToMatrix(pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], T);   // Pose6D2Matrix, :2164
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
    };
    pose_matrix();
    if (info.n_source <= 0 || info.n_target <= 0) {    // :2089-2100
        res->status = 3;
        std::memcpy(res->R, R, sizeof(R)); std::memcpy(res->t, t, sizeof(t));
        covariance_of(false, Hlast, res->icp_cov);
        if (final_pose6d) std::memcpy(final_pose6d, pose, sizeof(pose));
        return DCREG_OK;
    }
    double prev_rmse = std::numeric_limits<double>::max(), prev_fitness = 0.0;   // :2115-2116
    (void)dcreg_hint_misalignment(ctx, -1.0);   // nothing known about the start pose (scheduling only: include/dcreg.h)
    for (int it = 0; it < cfg->max_iterations; ++it) {
        const auto t_iter = Clock::now();
        pose_matrix();
        prm.euler_rpy[0] = pose[0]; prm.euler_rpy[1] = pose[1]; prm.euler_rpy[2] = pose[2];
        dcreg_lin_out lo;
        const int rc = dcreg_linearize(ctx, R, t, &prm, &lo);
        if (rc != DCREG_OK) return rc;
        if (lo.n_eff < 10) { res->iterations = it; res->converged = 0; res->status = 1; break; }   // :2272-2286 (final_iterations_ = iterCount)
        const double fitness = (double)lo.n_pt / (double)info.n_source;          // :2289
        const double rmse = std::sqrt(lo.sum_r2 / (double)lo.n_eff);            // :2291
        (void)dcreg_hint_misalignment(ctx, rmse);
        StepOut so;
        dcreg_unpack_hessian(lo.H_upper, so.H);
        dcreg_analyze_degeneracy(so.H, detection, handling, cfg, &so.an);
        dcreg_solve_degenerate_system(so.H, lo.g, handling, cfg, &so.an, so.dx);
        bool finite = true;
        for (double v : so.dx) finite = finite && std::isfinite(v);
        if (!finite) { res->iterations = it; res->converged = 0; res->status = 2; break; }         // :2605-2617
        for (int i = 0; i < 6; ++i) pose[i] += so.dx[i];                         // :2633-2638
        const double d_rmse = rmse - prev_rmse, d_fit = fitness - prev_fitness;  // :2645-2648
        prev_rmse = rmse; prev_fitness = fitness;
        std::memcpy(Hlast, so.H, sizeof(Hlast));                                 // :2649
        if (log && it < log_capacity) {
            dcreg_iter_log &L = log[it];
            std::memset(&L, 0, sizeof(L));
            L.iter_count = it;
            L.effective_points = lo.n_eff; L.corr_pt_count = lo.n_pt;
            L.fitness = fitness; L.rmse = rmse;
            L.objective_value = 0.5 * lo.sum_b2;                                 // :2357
            for (int i = 0; i < 6; ++i) { L.gradient[i] = -lo.g[i]; L.update_dx[i] = so.dx[i]; }   // :2356, :2630
            pose_matrix();
            std::memcpy(L.transform_matrix, T, sizeof(T));                       // :2675
            dcreg::poseError(cfg->gt_matrix, L.transform_matrix, &L.trans_error_vs_gt, &L.rot_error_vs_gt);
            std::memcpy(L.H_upper, lo.H_upper, sizeof(L.H_upper));
            L.analysis = so.an;
            L.iter_time_ms = ms_since(t_iter);
        }
        res->iterations = it + 1;
        if (std::fabs(d_rmse) < 1e-4 && std::fabs(d_fit) < 1e-4) { res->converged = 1; break; }   // :2679-2687
    }
    pose_matrix();
    std::memcpy(res->R, R, sizeof(R)); std::memcpy(res->t, t, sizeof(t));
    if (final_pose6d) std::memcpy(final_pose6d, pose, sizeof(pose));
    // covariance (:2695-2738): H_last^-1, PSD-clamped, mapped through blockdiag(J_euler->lie, I), clamped again
    covariance_of(false, Hlast, res->icp_cov);                                   // 1e6 * I
    double inv[36];
    if (res->converged && dcreg::invertSpd6(Hlast, inv)) {
        using dcreg::Mat6; using dcreg::Vec;
        Mat6 C; std::memcpy(C.v, inv, sizeof(inv));
        clamp_psd6(C, false);
        // computeEulerToLieJacobian, math_utils.hpp:125-136
        const double cr = std::cos(pose[0]), sr = std::sin(pose[0]), cp = std::cos(pose[1]), sp = std::sin(pose[1]);
        dcreg::Mat3 Jl; for (double &x : Jl.v) x = 0.0;
        Jl.v[0] = Jl.v[4] = Jl.v[8] = 1.0;
        if (std::fabs(cp) >= 1e-6) {
            dcreg::Mat3 E; const double e[9] = {1.0, 0.0, sp, 0.0, cr, -sr * cp, 0.0, sr, cr * cp};
            std::memcpy(E.v, e, sizeof(e));
            dcreg::Mat3 Ei;
            if (dcreg::fullPivLuInverse3(E, Ei)) Jl = Ei;
        }
        Mat6 J; for (double &x : J.v) x = 0.0;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) J.v[i * 6 + j] = Jl.v[i * 3 + j]; J.v[(i + 3) * 6 + i + 3] = 1.0; }
        Mat6 Cl = dcreg::mul(dcreg::mul(J, C), dcreg::transpose(J));
        clamp_psd6(Cl, true);
        std::memcpy(res->icp_cov, Cl.v, sizeof(Cl.v));
    }
    res->time_ms = ms_since(t_total);
    return DCREG_OK;
}

// Initial pose of Monte-Carlo trial k - the ONE definition both drivers use (the reference has no RNG):
//   k == 0 : the base pose itself (the reference's deterministic run);
//   k >= 1 : base + U(-a, a) per degree of freedom, u_j = 2 * genrand_res53() - 1 drawn in the order x y z roll pitch yaw
//            from MT19937 (32-bit, init_genrand) seeded with the low 32 bits of seed + k;
//            genrand_res53 = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 of two successive outputs a, b.
// The same numbers as numpy.random.RandomState((seed + k) & 0xFFFFFFFF).random_sample(6) (tested).
int dcreg_trial_pose(const double base_xyzrpy[6], uint64_t seed, int64_t k, double trans_amp, double rot_amp_rad, double T[16],
                     double pose_xyzrpy[6]) {
    if (!base_xyzrpy || !T || k < 0) return DCREG_E_INVALID;
    double p[6];
    std::memcpy(p, base_xyzrpy, sizeof(p));
    if (k > 0) {
        // numpy's RandomState(seed + k): MT19937 seeded by init_genrand, twelve outputs.  Only what those twelve outputs read is computed:
        // output i < 12 is the tempered twist of state words i, i + 1 and i + 397, so the seeding recurrence stops at word 408 and the
        // twist at word 11 (a third of the work of the full generator: 5000 poses were 9 ms of an experiment on two host threads)
        constexpr int kDraws = 12, kNeed = kDraws + 397;
        uint32_t mt[kNeed];
        mt[0] = (uint32_t)((seed + (uint64_t)k) & 0xFFFFFFFFull);
        for (int i = 1; i < kNeed; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        int idx = 0;
        auto next = [&]() -> uint32_t {
            const int i = idx++;
            const uint32_t v = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7FFFFFFFu);
            uint32_t y = mt[i + 397] ^ (v >> 1) ^ ((v & 1u) ? 0x9908B0DFu : 0u);
            y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
            return y;
        };
        for (int j = 0; j < 6; ++j) {
            const uint32_t a = next() >> 5, b = next() >> 6;
            const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            p[j] += (2.0 * u - 1.0) * (j < 3 ? trans_amp : rot_amp_rad);
        }
    }
    dcreg::pose6dToMatrix(p[3], p[4], p[5], p[0], p[1], p[2], T);
    if (pose_xyzrpy) std::memcpy(pose_xyzrpy, p, sizeof(p));
    return DCREG_OK;
}

int dcreg_set_host_threads(int n) {
    if (n < 1) return DCREG_E_INVALID;
    omp_set_num_threads(n);
    return DCREG_OK;
}
int dcreg_get_host_threads(void) { return omp_get_max_threads(); }

size_t dcreg_sizeof(const char *name) {
    if (!name) return 0;
    if (!std::strcmp(name, "dcreg_lin_params")) return sizeof(dcreg_lin_params);
    if (!std::strcmp(name, "dcreg_lin_out")) return sizeof(dcreg_lin_out);
    if (!std::strcmp(name, "dcreg_lin_debug")) return sizeof(dcreg_lin_debug);
    if (!std::strcmp(name, "dcreg_index_info")) return sizeof(dcreg_index_info);
    if (!std::strcmp(name, "dcreg_config")) return sizeof(dcreg_config);
    if (!std::strcmp(name, "dcreg_analysis")) return sizeof(dcreg_analysis);
    if (!std::strcmp(name, "dcreg_iter_log")) return sizeof(dcreg_iter_log);
    if (!std::strcmp(name, "dcreg_icp_result")) return sizeof(dcreg_icp_result);
    if (!std::strcmp(name, "dcreg_trial_result")) return sizeof(dcreg_trial_result);
    if (!std::strcmp(name, "dcreg_launch_stats")) return sizeof(dcreg_launch_stats);
    if (!std::strcmp(name, "dcreg_method_stats")) return sizeof(dcreg_method_stats);
    return 0;
}

const char *dcreg_version(void) { return "dcreg-mi355x 0.1.0 (gfx950)"; }

}  // extern "C"
