// Wall-time breakdown of one ICP step through the C-ABI (GPU box): linearise call (launch + kernel + result wait)
// vs host analyse/solve, for a synthetic cylinder pair of n points.
// build: g++ -O2 scripts/step_probe.cpp -o dcreg_amd/bin/step_probe -Ldcreg_amd/lib -ldcreg_hip -Wl,-rpath,'$ORIGIN/../lib'
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/dcreg.h"
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    const int steps = argc > 2 ? atoi(argv[2]) : 400;
    std::mt19937 rng(1); std::uniform_real_distribution<double> U(0, 1); std::normal_distribution<double> N(0, 0.01);
    std::vector<float> tgt(3 * (size_t)n), src(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double x, y, z;
        if (i & 1) { const double th = 2 * M_PI * U(rng); x = 40 * cos(th); y = 40 * sin(th); z = 20 * U(rng); }
        else { const double r = 40 * sqrt(U(rng)), th = 2 * M_PI * U(rng); x = r * cos(th); y = r * sin(th); z = 0; }
        tgt[3 * i] = (float)(x + N(rng)); tgt[3 * i + 1] = (float)(y + N(rng)); tgt[3 * i + 2] = (float)(z + N(rng));
        src[3 * i] = (float)(tgt[3 * i] + N(rng)); src[3 * i + 1] = (float)(tgt[3 * i + 1] + N(rng)); src[3 * i + 2] = (float)(tgt[3 * i + 2] + N(rng));
    }
    dcreg_ctx *ctx = nullptr;
    if (dcreg_backend_create(&ctx, 0) != DCREG_OK) { fprintf(stderr, "no device\n"); return 1; }
    dcreg_set_target(ctx, tgt.data(), n, 3, 1.0); dcreg_set_source(ctx, src.data(), n, 3);
    dcreg_config cfg; dcreg_default_config(&cfg); cfg.always_compute_schur = 1; cfg.use_weight_derivative = 1; cfg.KAPPA_TARGET = 10; cfg.STD_REG_GAMMA = 100;
    dcreg_lin_params prm; dcreg_default_lin_params(&prm, 1.0); prm.use_weight_derivative = 1;
    for (int timed = 0; timed < 2; ++timed) {
        dcreg_set_option(ctx, "time_kernels", timed);
        double T[16]; dcreg_pose6d_to_matrix(0.0035, -0.0017, 0.0087, 0.05, -0.08, 0.03, T);
        double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, t[3] = {T[3], T[7], T[11]};
        double t_lin = 0, t_host = 0; int cnt = 0;
        double ms; int64_t launches; dcreg_kernel_time(ctx, &ms, &launches, 1);
        for (int k = 0; k < steps + 40; ++k) {
            if (k % 20 == 0) { R[0] = T[0]; R[1] = T[1]; R[2] = T[2]; R[3] = T[4]; R[4] = T[5]; R[5] = T[6]; R[6] = T[8]; R[7] = T[9]; R[8] = T[10]; t[0] = T[3]; t[1] = T[7]; t[2] = T[11]; }
            const auto a = Clock::now();
            dcreg_lin_out lo; if (dcreg_linearize(ctx, R, t, &prm, &lo) != DCREG_OK) { fprintf(stderr, "%s\n", dcreg_last_error(ctx)); return 1; }
            const auto b = Clock::now();
            double H[36], dx[6]; dcreg_analysis an;
            dcreg_unpack_hessian(lo.H_upper, H);
            dcreg_analyze_degeneracy(H, DCREG_SCHUR_CONDITION_NUMBER, DCREG_PRECONDITIONED_CG, &cfg, &an);
            dcreg_solve_degenerate_system(H, lo.g, DCREG_PRECONDITIONED_CG, &cfg, &an, dx);
            dcreg_boxplus(R, t, dx, R, t);
            const auto c = Clock::now();
            if (k >= 40) { t_lin += us(a, b); t_host += us(b, c); ++cnt; }
            if (k == 39) dcreg_kernel_time(ctx, &ms, &launches, 1);
        }
        dcreg_kernel_time(ctx, &ms, &launches, 1);
        printf("n=%d time_kernels=%d: linearize call %.2f us, host analyse+solve %.2f us, step %.2f us; kernel (events) %.2f us over %lld launches\n",
               n, timed, t_lin / cnt, t_host / cnt, (t_lin + t_host) / cnt, launches ? ms * 1e3 / launches : -1.0, (long long)launches);
    }
    dcreg_backend_destroy(ctx);
    return 0;
}
