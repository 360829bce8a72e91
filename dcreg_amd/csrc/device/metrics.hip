// calculatePointToPointError (DCReg/include/utils.hpp:538-589) on the device index:
//   forward  : every aligned source point (T * p, double arithmetic, float store as pcl::transformPointCloud
//              does for a Matrix4d) -> exact 1-NN in the target index; sum sqrt(d2), and d2 / count where
//              sqrt(d2) < error_threshold
//   backward : every target point -> 1-NN in the aligned cloud.  A rigid motion preserves distances, so the
//              nearest aligned point of q is the image of the nearest source point of T^-1 q; a second grid
//              index is built over the (body-frame) source once and queried with T^-1-transformed targets.
//              (The reference measures float distances between float-rounded aligned points; the difference
//              is rounding of O(1e-7) relative and is covered by the test tolerance.)
#include <cmath>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../../include/dcreg.h"
#include "context.hpp"

using namespace dcreg;

namespace dcreg {
int build_aux_index(dcreg_ctx *c);   // context.hip
}

#define HIP_TRY2(ctx, expr)                                                                      \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return DCREG_E_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

static int reduce_p2p(dcreg_ctx *c, const float *d_d2, int64_t n, double thr, double out[3]) {
    const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    if (c->p2p_part_cap < (size_t)nb * 4) {
        if (c->d_p2p_part) (void)hipFree(c->d_p2p_part);
        c->d_p2p_part = nullptr; c->p2p_part_cap = 0;
        HIP_TRY2(c, hipMalloc((void **)&c->d_p2p_part, (size_t)nb * 4 * sizeof(double)));
        c->p2p_part_cap = (size_t)nb * 4;
    }
    hipLaunchKernelGGL(k_p2p_partial, dim3(nb), dim3(kBlock), 0, c->stream, d_d2, n, thr, c->d_p2p_part);
    HIP_TRY2(c, hipGetLastError());
    std::vector<double> h((size_t)nb * 4);
    HIP_TRY2(c, hipMemcpyAsync(h.data(), c->d_p2p_part, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY2(c, hipStreamSynchronize(c->stream));
    out[0] = out[1] = out[2] = 0.0;
    for (unsigned b = 0; b < nb; ++b) { out[0] += h[(size_t)b * 4]; out[1] += h[(size_t)b * 4 + 1]; out[2] += h[(size_t)b * 4 + 2]; }
    return DCREG_OK;
}

extern "C" int dcreg_p2p_error(dcreg_ctx *c, const double T[16], double error_threshold, double *rmse, double *fitness,
                               double *chamfer, int64_t *valid) {
    if (!c) return DCREG_E_INVALID;
    if (!T || !rmse || !fitness || !chamfer || !valid) { c->fail("null argument"); return DCREG_E_INVALID; }
    (void)roi_deactivate(c);              // the metrics are taken on the whole map (context.hpp, the window index)
    if (c->n_tgt <= 0 || c->n_src <= 0) { c->fail("target / source clouds are not set"); return DCREG_E_STATE; }
    HIP_TRY2(c, hipSetDevice(c->device));
    const int64_t ns = c->n_src, nt = c->n_tgt;
    const int64_t nmax = ns > nt ? ns : nt;
    if (c->nn_idx_cap < (size_t)nmax) { if (c->d_nn_idx) (void)hipFree(c->d_nn_idx); c->d_nn_idx = nullptr; HIP_TRY2(c, hipMalloc((void **)&c->d_nn_idx, sizeof(int32_t) * (size_t)nmax)); c->nn_idx_cap = (size_t)nmax; }
    if (c->nn_d2_cap < (size_t)nmax) { if (c->d_nn_d2) (void)hipFree(c->d_nn_d2); c->d_nn_d2 = nullptr; HIP_TRY2(c, hipMalloc((void **)&c->d_nn_d2, sizeof(float) * (size_t)nmax)); c->nn_d2_cap = (size_t)nmax; }
    // forward: aligned -> target
    PoseArg P{};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P.R[i * 3 + j] = T[i * 4 + j]; P.t[i] = T[i * 4 + 3]; }
    int rc = launch_knn(c, c->grid, c->d_src_raw, ns, 1, 0.0, &P, c->d_nn_idx, c->d_nn_d2);
    if (rc) return rc;
    double fwd[3];
    rc = reduce_p2p(c, c->d_nn_d2, ns, error_threshold, fwd);
    if (rc) return rc;
    // backward: target -> aligned  ==  T^-1 target -> source (body frame)
    rc = build_aux_index(c);
    if (rc) return rc;
    PoseArg Pi{};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Pi.R[i * 3 + j] = P.R[j * 3 + i];
        Pi.t[i] = -(P.R[0 * 3 + i] * P.t[0] + P.R[1 * 3 + i] * P.t[1] + P.R[2 * 3 + i] * P.t[2]);
    }
    rc = launch_knn(c, c->aux_grid, c->d_tgt_raw, nt, 1, 0.0, &Pi, c->d_nn_idx, c->d_nn_d2);
    if (rc) return rc;
    double bwd[3];
    rc = reduce_p2p(c, c->d_nn_d2, nt, INFINITY, bwd);
    if (rc) return rc;
    *rmse = std::sqrt(fwd[1] / (double)ns);                               // utils.hpp:568
    *valid = (int64_t)std::llround(fwd[2]);
    *fitness = (double)*valid / (double)ns;                               // utils.hpp:572
    *chamfer = (fwd[0] / (double)ns + bwd[0] / (double)nt) / 2.0;         // utils.hpp:587
    return DCREG_OK;
}
