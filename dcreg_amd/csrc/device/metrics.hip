// calculatePointToPointError (DCReg/include/utils.hpp:538-589) on the device index.
#include <hip/hip_runtime.h>

#include "../../../include/dcreg.h"
#include "context.hpp"

extern "C" int dcreg_p2p_error(dcreg_ctx *c, const double T[16], double error_threshold, double *rmse, double *fitness,
                               double *chamfer, int64_t *valid) {
    if (!c) return DCREG_E_INVALID;
    c->fail("dcreg_p2p_error: not built yet");
    return DCREG_E_STATE;
}
