// placeholder until the MPPI kernel lands
#include "../../include/mpb.h"
#include "mpb_common.h"
extern "C" int mpb_mppi_step(float*, const float*, const float*, const float*, const float*, const float*, const float*,
                             const float*, const float*, const float*, const float*, float*, float*, float*, float*, int,
                             int, int, int, int, float, float, float, float, int, uint64_t, uint32_t, void*) {
    return mpb_fail(MPB_E_UNSUPPORTED, "mpb_mppi_step: not implemented in this build");
}
