// placeholder until the block-tridiagonal GPMP2 kernels land
#include "../../include/mpb.h"
#include "mpb_common.h"
extern "C" int mpb_gpmp2_diag(const float*, const float*, float*, int, int, int, float, float, float, float, float, void*) {
    return mpb_fail(MPB_E_UNSUPPORTED, "mpb_gpmp2_diag: not implemented in this build");
}
extern "C" int mpb_gpmp2_step(float*, const float*, const float*, const float*, const float*, float*, int, int, int, float,
                              float, float, float, float, float, int, float, void*) {
    return mpb_fail(MPB_E_UNSUPPORTED, "mpb_gpmp2_step: not implemented in this build");
}
