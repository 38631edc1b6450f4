// mpb_gpmp2.h -- what the two forms of the GPMP2 solve (mpb_gpmp2.hip: block elimination; mpb_gpmp2_lr.hip: low-rank form) share.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define MPB_GP_MAX_FIELDS 4      // = MPB_MAX_FIELDS of mpb_geom.h (asserted in mpb_gpmp2.hip): collision fields chained in one buffer

// precisions 1 / sigma^2 of the factors, damping and step of one Gauss-Newton iteration (gpmp2.py:308-368)
struct GpConst {
    double dt, ks, kgp, kg, kc, delta, step;
    int trust;
};

// the low-rank form (mpb_gpmp2_lr.hip): can it take the shape, how many doubles of workspace it needs (carved from the section
// that holds the block form's elimination records, which it does not use), and its launches
bool mpb_gpmp2_lr_ok(int H, int D, int n_fields);
size_t mpb_gpmp2_lr_ws_doubles(int B, int H, int D);
int mpb_gpmp2_lr_launch(float* x, const float* start, const float* goal, const float* jac, const double* diag_mean, double* ws,
                        float* costs_out, int B, int H, int D, int n_fields, const GpConst& K, hipStream_t stream);
