// mpb_prior.hip -- initial particles from the constant-velocity GP prior.
//
// Replaces OptimizationPlanner.get_random_trajs (base.py:155-202) over MultiMPPrior
// (costs/factors/mp_priors_multi.py:100-110, :213-256).  The reference builds the dense M x M precision
// K^-1 = A^T Q^-1 A (M = 2D*H) in fp64, lets MultivariateNormal turn it into a dense scale_tril
// (flip-Cholesky + triangular solve, torch/distributions/multivariate_normal.py:80-86) and multiplies it
// with eps.  K^-1 is block tridiagonal with blocks (2x2) (x) I_D, so that scale_tril is U^-T with
// K^-1 = U U^T and U block upper-bidiagonal with (2x2 upper-triangular) (x) I_D blocks: every
// (sample, degree of freedom) pair is an independent 2-state chain and  x = mean + U^-T eps  is a forward
// substitution along the horizon.  The host computes the H small 2x2 factors in fp64
// (planners/base.py: gp_prior_factor); this kernel does the substitution, one thread per chain, in fp64.
#include "mpb_common.h"

// 1/x in fp64: v_rcp_f64 + two Newton steps (the IEEE division hipcc emits is ~40 instructions, twice per step)
__device__ __forceinline__ double prior_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__global__ __launch_bounds__(256) void gp_prior_sample_kernel(float* __restrict__ out, const double* __restrict__ means,
                                                              const double* __restrict__ eps,
                                                              const double* __restrict__ Udiag,
                                                              const double* __restrict__ Uoff, int G, int n, int H,
                                                              int D, uint32_t seed_lo, uint32_t seed_hi) {
    const int chain = blockIdx.x * blockDim.x + threadIdx.x;   // (i sample, g mode, d dof)
    if (chain >= n * G * D) return;
    const int d = chain % D, g = (chain / D) % G, i = chain / (D * G);
    const int dim = 2 * D;
    const double* mu = means + (size_t)g * H * dim;
    const double* ep = eps ? eps + ((size_t)i * G + g) * H * dim : nullptr;
    float* o = out + ((size_t)g * n + i) * H * dim;           // particle index = mode * n + sample (base.py:202)
    double yp = 0.0, yv = 0.0;
    for (int t = 0; t < H; ++t) {
        double ep_p, ep_v;
        if (ep) {
            ep_p = ep[t * dim + d];
            ep_v = ep[t * dim + D + d];
        } else {
            const uint4 r = philox4x32_10(make_uint4((uint32_t)chain, (uint32_t)t, 0x6770u, 0u), make_uint2(seed_lo, seed_hi));
            float a, b;
            box_muller(r.x, r.y, a, b);
            ep_p = a; ep_v = b;
        }
        double rp = ep_p, rv = ep_v;
        if (t > 0) {   // rhs -= U_{t-1,t}^T y_{t-1}
            const double* O = Uoff + (size_t)(t - 1) * 4;     // row-major 2x2
            rp -= O[0] * yp + O[2] * yv;
            rv -= O[1] * yp + O[3] * yv;
        }
        const double* U = Udiag + (size_t)t * 3;              // u00, u01, u11  (U_tt = [[u00,u01],[0,u11]])
        yp = rp * prior_rcp(U[0]);
        yv = (rv - U[1] * yp) * prior_rcp(U[2]);
        o[t * dim + d] = (float)(mu[t * dim + d] + yp);
        o[t * dim + D + d] = (float)(mu[t * dim + D + d] + yv);
    }
}

extern "C" int mpb_gp_prior_sample(float* out, const double* means, const double* eps, const double* Udiag,
                                   const double* Uoff, int G, int n, int H, int D, uint64_t seed, void* stream) {
    if (!out || !means || !Udiag || !Uoff) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample: null pointer");
    if (G < 1 || n < 0 || H < 2 || H > 4096 || D < 1 || D > 64) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample: bad shape");
    if (n == 0) return MPB_OK;
    const int chains = n * G * D;
    hipLaunchKernelGGL(gp_prior_sample_kernel, dim3((chains + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, means,
                       eps, Udiag, Uoff, G, n, H, D, (uint32_t)seed, (uint32_t)(seed >> 32));
    return mpb_check_launch("mpb_gp_prior_sample");
}
