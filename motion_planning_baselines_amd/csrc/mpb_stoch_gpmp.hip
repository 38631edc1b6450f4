// mpb_stoch_gpmp.hip -- cost of StochGPMP's samples (stoch_gpmp.py:235-242).
//
// costs = CostComposite.eval(samples) + T * V Sigma^-1 U^T with the cost list of
// build_gpmp2_cost_composite (gpmp2.py:23-91): CostGP.eval (cost_functions.py:271-289: start prior +
// GP factors), CostGoalPrior.eval (:520-536), CostCollision.eval (:171-189).  Sigma^-1 = A^T Q^-1 A is the
// precision of the SAMPLING prior (mp_priors_multi.py:213-251), so v^T Sigma^-1 u = (A v)^T Q^-1 (A u):
// the importance term is the bilinear twin of the GP cost and is evaluated factor by factor -- the
// dense N x N matrix of the reference (N = 2D*H) is never formed.
// One wave per sample trajectory, lane = waypoint; every term is local to (row t, row t+1).
#include "mpb_common.h"
#include "mpb_geom.h"

struct SgConst {
    float dt, ks, kgp, kg, kc;      // cost weights 1/sigma^2 (start, gp, goal prior, collision)
    float ss, sgp, sg;              // sampling-prior weights 1/sigma^2 (start, gp, goal)
    float temperature;
};

__global__ __launch_bounds__(256) void stoch_gpmp_cost_kernel(const float* __restrict__ samples,
                                                              const float* __restrict__ means,
                                                              const float* __restrict__ start,
                                                              const float* __restrict__ goal,
                                                              const float* __restrict__ geom, float* __restrict__ costs,
                                                              int P, int S, int H, int D, SgConst K) {
    __shared__ unsigned gridw[MPB_GRID_MAX_CELLS];
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // sample index p*S + s
    const bool dead = r >= P * S;            // such waves still take part in the block barriers below
    const int Dq = D;
    const int dim = 2 * D;
    // a lane's waypoint row (2D floats: positions, velocities) is fetched as D 8-byte pieces (rows start 8-byte aligned:
    // 2D floats per row); one 4-byte load per element -- 8 per degree of freedom and lane in the factor terms below, each
    // touching 28 cache lines per wave at D = 7 -- kept the memory pipe, not the arithmetic, busy
    auto load_row = [&](const float* base, int t, float (&row)[2 * MPB_MAX_DOF]) {
        const float2* p2 = reinterpret_cast<const float2*>(base + (size_t)t * dim);
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) {
            const float2 v = (i < D) ? p2[i] : make_float2(0.f, 0.f);
            row[2 * i] = v.x;
            row[2 * i + 1] = v.y;
        }
    };
    // ---- collision (waypoint 0 excluded): one pass per chained field, its broad-phase grid staged in LDS
    double acc = 0.0;
    for (const float* gp = geom; gp != nullptr; gp = geom_next(gp)) {
        const GeomView G = geom_view(gp);
        const bool use_grid = grid_usable(G);
        __syncthreads();
        if (use_grid) grid_stage(G, gridw, otab, threadIdx.x, blockDim.x);
        __syncthreads();
        if (dead) continue;
        const float* xs0 = samples + (size_t)r * H * 2 * Dq;
        for (int t = lane + 0; t < H; t += 64) {
            if (t < 1) continue;
            float row[2 * MPB_MAX_DOF];
            load_row(xs0, t, row);
            float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < Dq) ? row[i] : 0.f;
            // (the compile-time robot model where the geometry buffer carries its id -- set by pack_geometry, verified by
            // mpb_geom_check against the model's constants; bit-identical to the table-driven walk)
            float c;
            if (use_grid && G.model == PandaModel::ID) c = waypoint_cost_grid_model<PandaModel>(G, gridw, otab, q);
            else c = use_grid ? waypoint_cost_grid(G, gridw, otab, q) : waypoint_cost<false>(G, q, dq);
            acc += (double)K.kc * (double)(G.fscale * c);
        }
    }
    if (dead) return;
    const int p = r / S;
    const float dt = K.dt;
    const float* xs = samples + (size_t)r * H * dim;
    const float* us = means + (size_t)p * H * dim;
    // Qi = [[12/dt^3, -6/dt^2],[-6/dt^2, 4/dt]] (gp_factor.py:42-50), scaled per use
    const double qa = 12.0 / ((double)dt * dt * dt), qb = -6.0 / ((double)dt * dt), qc = 4.0 / (double)dt;
    for (int base = 0; base < H; base += 64) {
        const int t = base + lane;
        const bool on = t < H;
        float x0[2 * MPB_MAX_DOF], u0[2 * MPB_MAX_DOF], x1[2 * MPB_MAX_DOF], u1[2 * MPB_MAX_DOF];
        load_row(xs, on ? t : 0, x0);
        load_row(us, on ? t : 0, u0);
        // row t + 1 is the next lane's row t (lane 63: the first row of the next 64-waypoint chunk, fetched by itself)
#pragma unroll
        for (int k = 0; k < 2 * MPB_MAX_DOF; ++k) {
            x1[k] = (k < dim) ? __shfl_down(x0[k], 1, 64) : 0.f;
            u1[k] = (k < dim) ? __shfl_down(u0[k], 1, 64) : 0.f;
        }
        if (lane == 63 && t + 1 < H) {
            load_row(xs, t + 1, x1);
            load_row(us, t + 1, u1);
        }
        if (!on) continue;
        for (int i = 0; i < D; ++i) {
            const double xp = x0[i], xv = x0[D + i], up = u0[i], uv = u0[D + i];
            if (t == 0) {   // start prior (unary_factor.py:24) and its bilinear twin
                const double ep = (double)start[(size_t)p * dim + i] - xp, ev = (double)start[(size_t)p * dim + D + i] - xv;
                acc += (double)K.ks * (ep * ep + ev * ev);
                acc += (double)K.temperature * (double)K.ss * (xp * up + xv * uv);
            }
            if (t == H - 1) {   // goal prior
                const double ep = (double)goal[(size_t)p * dim + i] - xp, ev = (double)goal[(size_t)p * dim + D + i] - xv;
                acc += (double)K.kg * (ep * ep + ev * ev);
                acc += (double)K.temperature * (double)K.sg * (xp * up + xv * uv);
            }
            if (t < H - 1) {   // GP factor t: e = x_{t+1} - Phi x_t (gp_factor.py:52-56)
                const double xp1 = x1[i], xv1 = x1[D + i], up1 = u1[i], uv1 = u1[D + i];
                const double exp_ = xp1 - (xp + (double)dt * xv), exv = xv1 - xv;
                const double eup = up1 - (up + (double)dt * uv), euv = uv1 - uv;
                acc += (double)K.kgp * (qa * exp_ * exp_ + 2.0 * qb * exp_ * exv + qc * exv * exv);
                acc += (double)K.temperature * (double)K.sgp *
                       (qa * exp_ * eup + qb * (exp_ * euv + exv * eup) + qc * exv * euv);
            }
        }
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) costs[r] = (float)acc;
}

extern "C" int mpb_stoch_gpmp_costs(const float* samples, const float* means, const float* start, const float* goal,
                                    const float* geom, float* costs, int P, int S, int H, int D, float dt,
                                    float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                                    float sigma_start_sample, float sigma_gp_sample, float sigma_goal_sample,
                                    float temperature, void* stream) {
    if (!samples || !means || !start || !goal || !geom || !costs) return mpb_fail(MPB_E_INVALID, "mpb_stoch_gpmp_costs: null pointer");
    if (P < 0 || S < 1 || H < 2 || H > 4096 || D < 1 || D > MPB_MAX_DOF) return mpb_fail(MPB_E_INVALID, "mpb_stoch_gpmp_costs: bad shape");
    if (P == 0) return MPB_OK;
    SgConst K;
    K.dt = dt;
    K.ks = 1.f / (sigma_start * sigma_start); K.kgp = 1.f / (sigma_gp * sigma_gp);
    K.kg = 1.f / (sigma_goal * sigma_goal); K.kc = 1.f / (sigma_coll * sigma_coll);
    K.ss = 1.f / (sigma_start_sample * sigma_start_sample); K.sgp = 1.f / (sigma_gp_sample * sigma_gp_sample);
    K.sg = 1.f / (sigma_goal_sample * sigma_goal_sample);
    K.temperature = temperature;
    const int B = P * S;
    hipLaunchKernelGGL(stoch_gpmp_cost_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, samples, means, start,
                       goal, geom, costs, P, S, H, D, K);
    return mpb_check_launch("mpb_stoch_gpmp_costs");
}

// ------------------------------------------------------------------------------------------------
// The whole StochGPMP.optimize loop (stoch_gpmp.py:281-313) in one call: per iteration the fp64 copy of the means the
// sampler starts from, the sampler (dense MFMA form when scale_tril is given and H <= 128, else the chain form), the
// costs, and the update without covariance product -- the three entry points above, enqueued back to back.
// ------------------------------------------------------------------------------------------------
int mpb_gp_prior_dense_launch(float* out, const double* means, const float* means32, const double* eps,
                              const double* scale_tril, int G, int n, int H, int D, uint64_t seed, void* stream);   // mpb_prior.hip

__global__ void sg_widen_kernel(const float* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

extern "C" int mpb_stoch_gpmp_step(float* means, double* means64, float* samples, float* costs, float* weights,
                                   const double* Udiag, const double* Uoff, const double* scale_tril, const float* start,
                                   const float* goal, const float* geom, int P, int S, int H, int D, float dt,
                                   float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                                   float sigma_start_sample, float sigma_gp_sample, float sigma_goal_sample,
                                   float temperature, float step_size, int n_iters, uint64_t seed, void* stream) {
    if (!means || !means64 || !samples || !costs || !weights || !Udiag || !Uoff || !start || !goal || !geom)
        return mpb_fail(MPB_E_INVALID, "mpb_stoch_gpmp_step: null pointer");
    if (P < 0 || S < 1 || H < 2 || H > MPB_MAX_H || D < 1 || D > MPB_MAX_DOF || n_iters < 0)
        return mpb_fail(MPB_E_INVALID, "mpb_stoch_gpmp_step: bad shape");
    if (P == 0) return MPB_OK;
    const size_t n = (size_t)P * H * 2 * D;
    for (int it = 0; it < n_iters; ++it) {
        int rc;
        if (scale_tril != nullptr && H <= 128) {      // dense MFMA sampler: reads the fp32 means directly
            rc = mpb_gp_prior_dense_launch(samples, nullptr, means, nullptr, scale_tril, P, S, H, D, seed + (uint64_t)it, stream);
        } else {
            hipLaunchKernelGGL(sg_widen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, means,
                               means64, n);
            rc = mpb_gp_prior_sample(samples, means64, nullptr, Udiag, Uoff, P, S, H, D, seed + (uint64_t)it, stream);
        }
        if (rc) return rc;
        rc = mpb_stoch_gpmp_costs(samples, means, start, goal, geom, costs, P, S, H, D, dt, sigma_start, sigma_gp, sigma_goal,
                                  sigma_coll, sigma_start_sample, sigma_gp_sample, sigma_goal_sample, temperature, stream);
        if (rc) return rc;
        rc = mpb_stomp_update(means, samples, costs, weights, nullptr, P, S, H, 2 * D, step_size, temperature, stream);
        if (rc) return rc;
    }
    return mpb_check_launch("mpb_stoch_gpmp_step");
}
