// mpb_mppi.hip -- MPPI on point-particle dynamics, one workgroup per problem, one thread per control sample.
//
// Replaces MPPI.optimize's loop body (mppi.py:145-152): ControlTrajectoryGaussian.sample
// (priors/gaussian.py:276-298: per control dim, mean + scale_tril @ eps), the sequential Euler rollout
// (mppi.py:205-209 over PointParticleDynamics.dynamics, point.py:102-140, deterministic), traj_cost
// (point.py:154-226 incl. quirks Q6 / Q8), the importance-sampling term (mppi.py:125-128) and
// update_controller (mppi.py:72-86).  All `n_iters` iterations run inside one launch.
//
// Only velocity control is served (state_dim == control_dim): with control_type='acceleration' the
// reference's dynamics slices an empty tensor (point.py:114-118 uses the doubled self.state_dim) and
// cannot run, so there is nothing to match.
#include "mpb_common.h"
#include "mpb_geom.h"

#define MPPI_MAX_C 4

__device__ __forceinline__ float block_sum(float v, float* red, int lane, int wave, int nw) {
    v = wave_sum_f32(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red, int lane, int wave, int nw) {
    v = wave_max_f32(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

__global__ __launch_bounds__(1024) void mppi_kernel(
    float* __restrict__ mean, const float* __restrict__ eps, const float* __restrict__ tril,
    const float* __restrict__ cov_inv, const float* __restrict__ state0, const float* __restrict__ goal,
    const float* __restrict__ ctrl_min, const float* __restrict__ ctrl_max, const float* __restrict__ discount,
    const float* __restrict__ cw, const float* __restrict__ geom, float* __restrict__ controls,
    float* __restrict__ states, float* __restrict__ costs, float* __restrict__ weights, int S, int T, int c,
    float dt, float k_sigma, float weight, float temp, float step_size, int n_iters, uint32_t seed_lo,
    uint32_t seed_hi, uint32_t iter0) {
    extern __shared__ float lds[];
    float* wvec = lds;           // c*T : Cov_inv[i] @ mean[:, i]
    float* red = wvec + c * T;   // 32 floats of scratch for block reductions
    float* wts = red + 32;       // S : sample weights
    const int SP = blockDim.x;   // controls slab in LDS, sample index fastest: Us[(t*c + i)*SP + s] (conflict-free)
    float* Us = wts + ((S + 3) & ~3);
    const int prob = blockIdx.x;
    const int s = threadIdx.x;
    const bool live = s < S;
    const int lane = s & 63, wave = s >> 6, nw = blockDim.x >> 6;
    float* m = mean + (size_t)prob * T * c;
    float* U = controls + ((size_t)prob * S + (live ? s : 0)) * T * c;
    float* X = states + ((size_t)prob * S + (live ? s : 0)) * T * c;  // velocity control: state_dim == c
    const float w_pos = cw[0], w_ctrl = cw[2], w_posT = cw[3];
    GeomView G;
    if (geom != nullptr) G = geom_view(geom);

    for (int it = 0; it < n_iters; ++it) {
        __syncthreads();
        // ---- w_i = Cov_inv[i] @ mean_i (vector of the importance-sampling term)
        for (int e = threadIdx.x; e < c * T; e += blockDim.x) {
            const int i = e / T, t = e - i * T;
            float a = 0.f;
            for (int k = 0; k < T; ++k) a = fmaf(cov_inv[((size_t)i * T + t) * T + k], m[k * c + i], a);
            wvec[e] = a;
        }
        __syncthreads();
        float cost = 0.f, coll = 0.f;
        if (live) {
            // ---- standard normals into this sample's controls slab, then U = mean + L eps in place
            //      (L lower triangular: row t only needs eps[k <= t], so sweep t downwards)
            for (int i = 0; i < c; ++i) {
                if (eps != nullptr) {
                    const float* ep = eps + ((((size_t)it * gridDim.x + prob) * c + i) * S + s) * T;
                    for (int t = 0; t < T; ++t) Us[(t * c + i) * SP + s] = ep[t];
                } else {
                    for (int t4 = 0; t4 < T; t4 += 4) {
                        const uint4 r = philox4x32_10(
                            make_uint4((uint32_t)prob, (uint32_t)s, (uint32_t)(t4 >> 2) | ((uint32_t)i << 16), iter0 + (uint32_t)it),
                            make_uint2(seed_lo, seed_hi));
                        float n[4];
                        box_muller(r.x, r.y, n[0], n[1]);
                        box_muller(r.z, r.w, n[2], n[3]);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (t4 + q < T) Us[((t4 + q) * c + i) * SP + s] = n[q];
                    }
                }
                for (int t = T - 1; t >= 0; --t) {
                    const float* trow = tril + ((size_t)i * T + t) * T;   // wave-uniform row: scalar loads
                    float a = 0.f;
#pragma unroll 8
                    for (int k = 0; k <= t; ++k) a = fmaf(trow[k], Us[(k * c + i) * SP + s], a);
                    const float u = m[t * c + i] + a;
                    Us[(t * c + i) * SP + s] = u;
                    U[t * c + i] = u;                                      // API-visible controls
                }
            }
            // ---- Euler rollout (mppi.py:205-209) + quadratic cost (point.py:198-226)
            float x[MPPI_MAX_C], is_term[MPPI_MAX_C];
#pragma unroll
            for (int i = 0; i < MPPI_MAX_C; ++i) {
                x[i] = (i < c) ? state0[(size_t)prob * c + i] : 0.f;
                is_term[i] = 0.f;
            }
            float pos_cost = 0.f, ctl_cost = 0.f, term = 0.f;
            for (int t = 0; t < T; ++t) {
                float pc = 0.f, cc = 0.f, tc = 0.f;
#pragma unroll
                for (int i = 0; i < MPPI_MAX_C; ++i) {
                    if (i < c) {
                        X[t * c + i] = x[i];
                        const float dx = x[i] - goal[(size_t)prob * c + i];
                        pc += dx * dx * w_pos;
                        tc += dx * dx * w_posT;
                        const float u = Us[(t * c + i) * SP + s];
                        cc += u * u * w_ctrl;
                        is_term[i] = fmaf(u, wvec[i * T + t], is_term[i]);
                    }
                }
                pos_cost += pc * discount[t];
                ctl_cost += cc * discount[t];
                if (t == T - 1) term = tc * discount[T - 1];
                if (geom != nullptr && t >= 1) {
                    float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
#pragma unroll
                    for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < MPPI_MAX_C && i < c) ? x[i < MPPI_MAX_C ? i : 0] : 0.f;
                    coll += waypoint_cost<false>(G, q, dq);
                }
                if (t < T - 1) {
#pragma unroll
                    for (int i = 0; i < MPPI_MAX_C; ++i) {
                        if (i < c) {
                            const float u = fminf(fmaxf(Us[(t * c + i) * SP + s], ctrl_min[i]), ctrl_max[i]);  // point.py:112
                            x[i] = x[i] + u * dt;                                                   // point.py:139
                        }
                    }
                }
            }
            cost = pos_cost + 0.f /* vel_cost: empty slice, quirk Q8 */ + ctl_cost + term;
#pragma unroll
            for (int i = 0; i < MPPI_MAX_C; ++i)
                if (i < c) cost += temp * is_term[i];
        }
        // ---- quirk Q6: the per-sample collision costs collapse into ONE scalar added to every sample
        if (geom != nullptr) {
            const float total = block_sum(live ? weight * (k_sigma * coll) : 0.f, red, lane, wave, nw);
            cost += total;
        }
        // ---- softmax over samples (mppi.py:73-76)
        const float xs = live ? -cost / temp : -3.0e38f;
        const float mx = block_max(xs, red, lane, wave, nw);
        const float ex = live ? expf(xs - mx) : 0.f;
        const float z = block_sum(ex, red, lane, wave, nw);
        const float w = ex / z;
        if (live) {
            wts[s] = w;
            costs[(size_t)prob * S + s] = cost;
            weights[(size_t)prob * S + s] = w;
        }
        __threadfence_block();
        __syncthreads();
        // ---- mean += step * sum_s w_s (U_s - mean)   (mppi.py:79-84), one thread per (t, i)
        for (int e = threadIdx.x; e < T * c; e += blockDim.x) {
            const float mu = m[e];
            float a = 0.f;
            for (int ss = 0; ss < S; ++ss) a += wts[ss] * (Us[e * SP + ss] - mu);
            m[e] = mu + step_size * a;
        }
        __threadfence_block();
    }
}

extern "C" int mpb_mppi_step(float* mean, const float* eps, const float* scale_tril, const float* cov_inv,
                             const float* state0, const float* goal, const float* ctrl_min, const float* ctrl_max,
                             const float* discount, const float* c_weights, const float* geom, float* controls,
                             float* states, float* costs, float* weights, int NP, int S, int T, int c, int control_type,
                             float dt, float k_sigma, float weight, float temp, float step_size, int n_iters,
                             uint64_t seed, uint32_t iter0, void* stream) {
    if (!mean || !scale_tril || !cov_inv || !state0 || !goal || !ctrl_min || !ctrl_max || !discount || !c_weights ||
        !controls || !states || !costs || !weights)
        return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: null pointer");
    if (NP < 0 || S < 1 || S > 1024 || T < 2 || T > MPB_MAX_H || c < 1 || c > MPPI_MAX_C || n_iters < 0)
        return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: bad shape");
    if (!(temp > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: temp must be > 0");
    if (control_type != 0)
        return mpb_fail(MPB_E_UNSUPPORTED, "mpb_mppi_step: only velocity control (the reference's acceleration mode cannot run)");
    if (NP == 0 || n_iters == 0) return MPB_OK;
    const int threads = (S + 63) & ~63;
    const size_t lds = ((size_t)c * T + 32 + ((S + 3) & ~3) + (size_t)T * c * threads) * sizeof(float);
    if (lds > 150 * 1024) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_mppi_step: S*T*c too large for the LDS controls slab");
    hipLaunchKernelGGL(mppi_kernel, dim3(NP), dim3(threads), lds, (hipStream_t)stream, mean, eps, scale_tril, cov_inv,
                       state0, goal, ctrl_min, ctrl_max, discount, c_weights, geom, controls, states, costs, weights, S, T,
                       c, dt, k_sigma, weight, temp, step_size, n_iters, (uint32_t)seed, (uint32_t)(seed >> 32), iter0);
    return mpb_check_launch("mpb_mppi_step");
}
