// mpb_chomp.hip -- CHOMP (mp_baselines/planners/chomp.py:127-169): the whole optimisation loop of every particle in
// one launch, analytic collision gradient (J^T grad sdf), smoothness gradient from the tridiagonal R.
// (Own translation unit: built with the max-ILP scheduling strategy, which suits this latency-bound kernel --
// C2: 3.6 -> 3.2 us / iteration -- but not the STOMP kernels of mpb_kernels.hip.)
#include <type_traits>

#include "mpb_common.h"
#include "mpb_geom.h"

#define MPB_MAX_D (2 * MPB_MAX_DOF)

// ------------------------------------------------------------------------------------------------
// CHOMP: one workgroup per particle, one thread per waypoint, the whole optimisation loop in one
// launch.  The trajectory tile lives in LDS (the finite-difference stencil reads the h-1 / h+1 rows
// from there); each thread keeps its own row in registers.
// ------------------------------------------------------------------------------------------------
// Lean variant for robots with up to 3 degrees of freedom (point masses, planar arms): exhaustive evaluator only, no
// LDS grid image -- against a few dozen obstacles the straight SGPR-operand loop wins (C2: 3.6 vs 4.0 us / iteration).
__global__ __launch_bounds__(256) void chomp_lean_kernel(float* __restrict__ means, const float* __restrict__ R,
                             const float* __restrict__ geom, float* __restrict__ costs_out, int B_global,
                             int H, int d, int D, float k_sigma, float weight, float w_prior, float lr,
                             float grad_clip, int n_iters) {
    extern __shared__ float tile[];  // H x d
    __shared__ double red[16];
    const int b = blockIdx.x;
    const int h = threadIdx.x;
    const bool active = h < H;
    float x[MPB_MAX_D];
    float* row = means + ((size_t)b * H + (active ? h : 0)) * d;
#pragma unroll
    for (int c = 0; c < MPB_MAX_D; ++c) x[c] = (active && c < d) ? row[c] : 0.f;
    // tridiagonal band of R (chomp.py:81-101) for this row
    const float r_lo = (active && h > 0) ? R[h * H + h - 1] : 0.f;
    const float r_di = active ? R[h * H + h] : 0.f;
    const float r_up = (active && h < H - 1) ? R[h * H + h + 1] : 0.f;
    // d/dx of w_prior * sum_b sum_c x^T R x summed over B costs: (B*w) * (R x + R^T x)
    const float bw = (float)B_global * w_prior;
    const bool interior = active && h > 0 && h < H - 1;
    for (int it = 0; it < n_iters; ++it) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < MPB_MAX_D; ++c)
                if (c < d) tile[h * d + c] = x[c];
        }
        __syncthreads();
        float dq[MPB_MAX_DOF];
        float cw = 0.f;
        if (active && h >= 1) {
            float q[MPB_MAX_DOF];
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = x[i];
            cw = waypoint_cost_chain<true>(geom, q, dq);
        } else {
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
        }
        if (costs_out != nullptr && it == n_iters - 1) {
            double cs = wave_sum_f64((double)cw);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cs;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += red[i];
                costs_out[b] = weight * (k_sigma * (float)t);
            }
        }
        if (interior) {
            const float sc = weight * k_sigma;
#pragma unroll
            for (int c = 0; c < MPB_MAX_D; ++c) {
                if (c < d) {
                    const float xm = tile[(h - 1) * d + c], xp = tile[(h + 1) * d + c];
                    const float rx = r_lo * xm + r_di * x[c] + r_up * xp;
                    float g = bw * (rx + rx);
                    if (c < MPB_MAX_DOF && c < D) g += sc * dq[c < MPB_MAX_DOF ? c : 0];
                    g = fminf(fmaxf(g, -grad_clip), grad_clip);
                    x[c] += -lr * g;
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < MPB_MAX_D; ++c)
            if (c < d) row[c] = x[c];
    }
}


// General variant (D > 3): broad-phase grid for the gradient evaluator, specialised loops per evaluator.
__global__ __launch_bounds__(256) void chomp_kernel(float* __restrict__ means, const float* __restrict__ R,
                             const float* __restrict__ geom, float* __restrict__ costs_out, int B_global,
                             int H, int d, int D, float k_sigma, float weight, float w_prior, float lr,
                             float grad_clip, int n_iters) {
    extern __shared__ float tile[];  // H x d
    __shared__ double red[16];
    __shared__ unsigned gridw[MPB_GRID_MAX_CELLS];              // broad-phase grid of the collision field
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const int b = blockIdx.x;
    const int h = threadIdx.x;
    const bool active = h < H;
    float x[MPB_MAX_D];
    float* row = means + ((size_t)b * H + (active ? h : 0)) * d;
#pragma unroll
    for (int c = 0; c < MPB_MAX_D; ++c) x[c] = (active && c < d) ? row[c] : 0.f;
    // tridiagonal band of R (chomp.py:81-101) for this row
    const float r_lo = (active && h > 0) ? R[h * H + h - 1] : 0.f;
    const float r_di = active ? R[h * H + h] : 0.f;
    const float r_up = (active && h < H - 1) ? R[h * H + h + 1] : 0.f;
    // d/dx of w_prior * sum_b sum_c x^T R x summed over B costs: (B*w) * (R x + R^T x)
    const float bw = (float)B_global * w_prior;
    const bool interior = active && h > 0 && h < H - 1;
    // one collision field (the usual case): its grid is staged once for all iterations; several chained fields
    // take turns in the LDS image inside the loop
    const bool single = geom_next(geom) == nullptr;
    const GeomView G0 = geom_view(geom);
    const bool grid0 = grid_usable_grad(G0);
    if (single && grid0) grid_stage(G0, gridw, otab, threadIdx.x, blockDim.x);
    // the optimisation loop, specialised per evaluator (MODE is a compile-time tag; one of the three runs)
    auto run = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    for (int it = 0; it < n_iters; ++it) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < MPB_MAX_D; ++c)
                if (c < d) tile[h * d + c] = x[c];
        }
        __syncthreads();
        float dq[MPB_MAX_DOF], q[MPB_MAX_DOF];
        float cw = 0.f;
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) {
            q[i] = x[i];
            dq[i] = 0.f;
        }
        const bool eval = active && h >= 1;
        if constexpr (MODE == 0) {
            // single field, exhaustive evaluator (point robots against a few dozen obstacles: C2)
            if (eval) {
                cw = G0.fscale * waypoint_cost<true>(G0, q, dq);
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] *= G0.fscale;
            }
        } else if constexpr (MODE == 1) {
            // single field through its broad-phase grid (staged once, before the loop)
            if (eval) {
                cw = G0.fscale * waypoint_cost_grid_grad(G0, gridw, otab, q, dq);
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] *= G0.fscale;
            }
        } else {
            for (const float* gp = geom; gp != nullptr; gp = geom_next(gp)) {
                const GeomView G = geom_view(gp);
                const bool ug = grid_usable_grad(G);
                __syncthreads();
                if (ug) grid_stage(G, gridw, otab, threadIdx.x, blockDim.x);
                __syncthreads();
                if (eval) {
                    float dqf[MPB_MAX_DOF];
                    cw = fmaf(G.fscale, ug ? waypoint_cost_grid_grad(G, gridw, otab, q, dqf) : waypoint_cost<true>(G, q, dqf), cw);
#pragma unroll
                    for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = fmaf(G.fscale, dqf[i], dq[i]);
                }
            }
        }
        if (costs_out != nullptr && it == n_iters - 1) {
            double cs = wave_sum_f64((double)cw);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cs;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += red[i];
                costs_out[b] = weight * (k_sigma * (float)t);
            }
        }
        if (interior) {
            const float sc = weight * k_sigma;
#pragma unroll
            for (int c = 0; c < MPB_MAX_D; ++c) {
                if (c < d) {
                    const float xm = tile[(h - 1) * d + c], xp = tile[(h + 1) * d + c];
                    const float rx = r_lo * xm + r_di * x[c] + r_up * xp;
                    float g = bw * (rx + rx);
                    if (c < MPB_MAX_DOF && c < D) g += sc * dq[c < MPB_MAX_DOF ? c : 0];
                    g = fminf(fmaxf(g, -grad_clip), grad_clip);
                    x[c] += -lr * g;
                }
            }
        }
    }
    };
    if (single && !grid0) run(std::integral_constant<int, 0>{});
    else if (single) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    if (active) {
#pragma unroll
        for (int c = 0; c < MPB_MAX_D; ++c)
            if (c < d) row[c] = x[c];
    }
}


static bool chomp_shape_ok(int H, int d, int D) {
    return H >= 3 && H <= MPB_MAX_H && D >= 1 && D <= MPB_MAX_DOF && (d == D || d == 2 * D);
}

extern "C" int mpb_chomp_step(float* means, const float* R, const float* geom, float* costs_out, int B_local,
                              int B_global, int H, int d, int D, float k_sigma, float weight, float w_prior, float lr,
                              float grad_clip, int n_iters, void* stream) {
    if (B_local == 0) return MPB_OK;
    if (!means || !R || !geom) return mpb_fail(MPB_E_INVALID, "mpb_chomp_step: null pointer");
    if (B_local < 0 || B_global < B_local || !chomp_shape_ok(H, d, D) || n_iters < 0) return mpb_fail(MPB_E_INVALID, "mpb_chomp_step: bad shape");
    if (B_local == 0 || n_iters == 0) return MPB_OK;
    const int threads = (H + 63) & ~63;
    if (D <= 3)
        hipLaunchKernelGGL(chomp_lean_kernel, dim3(B_local), dim3(threads), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
    else
        hipLaunchKernelGGL(chomp_kernel, dim3(B_local), dim3(threads), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
    return mpb_check_launch("mpb_chomp_step");
}
