// mpb_chomp.hip -- CHOMP (mp_baselines/planners/chomp.py:127-169): the whole optimisation loop of every particle in
// one launch, analytic collision gradient (J^T grad sdf), smoothness gradient from the tridiagonal R.
// (Own translation unit: built with the max-ILP scheduling strategy, which suits this latency-bound kernel --
// C2: 3.6 -> 3.2 us / iteration -- but not the STOMP kernels of mpb_kernels.hip.)
#include <stdlib.h>
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
                    const float rx = fmaf(r_up, xp, fmaf(r_di, x[c], r_lo * xm));   // explicit: the three terms cancel to ~1e-7 of their size, every CHOMP kernel must round alike
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


// Point robots, FOUR lanes per waypoint, obstacles in registers (round 2; C2's kernel).  The lean kernel above is one
// wave per particle walking a serial chain of ~700 instructions per iteration, and every block of four obstacles costs a
// ~150-cycle wait on the scalar cache with one wave per SIMD to hide it.  Here the obstacles of the (single) field are
// dealt round-robin to the four lanes of a quad -- at most 32 spheres and 8 boxes: eight spheres and two boxes per lane,
// loaded ONCE into registers, so an iteration issues no obstacle load at all -- each lane keeps its own running nearest,
// and the quad combines with two DPP exchanges on (signed distance, obstacle index): the lower index wins a tie, which
// is the exhaustive loop's "first minimum in obstacle order".  Same arithmetic per obstacle as point_cost<true>.  All
// four lanes carry the waypoint's row redundantly.  The launcher takes this kernel only when geom_flags says the buffer
// qualifies (MPB_GEOM_FLAG_POINT_SMALL); the kernel re-checks the device header and poisons the trajectory otherwise.
template <int CTRL>
__device__ __forceinline__ float quad_f32(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ __forceinline__ int quad_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

__global__ __launch_bounds__(1024) void chomp_point4_kernel(float* __restrict__ means, const float* __restrict__ R,
                             const float* __restrict__ geom, float* __restrict__ costs_out, int B_global,
                             int H, int d, int D, float k_sigma, float weight, float w_prior, float lr,
                             float grad_clip, int n_iters) {
    extern __shared__ float tile[];  // H x d
    __shared__ double red[16];
    const int b = blockIdx.x;
    const int h = threadIdx.x >> 2, sub = threadIdx.x & 3;
    const bool active = h < H;
    float x[6];
    float* row = means + ((size_t)b * H + (active ? h : 0)) * d;
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = (active && c < d) ? row[c] : 0.f;
    const float r_lo = (active && h > 0) ? R[h * H + h - 1] : 0.f;
    const float r_di = active ? R[h * H + h] : 0.f;
    const float r_up = (active && h < H - 1) ? R[h * H + h + 1] : 0.f;
    const float bw = (float)B_global * w_prior;
    const bool interior = active && h > 0 && h < H - 1;
    const GeomView G = geom_view(geom);
    if (G.kind != MPB_KIND_POINT || G.next != 0 || G.n_sph > 32 || G.n_box > 8) {     // geom_flags lied: no silent mis-read
        if (active && sub == 0)
            for (int c = 0; c < d && c < 6; ++c) reinterpret_cast<unsigned*>(row)[c] = 0x7FC00000u;
        return;
    }
    float4 sreg[8], bcen[2], bhal[2];
    {
        const float4* sp = reinterpret_cast<const float4*>(G.sph);
        const float4* bp = reinterpret_cast<const float4*>(G.box);
#pragma unroll
        for (int k = 0; k < 8; ++k) sreg[k] = (sub + 4 * k < G.n_sph) ? sp[sub + 4 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool on = sub + 4 * k < G.n_box;
            bcen[k] = on ? bp[2 * (sub + 4 * k)] : make_float4(0.f, 0.f, 0.f, 0.f);
            bhal[k] = on ? bp[2 * (sub + 4 * k) + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float thr = G.margin + G.links[4];
    for (int it = 0; it < n_iters; ++it) {
        __syncthreads();
        if (active && sub == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (c < d) tile[h * d + c] = x[c];
        }
        __syncthreads();
        const float px0 = x[0], py0 = x[1], pz0 = (G.n_dof > 2) ? x[2] : 0.f;
        float best = 3.0e38f, vx = 0.f, vy = 0.f, vz = 0.f, vn = 1.f;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = sub + 4 * k;
            const float4 s = sreg[k];
            const float dx = px0 - s.x, dy = py0 - s.y, dz = pz0 - s.z;
            float d2 = dx * dx + dy * dy + dz * dz;
            d2 = fmaxf(d2, 1e-30f);
            const float dist = fast_sqrt(d2);
            const float sd = (o < G.n_sph) ? dist - s.w : 3.0e38f;      // an empty slot is never the nearest
            const bool better = sd < best;
            vx = better ? dx : vx; vy = better ? dy : vy; vz = better ? dz : vz; vn = better ? dist : vn;
            bidx = better ? o : bidx;
            best = fminf(best, sd);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o = sub + 4 * k;
            const float4 c = bcen[k], hh = bhal[k];
            const float px = px0 - c.x, py = py0 - c.y, pz = pz0 - c.z;
            const float ax = fabsf(px) - hh.x, ay = fabsf(py) - hh.y, az = fabsf(pz) - hh.z;
            const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
            float o2 = qx * qx + qy * qy + qz * qz;
            o2 = fmaxf(o2, 1e-30f);
            const float outside = fast_sqrt(o2);
            const float mx = fmaxf(ax, fmaxf(ay, az));
            const float sd = (o < G.n_box) ? outside + fminf(mx, 0.f) : 3.0e38f;
            const bool better = sd < best;
            const bool out = mx > 0.f;
            const bool ix = (ax >= ay) && (ax >= az);
            const bool iy = !ix && (ay >= az);
            const float nx = out ? copysignf(qx, px) : (ix ? copysignf(1.f, px) : 0.f);
            const float ny = out ? copysignf(qy, py) : (iy ? copysignf(1.f, py) : 0.f);
            const float nz = out ? copysignf(qz, pz) : ((!ix && !iy) ? copysignf(1.f, pz) : 0.f);
            vx = better ? nx : vx; vy = better ? ny : vy; vz = better ? nz : vz;
            vn = better ? (out ? outside : 1.f) : vn;
            bidx = better ? G.n_sph + o : bidx;
            best = fminf(best, sd);
        }
        // nearest over the quad: lower signed distance, then lower obstacle index (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E)
#define CHOMP4_COMBINE(CTRL)                                                                         \
        {                                                                                              \
            const float ob = quad_f32<CTRL>(best), ox = quad_f32<CTRL>(vx), oy = quad_f32<CTRL>(vy);   \
            const float oz = quad_f32<CTRL>(vz), on = quad_f32<CTRL>(vn);                              \
            const int oi = quad_i32<CTRL>(bidx);                                                       \
            const bool take = (ob < best) || (ob == best && oi < bidx);                                \
            best = take ? ob : best; vx = take ? ox : vx; vy = take ? oy : vy; vz = take ? oz : vz;    \
            vn = take ? on : vn; bidx = take ? oi : bidx;                                              \
        }
        CHOMP4_COMBINE(0xB1)
        CHOMP4_COMBINE(0x4E)
#undef CHOMP4_COMBINE
        const float hng = fmaxf(thr - best, 0.f);
        const float scn = (hng > 0.f) ? -1.0f / vn : 0.f;
        const bool eval = active && h >= 1;
        const float cw = eval ? G.fscale * hng : 0.f;
        const float dqv[3] = {eval ? G.fscale * (vx * scn) : 0.f, eval ? G.fscale * (vy * scn) : 0.f,
                              (eval && G.n_dof > 2) ? G.fscale * (vz * scn) : 0.f};
        if (costs_out != nullptr && it == n_iters - 1) {
            double cs = wave_sum_f64((double)(sub == 0 ? cw : 0.f));
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
            for (int c = 0; c < 6; ++c) {
                if (c < d) {
                    const float xm = tile[(h - 1) * d + c], xp = tile[(h + 1) * d + c];
                    const float rx = fmaf(r_up, xp, fmaf(r_di, x[c], r_lo * xm));   // explicit: the three terms cancel to ~1e-7 of their size, every CHOMP kernel must round alike
                    float g = bw * (rx + rx);
                    if (c < 3 && c < D) g += sc * dqv[c < 3 ? c : 0];
                    g = fminf(fmaxf(g, -grad_clip), grad_clip);
                    x[c] += -lr * g;
                }
            }
        }
    }
    if (active && sub == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (c < d) row[c] = x[c];
    }
}

// General variant (D > 3): broad-phase grid for the gradient evaluator, specialised loops per evaluator.
// MODEL != 0: every chained field carries that compile-time robot model and a usable grid (launcher: geom_flags); the
// tag is re-checked on the device (a mismatch poisons the trajectory with NaNs instead of mis-reading the buffer).
template <int MODEL>
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
                if (MODEL == PandaModel::ID) {
                    cw = (G0.model == PandaModel::ID) ? G0.fscale * waypoint_cost_grid_grad_model<PandaModel>(G0, gridw, otab, q, dq)
                                                     : __uint_as_float(0x7FC00000u);
                    if (G0.model != PandaModel::ID) {
#pragma unroll
                        for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = cw;
                    }
                } else {
                    cw = G0.fscale * waypoint_cost_grid_grad(G0, gridw, otab, q, dq);
                }
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
                    float cf;
                    if (MODEL == PandaModel::ID) {
                        if (ug && G.model == PandaModel::ID) {
                            cf = waypoint_cost_grid_grad_model<PandaModel>(G, gridw, otab, q, dqf);
                        } else {
                            cf = __uint_as_float(0x7FC00000u);
#pragma unroll
                            for (int i = 0; i < MPB_MAX_DOF; ++i) dqf[i] = cf;
                        }
                    } else {
                        cf = ug ? waypoint_cost_grid_grad(G, gridw, otab, q, dqf) : waypoint_cost<true>(G, q, dqf);
                    }
                    cw = fmaf(G.fscale, cf, cw);
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
                    const float rx = fmaf(r_up, xp, fmaf(r_di, x[c], r_lo * xm));   // explicit: the three terms cancel to ~1e-7 of their size, every CHOMP kernel must round alike
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

extern "C" int mpb_chomp_step(float* means, const float* R, const float* geom, int geom_flags, float* costs_out, int B_local,
                              int B_global, int H, int d, int D, float k_sigma, float weight, float w_prior, float lr,
                              float grad_clip, int n_iters, void* stream) {
    if (B_local == 0) return MPB_OK;
    if (!means || !R || !geom) return mpb_fail(MPB_E_INVALID, "mpb_chomp_step: null pointer");
    if (B_local < 0 || B_global < B_local || !chomp_shape_ok(H, d, D) || n_iters < 0) return mpb_fail(MPB_E_INVALID, "mpb_chomp_step: bad shape");
    if (B_local == 0 || n_iters == 0) return MPB_OK;
    const int threads = (H + 63) & ~63;
    static const bool no_p4 = getenv("MPB_CHOMP_LEAN") != nullptr;    // A/B aid
    if ((geom_flags & 0x200) && D <= 3 && 4 * H <= 1024 && !no_p4) {  // point robot, one field, <= 32 spheres + 8 boxes
        hipLaunchKernelGGL(chomp_point4_kernel, dim3(B_local), dim3((4 * H + 63) & ~63), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
        return mpb_check_launch("mpb_chomp_step");
    }
    if (D <= 3)
        hipLaunchKernelGGL(chomp_lean_kernel, dim3(B_local), dim3(threads), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
    else if ((geom_flags & 0xFF) == PandaModel::ID && (geom_flags & 0x100) && D == PandaModel::N_DOF)
        hipLaunchKernelGGL(chomp_kernel<PandaModel::ID>, dim3(B_local), dim3(threads), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
    else
        hipLaunchKernelGGL(chomp_kernel<0>, dim3(B_local), dim3(threads), (size_t)H * d * 4, (hipStream_t)stream, means,
                           R, geom, costs_out, B_global, H, d, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters);
    return mpb_check_launch("mpb_chomp_step");
}
