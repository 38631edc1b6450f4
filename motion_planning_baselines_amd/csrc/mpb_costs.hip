// mpb_costs.hip -- the trajectory-only cost terms of costs/cost_functions.py in ONE streaming pass.
//
// Replaces the eval() of CostGP (cost_functions.py:271-289), CostGPTrajectory (:344-354),
// CostGPTrajectoryPositionOnlyWrapper (:365-368), CostSmoothnessCHOMP (:384-387), CostJointLimits
// (:406-426) and CostGoalPrior (:523-536).  The reference launches a handful of broadcast / matmul ATen
// ops per term and re-reads the (B,H,d) batch for each; here one wave owns one trajectory, stages it
// once in LDS with coalesced loads, and every enabled term is evaluated from that tile (lane = waypoint),
// so a CostComposite of several such terms costs one read of the batch.  HBM-bound: algorithmic bytes
// 4*B*H*d read + 4*B written.  Sums are carried in fp64 (free: the kernel waits on memory).
#include "mpb_common.h"

#define COSTS_WAVES 4

__global__ __launch_bounds__(64 * COSTS_WAVES) void cost_terms_kernel(
    const float* __restrict__ trajs, float* __restrict__ out, double* __restrict__ jl_total,
    const float* __restrict__ start_state, const float* __restrict__ goal_states, const float* __restrict__ q_min,
    const float* __restrict__ q_max, int B, int H, int d, int D, int trajs_per_goal, uint32_t flags, float dt,
    float k_gp, float k_start, float k_goal, float k_smooth, float k_jlim, float jl_eps, int accumulate) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * COSTS_WAVES + wave;
    if (b >= B) return;                                   // whole wave leaves together; no block barriers below
    const int LD = d | 1;                                 // odd row stride: lanes (= rows) hit distinct banks
    float* X = lds + (size_t)wave * H * LD;               // this wave's trajectory tile, rows of LD words
    const float* src = trajs + (size_t)b * H * d;
    const int n = H * d;
    for (int e = lane; e < n; e += 64) {                  // coalesced: consecutive lanes, consecutive words
        const int r = e / d;
        X[r * LD + (e - r * d)] = src[e];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const bool vel_fd = flags & MPB_TERM_VEL_FD;          // velocities = central differences of positions
    const double ddt = (double)dt;
    const double q11 = 12.0 / (ddt * ddt * ddt), q12 = -6.0 / (ddt * ddt), q22 = 4.0 / ddt;
    const double inv_dt4 = 1.0 / (ddt * ddt * ddt * ddt), inv_2dt = 1.0 / (2.0 * ddt);
    double acc = 0.0, jl = 0.0;
    for (int h = lane; h < H; h += 64) {
        const float* xh = X + h * LD;
        // ---- GP prior factor h -> h+1 (gp_factor.py:52-56 error, :42-50 Q^-1)
        if ((flags & MPB_TERM_GP) && h + 1 < H) {
            const float* xn = xh + LD;
            double s = 0.0;
            for (int i = 0; i < D; ++i) {
                double v0, v1;
                if (vel_fd) {
                    v0 = (h >= 1) ? ((double)xn[i] - (double)xh[i - LD]) * inv_2dt : 0.0;
                    v1 = (h + 2 < H) ? ((double)xn[i + LD] - (double)xh[i]) * inv_2dt : 0.0;
                } else {
                    v0 = (double)xh[D + i];
                    v1 = (double)xn[D + i];
                }
                const double ep = (double)xn[i] - ((double)xh[i] + ddt * v0), ev = v1 - v0;
                s += q11 * ep * ep + 2.0 * q12 * ep * ev + q22 * ev * ev;
            }
            acc += (double)k_gp * s;
        }
        // ---- start prior on x_0 (unary_factor.py:18-32), all 2D state dims
        if ((flags & MPB_TERM_START) && h == 0) {
            double s = 0.0;
            for (int i = 0; i < 2 * D; ++i) {
                const double e = (double)start_state[i] - (double)xh[i];
                s += e * e;
            }
            acc += (double)k_start * s;
        }
        // ---- goal prior on x_{H-1}: trajectory b belongs to goal b / trajs_per_goal (cost_functions.py:525-533)
        if ((flags & MPB_TERM_GOAL) && h == H - 1) {
            const float* g = goal_states + (size_t)(b / trajs_per_goal) * 2 * D;
            double s = 0.0;
            for (int i = 0; i < 2 * D; ++i) {
                const double e = (double)g[i] - (double)xh[i];
                s += e * e;
            }
            acc += (double)k_goal * s;
        }
        // ---- CHOMP smoothness x^T R x with R = K^T K (chomp.py:81-101) in factored form: rows of K x are
        //      x_0, x_h - x_{h-1}, -x_{H-1}; every state column of the trajectory takes part
        if (flags & MPB_TERM_SMOOTH) {
            double s = 0.0;
            for (int i = 0; i < d; ++i) {
                const double x = (double)xh[i];
                const double df = (h >= 1) ? x - (double)xh[i - LD] : x;
                s += df * df;
                if (h == H - 1) s += x * x;
            }
            acc += (double)k_smooth * inv_dt4 * s;
        }
        // ---- joint limits (cost_functions.py:406-426)
        if (flags & MPB_TERM_JLIM) {
            for (int i = 0; i < D; ++i) {
                const double q = (double)xh[i];
                const double lo = (double)q_min[i] + (double)jl_eps - q, hi = q - ((double)q_max[i] - (double)jl_eps);
                if (lo > 0.0) jl += lo * lo;
                if (hi > 0.0) jl += hi * hi;
            }
        }
    }
    acc = wave_sum_f64(acc);
    if (flags & MPB_TERM_JLIM) {
        jl = wave_sum_f64(jl);
        if (lane == 0 && jl != 0.0) atomicAdd(jl_total, (double)k_jlim * jl);
    }
    if (lane == 0) out[b] = (accumulate ? out[b] : 0.f) + (float)acc;
}

// the reference's joint-limit cost is one scalar for the whole batch that the composite broadcasts
__global__ void cost_add_scalar_kernel(float* __restrict__ out, const double* __restrict__ v, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] += (float)v[0];
}

extern "C" int mpb_cost_terms_eval(const float* trajs, float* out, double* jl_total, const float* start_state,
                                   const float* goal_states, const float* q_min, const float* q_max, int B, int H,
                                   int d, int n_dof, int trajs_per_goal, uint32_t flags, float dt, float k_gp,
                                   float k_start, float k_goal, float k_smooth, float k_jlim, float jl_eps,
                                   int accumulate, int broadcast_jlim, void* stream) {
    if (B < 0 || H < 2 || H > MPB_MAX_H || n_dof < 1 || d < 1)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: bad shape");
    if (flags & ~(uint32_t)MPB_TERM_ALL) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: unknown term flag");
    const bool fd = flags & MPB_TERM_VEL_FD;
    if ((flags & (MPB_TERM_GP | MPB_TERM_JLIM)) && d < n_dof)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: d < n_dof");
    if ((flags & MPB_TERM_GP) && !fd && d != 2 * n_dof)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: the GP term needs d == 2*n_dof (or MPB_TERM_VEL_FD with d == n_dof)");
    if (fd && d != n_dof) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: MPB_TERM_VEL_FD needs d == n_dof");
    if ((flags & (MPB_TERM_START | MPB_TERM_GOAL)) && d != 2 * n_dof)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: start / goal priors need d == 2*n_dof");
    if ((flags & MPB_TERM_GP) && !(dt > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: dt must be > 0");
    if ((flags & MPB_TERM_SMOOTH) && !(dt > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: dt must be > 0");
    if (B == 0) return MPB_OK;
    if (!trajs || !out) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: null pointer");
    if ((flags & MPB_TERM_START) && !start_state) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: start_state is NULL");
    if ((flags & MPB_TERM_GOAL) && (!goal_states || trajs_per_goal < 1))
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: goal_states NULL or trajs_per_goal < 1");
    if ((flags & MPB_TERM_JLIM) && (!q_min || !q_max || !jl_total))
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_eval: joint-limit term needs q_min, q_max, jl_total");
    const size_t lds = (size_t)COSTS_WAVES * H * (d | 1) * sizeof(float);
    if (lds > 150 * 1024) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_cost_terms_eval: H*d too large for the LDS tile");
    hipStream_t st = (hipStream_t)stream;
    if (flags & MPB_TERM_JLIM) {
        if (hipMemsetAsync(jl_total, 0, sizeof(double), st) != hipSuccess)
            return mpb_fail(MPB_E_HIP, "mpb_cost_terms_eval: hipMemsetAsync failed");
    }
    hipLaunchKernelGGL(cost_terms_kernel, dim3((B + COSTS_WAVES - 1) / COSTS_WAVES), dim3(64 * COSTS_WAVES), lds, st,
                       trajs, out, jl_total, start_state, goal_states, q_min, q_max, B, H, d, n_dof, trajs_per_goal,
                       flags, dt, k_gp, k_start, k_goal, k_smooth, k_jlim, jl_eps, accumulate);
    if ((flags & MPB_TERM_JLIM) && broadcast_jlim)
        hipLaunchKernelGGL(cost_add_scalar_kernel, dim3((B + 255) / 256), dim3(256), 0, st, out, jl_total, B);
    return mpb_check_launch("mpb_cost_terms_eval");
}

// ------------------------------------------------------------------------------------------------
// Analytic gradient of the same terms (what the reference obtains by autograd through CostComposite.eval:
// chomp.py:135-139) and, optionally, the CHOMP update in the same pass (chomp.py:141-147).
//   g[b,h,:] = grad_in[b,h,:]                                          (e.g. the collision gradient; may be NULL)
//            + d/dx [ enabled terms of cost_terms_kernel ]              (joint limits times jl_scale: the batch-global
//                                                                        scalar is added to EVERY trajectory's cost, so
//                                                                        the gradient of costs.sum() carries the batch size)
//            + prior_bw * (R + R^T) x                                   (CHOMP's smoothness prior incl. quirk Q3's batch
//                                                                        factor; R's tridiagonal band is read from Rm)
//   apply == 0: grad_out = g;   apply != 0: x -= lr * mask(clamp(g, -clip, clip)), mask zeroes rows 0 and H-1.
// One wave per trajectory, lane = waypoint; the trajectory is staged once in LDS (every term's stencil reads its
// neighbours' rows from there, and the update writes global memory only, so no wave races with another).
// MPB_TERM_VEL_FD (velocities = central differences of the positions): the gradient with respect to the virtual
// velocities goes through a second LDS tile and is pulled back onto the positions (v_h = (x_{h+1} - x_{h-1}) / 2dt).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * COSTS_WAVES) void cost_terms_grad_kernel(
    float* __restrict__ trajs, const float* __restrict__ grad_in, float* __restrict__ grad_out,
    const float* __restrict__ Rm, const float* __restrict__ start_state, const float* __restrict__ goal_states,
    const float* __restrict__ q_min, const float* __restrict__ q_max, int B, int H, int d, int D, int trajs_per_goal,
    uint32_t flags, float dt, float k_gp, float k_start, float k_goal, float k_smooth, float k_jlim, float jl_eps,
    float jl_scale, float prior_bw, float lr, float grad_clip, int apply) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * COSTS_WAVES + wave;
    if (b >= B) return;                                   // whole wave leaves together; no block barriers below
    const int LD = d | 1;
    float* X = lds + (size_t)wave * 2 * H * LD;           // trajectory tile
    float* V = X + (size_t)H * LD;                        // VEL_FD: gradient w.r.t. the virtual velocities (H x D)
    float* dst = trajs + (size_t)b * H * d;
    const int n = H * d;
    for (int e = lane; e < n; e += 64) {
        const int r = e / d;
        X[r * LD + (e - r * d)] = dst[e];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool vel_fd = flags & MPB_TERM_VEL_FD;
    const double ddt = (double)dt;
    const double q11 = 12.0 / (ddt * ddt * ddt), q12 = -6.0 / (ddt * ddt), q22 = 4.0 / ddt;
    const double inv_dt4 = 1.0 / (ddt * ddt * ddt * ddt), inv_2dt = 1.0 / (2.0 * ddt);
    // virtual velocity of row h, dof i (VEL_FD) or the stored one
    auto vel = [&](int h, int i) -> double {
        if (!vel_fd) return (double)X[h * LD + D + i];
        return (h >= 1 && h + 1 < H) ? ((double)X[(h + 1) * LD + i] - (double)X[(h - 1) * LD + i]) * inv_2dt : 0.0;
    };
    // ---- pass 1 (VEL_FD only): d cost_gp / d v_h into V
    if ((flags & MPB_TERM_GP) && vel_fd) {
        for (int h = lane; h < H; h += 64) {
            for (int i = 0; i < D; ++i) {
                double gv = 0.0;
                if (h + 1 < H) {
                    const double ep = (double)X[(h + 1) * LD + i] - ((double)X[h * LD + i] + ddt * vel(h, i));
                    const double ev = vel(h + 1, i) - vel(h, i);
                    const double a = 2.0 * (q11 * ep + q12 * ev), bb = 2.0 * (q12 * ep + q22 * ev);
                    gv -= ddt * a + bb;
                }
                if (h >= 1) {
                    const double ep = (double)X[h * LD + i] - ((double)X[(h - 1) * LD + i] + ddt * vel(h - 1, i));
                    const double ev = vel(h, i) - vel(h - 1, i);
                    gv += 2.0 * (q12 * ep + q22 * ev);
                }
                V[h * D + i] = (float)((double)k_gp * gv);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int h = lane; h < H; h += 64) {
        const float* xh = X + h * LD;
        for (int c = 0; c < d; ++c) {
            double g = grad_in ? (double)grad_in[((size_t)b * H + h) * d + c] : 0.0;
            const bool is_pos = c < D;
            const int i = is_pos ? c : c - D;
            // ---- GP prior factors (h-1, h) and (h, h+1)
            if ((flags & MPB_TERM_GP) && i < D && (is_pos || (!vel_fd && c < 2 * D))) {
                double gg = 0.0;
                if (h + 1 < H) {
                    const double ep = (double)X[(h + 1) * LD + i] - ((double)xh[i] + ddt * vel(h, i));
                    const double ev = vel(h + 1, i) - vel(h, i);
                    const double a = 2.0 * (q11 * ep + q12 * ev), bb = 2.0 * (q12 * ep + q22 * ev);
                    gg += is_pos ? -a : -(ddt * a + bb);
                }
                if (h >= 1) {
                    const double ep = (double)xh[i] - ((double)X[(h - 1) * LD + i] + ddt * vel(h - 1, i));
                    const double ev = vel(h, i) - vel(h - 1, i);
                    const double a = 2.0 * (q11 * ep + q12 * ev), bb = 2.0 * (q12 * ep + q22 * ev);
                    gg += is_pos ? a : bb;
                }
                g += (double)k_gp * gg;
                if (vel_fd) {   // pull the virtual-velocity gradient back: v_{h-1} and v_{h+1} depend on x_h
                    if (h - 1 >= 1 && h < H) g += (double)V[(h - 1) * D + i] * inv_2dt;          // v_{h-1} = (x_h - x_{h-2}) / 2dt, h-1 interior
                    if (h + 1 + 1 < H && h + 1 >= 1) g -= (double)V[(h + 1) * D + i] * inv_2dt;  // v_{h+1} = (x_{h+2} - x_h) / 2dt, h+1 interior
                }
            }
            if ((flags & MPB_TERM_START) && h == 0 && c < 2 * D) g -= 2.0 * (double)k_start * ((double)start_state[c] - (double)xh[c]);
            if ((flags & MPB_TERM_GOAL) && h == H - 1 && c < 2 * D)
                g -= 2.0 * (double)k_goal * ((double)goal_states[(size_t)(b / trajs_per_goal) * 2 * D + c] - (double)xh[c]);
            if (flags & MPB_TERM_SMOOTH) {
                const double x = (double)xh[c];
                const double df = (h >= 1) ? x - (double)xh[c - LD] : x;
                double gs = 2.0 * df;
                if (h + 1 < H) gs -= 2.0 * ((double)xh[c + LD] - x);
                if (h == H - 1) gs += 2.0 * x;
                g += (double)k_smooth * inv_dt4 * gs;
            }
            if ((flags & MPB_TERM_JLIM) && c < D) {
                const double q = (double)xh[c];
                const double lo = (double)q_min[c] + (double)jl_eps - q, hi = q - ((double)q_max[c] - (double)jl_eps);
                double gj = 0.0;
                if (lo > 0.0) gj -= 2.0 * lo;
                if (hi > 0.0) gj += 2.0 * hi;
                g += (double)jl_scale * (double)k_jlim * gj;
            }
            if (prior_bw != 0.f) {   // CHOMP prior: (B w) (R x + R^T x), R tridiagonal (chomp.py:81-101, :165)
                const float xm = (h >= 1) ? xh[c - LD] : 0.f, xp = (h + 1 < H) ? xh[c + LD] : 0.f;
                const float r_lo = (h >= 1) ? Rm[h * H + h - 1] : 0.f, r_di = Rm[h * H + h], r_up = (h + 1 < H) ? Rm[h * H + h + 1] : 0.f;
                const float rx = fmaf(r_up, xp, fmaf(r_di, xh[c], r_lo * xm));   // same association as the CHOMP kernels (mpb_chomp.hip)
                g += (double)(prior_bw * (rx + rx));
            }
            if (apply) {
                float gf = fminf(fmaxf((float)g, -grad_clip), grad_clip);
                if (h == 0 || h == H - 1) gf = 0.f;
                dst[(size_t)h * d + c] = xh[c] + (-lr * gf);
            } else {
                grad_out[((size_t)b * H + h) * d + c] = (float)g;
            }
        }
    }
}

extern "C" int mpb_cost_terms_grad(float* trajs, const float* grad_in, float* grad_out, const float* R,
                                   const float* start_state, const float* goal_states, const float* q_min,
                                   const float* q_max, int B, int H, int d, int n_dof, int trajs_per_goal, uint32_t flags,
                                   float dt, float k_gp, float k_start, float k_goal, float k_smooth, float k_jlim,
                                   float jl_eps, float jl_scale, float prior_bw, float lr, float grad_clip, int apply,
                                   void* stream) {
    if (B < 0 || H < 2 || H > MPB_MAX_H || n_dof < 1 || d < 1) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: bad shape");
    if (flags & ~(uint32_t)MPB_TERM_ALL) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: unknown term flag");
    const bool fd = flags & MPB_TERM_VEL_FD;
    if ((flags & (MPB_TERM_GP | MPB_TERM_JLIM)) && d < n_dof) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: d < n_dof");
    if ((flags & MPB_TERM_GP) && !fd && d != 2 * n_dof)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: the GP term needs d == 2*n_dof (or MPB_TERM_VEL_FD with d == n_dof)");
    if (fd && d != n_dof) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: MPB_TERM_VEL_FD needs d == n_dof");
    if ((flags & (MPB_TERM_START | MPB_TERM_GOAL)) && d != 2 * n_dof)
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: start / goal priors need d == 2*n_dof");
    if ((flags & (MPB_TERM_GP | MPB_TERM_SMOOTH)) && !(dt > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: dt must be > 0");
    if (B == 0) return MPB_OK;
    if (!trajs || (!apply && !grad_out)) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: null pointer");
    if ((flags & MPB_TERM_START) && !start_state) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: start_state is NULL");
    if ((flags & MPB_TERM_GOAL) && (!goal_states || trajs_per_goal < 1))
        return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: goal_states NULL or trajs_per_goal < 1");
    if ((flags & MPB_TERM_JLIM) && (!q_min || !q_max)) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: joint-limit term needs q_min, q_max");
    if (prior_bw != 0.f && !R) return mpb_fail(MPB_E_INVALID, "mpb_cost_terms_grad: the CHOMP prior needs R");
    const size_t lds = (size_t)COSTS_WAVES * 2 * H * (d | 1) * sizeof(float);
    if (lds > 150 * 1024) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_cost_terms_grad: H*d too large for the LDS tiles");
    hipLaunchKernelGGL(cost_terms_grad_kernel, dim3((B + COSTS_WAVES - 1) / COSTS_WAVES), dim3(64 * COSTS_WAVES), lds,
                       (hipStream_t)stream, trajs, grad_in, grad_out, R, start_state, goal_states, q_min, q_max, B, H, d,
                       n_dof, trajs_per_goal, flags, dt, k_gp, k_start, k_goal, k_smooth, k_jlim, jl_eps, jl_scale, prior_bw,
                       lr, grad_clip, apply);
    return mpb_check_launch("mpb_cost_terms_grad");
}

// ------------------------------------------------------------------------------------------------
// trajectory utilities: streaming element-wise kernels, one thread per output word (coalesced stores)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void traj_interpolate_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               size_t total, int H, int Ho, int d, int n1) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / d;
        const int i = (int)(e - row * d);
        const size_t b = row / Ho;
        const int ho = (int)(row - b * Ho);
        const int seg = ho / n1, k = ho - seg * n1;              // ho = seg*(n+1) + k; the last row is seg = H-1, k = 0
        const float* p = x + ((size_t)b * H + seg) * d + i;
        const float x0 = p[0];
        float v = x0;
        if (k > 0) v = x0 + ((float)k / (float)n1) * (p[d] - x0);
        out[e] = v;
    }
}

__global__ __launch_bounds__(256) void traj_fd_kernel(const float* __restrict__ pos, float* __restrict__ out, size_t total,
                                                      int H, int D, float inv_2dt) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / (2 * D);
        const int i = (int)(e - row * 2 * D);
        const int t = (int)(row % H);
        const float* p = pos + row * D;
        float v;
        if (i < D) {
            v = p[i];
        } else {
            const int j = i - D;
            v = (t >= 1 && t + 1 < H) ? (p[D + j] - p[j - D]) * inv_2dt : 0.f;
        }
        out[e] = v;
    }
}

static inline int stream_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : g);
}

extern "C" int mpb_traj_interpolate(const float* trajs, float* out, int B, int H, int d, int n_interp, void* stream) {
    if (B < 0 || H < 2 || d < 1 || n_interp < 0) return mpb_fail(MPB_E_INVALID, "mpb_traj_interpolate: bad shape");
    if (B == 0) return MPB_OK;
    if (!trajs || !out) return mpb_fail(MPB_E_INVALID, "mpb_traj_interpolate: null pointer");
    const int n1 = n_interp + 1, Ho = (H - 1) * n1 + 1;
    const size_t total = (size_t)B * Ho * d;
    hipLaunchKernelGGL(traj_interpolate_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, trajs, out, total,
                       H, Ho, d, n1);
    return mpb_check_launch("mpb_traj_interpolate");
}

extern "C" int mpb_traj_finite_difference(const float* pos, float* out, int B, int H, int D, float dt, void* stream) {
    if (B < 0 || H < 2 || D < 1 || !(dt > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_traj_finite_difference: bad shape or dt");
    if (B == 0) return MPB_OK;
    if (!pos || !out) return mpb_fail(MPB_E_INVALID, "mpb_traj_finite_difference: null pointer");
    const size_t total = (size_t)B * H * 2 * D;
    hipLaunchKernelGGL(traj_fd_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, pos, out, total, H, D,
                       1.f / (2.f * dt));
    return mpb_check_launch("mpb_traj_finite_difference");
}

// ------------------------------------------------------------------------------------------------
// warm start of HybridPlanner (hybrid_planner.py:42-66): every sample-based path (a polyline with its own
// number of waypoints) becomes H support points + velocities.  One wave per path: segment lengths ->
// wave-wide inclusive scan into LDS (cumulative arc length) -> every output waypoint bisects its arc-length
// target and interpolates linearly.  Ragged input: paths padded to Lmax rows, lengths[n] valid rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void traj_resample_kernel(const float* __restrict__ paths, const int* __restrict__ lengths,
                                                           float* __restrict__ out, int Lmax, int H, int D, float dt) {
    extern __shared__ double cum[];                        // cum[k] = arc length from waypoint 0 to waypoint k (fp64:
    const int n = blockIdx.x, lane = threadIdx.x;          // hundreds of segments are summed)
    const int L = min(max(lengths[n], 1), Lmax);
    const float* P = paths + (size_t)n * Lmax * D;
    double carry = 0.0;
    for (int base = 0; base < L; base += 64) {
        const int k = base + lane;
        double seg = 0.0;
        if (k >= 1 && k < L) {
            double s2 = 0.0;
            for (int i = 0; i < D; ++i) {
                const double df = (double)P[(size_t)k * D + i] - (double)P[(size_t)(k - 1) * D + i];
                s2 = fma(df, df, s2);
            }
            seg = sqrt(s2);
        }
        for (int off = 1; off < 64; off <<= 1) {           // inclusive scan over the wave
            const double up = __shfl_up(seg, off, 64);
            if (lane >= off) seg += up;
        }
        if (k < L) cum[k] = carry + seg;
        carry += __shfl(seg, 63, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double total = cum[L - 1];
    const float* p0 = P;
    const float* p1 = P + (size_t)(L - 1) * D;
    for (int h = lane; h < H; h += 64) {
        float* o = out + ((size_t)n * H + h) * 2 * D;
        const double target = total * ((double)h / (double)(H - 1));
        // largest k with cum[k] <= target (cum is non-decreasing, cum[0] = 0)
        int lo = 0, hi = L - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cum[mid] <= target) lo = mid; else hi = mid - 1;
        }
        const int k0 = lo, k1 = min(lo + 1, L - 1);
        const double span = cum[k1] - cum[k0];
        const double a = span > 0.0 ? (target - cum[k0]) / span : 0.0;
        const bool interior = h >= 1 && h + 1 < H;
        for (int i = 0; i < D; ++i) {
            const double x0 = P[(size_t)k0 * D + i], x1 = P[(size_t)k1 * D + i];
            o[i] = (h == 0) ? p0[i] : (h == H - 1) ? p1[i] : (float)(x0 + a * (x1 - x0));
            // average velocity of the whole motion on the interior points, rest at both ends
            o[D + i] = interior ? (float)(((double)p1[i] - (double)p0[i]) / ((double)(H - 1) * (double)dt)) : 0.f;
        }
    }
}

extern "C" int mpb_traj_resample(const float* paths, const int* lengths, float* out, int N, int Lmax, int H, int D, float dt,
                                 void* stream) {
    if (N < 0 || Lmax < 1 || H < 2 || D < 1 || !(dt > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_traj_resample: bad shape or dt");
    if ((size_t)Lmax * sizeof(double) > 150 * 1024) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_traj_resample: Lmax too large for LDS");
    if (N == 0) return MPB_OK;
    if (!paths || !lengths || !out) return mpb_fail(MPB_E_INVALID, "mpb_traj_resample: null pointer");
    hipLaunchKernelGGL(traj_resample_kernel, dim3(N), dim3(64), (size_t)Lmax * sizeof(double), (hipStream_t)stream, paths, lengths,
                       out, Lmax, H, D, dt);
    return mpb_check_launch("mpb_traj_resample");
}

// ------------------------------------------------------------------------------------------------
// GP factor error (gp_factor.py:52-56): err[b,t] = x[b,t+1] - Phi x[b,t], Phi = [[I, dt I],[0, I]]; (B,H,2D) -> (B,H-1,2D)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_factor_error_kernel(const float* __restrict__ x, float* __restrict__ out, size_t total,
                                                              int H, int D, float dt) {
    const int dim = 2 * D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / dim;                       // b * (H-1) + t
        const int c = (int)(e - row * dim);
        const size_t b = row / (H - 1);
        const int t = (int)(row - b * (H - 1));
        const float* p = x + ((size_t)b * H + t) * dim;
        out[e] = (c < D) ? p[dim + c] - (p[c] + dt * p[D + c]) : p[dim + c] - p[c];
    }
}

extern "C" int mpb_gp_factor_error(const float* x, float* out, int B, int H, int D, float dt, void* stream) {
    if (B < 0 || H < 2 || D < 1) return mpb_fail(MPB_E_INVALID, "mpb_gp_factor_error: bad shape");
    if (B == 0) return MPB_OK;
    if (!x || !out) return mpb_fail(MPB_E_INVALID, "mpb_gp_factor_error: null pointer");
    const size_t total = (size_t)B * (H - 1) * 2 * D;
    hipLaunchKernelGGL(gp_factor_error_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, out, total, H, D, dt);
    return mpb_check_launch("mpb_gp_factor_error");
}
