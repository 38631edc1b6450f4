// mpb_gpmp2_lr.hip -- the GPMP2 Gauss-Newton step (gpmp2.py:308-368, :451-452) in LOW-RANK form (round 6).
//
// The normal equations of the reference (cost_functions.py:107-144: rows of the start prior, the GP factors, the goal prior
// and one collision row per waypoint; gpmp2.py:355-368: J^T J = A^T K A + damping) split as
//
//     J^T J = A0 + V C V^T,      g = g_rest + V C c
//
// * A0 = start / goal priors + GP blocks + damping.  It is THE SAME FOR EVERY PARTICLE: the sigmas are the planner's, and the
//   trust-region damping is delta * mean_b diag(A^T K A) -- a batch mean (quirk Q9).  And every one of its blocks is
//   (2 x 2) (x) I_D or diagonal: A0 decouples over the degrees of freedom into D chains of H 2 x 2 blocks (position,
//   velocity of one joint along the trajectory).
// * V C V^T = the collision factors: column a of V is h_t = -d c_t / d q_t (field_factor.py:54) embedded at the position
//   rows of waypoint t, C = I / sigma_coll^2; only ACTIVE rows (h_t != 0: waypoints inside the hinge margin, ~10 % of them at
//   C4) take part.
//
// Woodbury, with the right-hand side kept apart so that `kc c h` never meets A0^-1 on its own (the cancellation the
// Sherman-Morrison form of round 5 avoids waypoint by waypoint -- here for all of them at once):
//
//     u0 = A0^-1 g_rest,     M = C^-1 + V^T A0^-1 V,     M w = c - V^T u0,     dtheta = A0^-1 (g_rest + V w)
//
// scripts/gpmp2_lowrank_prototype.py (numpy fp64 against a long-double-refined dense solve): as accurate as dense fp64 Cholesky
// at every collision / GP precision ratio from 1e6 to 1e14, active sets from 8 to 84 rows.
//
// Four launches per iteration:
//   gpmp2_chain_kernel    shared by all particles: the block-Thomas factors (W_t = S_t^-1, F_t = W_t U) of the D chains and the
//                         position-position entries G_i(s, t) of their inverses, D H^2 doubles (0.9 MB at C4: stays in L2);
//   gpmp2_lr_sweep<false> u0: lane = (particle, joint) -- 9 particles per wave at D = 7 --, the factors in LDS, the joint's own
//                         gradient formed on the fly from x (the GP factors couple a joint only with itself), H steps down and H
//                         up with two dependent fma per step; x and the z_t records travel through register rings 16 steps
//                         ahead (the first version of the round walked the chain inside the per-particle kernel with the factors
//                         read from L2 one step ahead: 512 L2 round trips in a row per particle, 0.91 ms at C4);
//   gpmp2_lr_cap          per particle, one wave: the active rows compacted by ballot, M (n_a x n_a) from the G table in 8 x 8 lane
//                         tiles into LDS (packed lower triangle, column major), Cholesky and both triangular solves in LDS, w
//                         scattered to a dense per-waypoint vector;
//   gpmp2_lr_sweep<true>  dtheta = A0^-1 (g_rest + V w) and x += step * dtheta.
// Against the block elimination of rounds 1-5 (mpb_gpmp2.hip: one 16 x 16 fp64 Gauss-Jordan inverse per waypoint and particle, 127
// of them in a row; 352 MB of W_t records written and read back): ~1.2 MFLOP and ~2 600 dependent pivot steps per particle become
// ~30 kFLOP and 4 H + 2 n_a short steps.  The block kernel stays for what this form does not take: chained fields whose rows
// exceed the LDS tile (F (H - 1) > 127), H > 128.
#include <stdlib.h>
#include <type_traits>

#include "mpb_common.h"
#include "mpb_gpmp2.h"

#define LR_REC 8                       // doubles per (joint, waypoint) record: W00 W01 W11 F00 F01 F10 F11 (pad)
#define LR_NMAX 127                    // active rows a particle may have (with the right-hand side 128 rows: 36 tiles of 16 x 16 doubles, 72 KB of LDS)
#define LR_PF 16                       // steps the sweeps read ahead (register rings)
#define LR_ZREC 3                      // doubles per (waypoint, lane) record of a sweep: z0, z1, (x_pos, x_vel as two floats)

typedef double lr_d2 __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void lr_static_for(F&& f) {          // f(integral_constant<int, I>) for I in [I, N): indices stay compile-time constants
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lr_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ double lr_rcp(double x) {      // v_rcp_f64 + two Newton steps: <= 1.0 x 2^-53 (profiles/r03_rcp_accuracy.txt)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double lr_readlane(double v, int l) {      // l: wave-uniform
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lr_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

struct LrCoef {          // the 2 x 2 coefficient blocks of one joint's chain (Kronecker factors of the reference's blocks)
    double a, bq, cq;            // Qi           = [[a, bq], [bq, cq]]            (gp_factor.py:42-50)
    double p00, p01, p11;        // Phi^T Qi Phi
    double u00, u01, u10, u11;   // U = -Phi^T Qi: block (t, t + 1) of J^T J
};
__device__ __forceinline__ LrCoef lr_coef(const GpConst& K) {
    const double dt = K.dt;
    LrCoef c;
    c.a = 12.0 / (dt * dt * dt) * K.kgp; c.bq = -6.0 / (dt * dt) * K.kgp; c.cq = 4.0 / dt * K.kgp;
    c.p00 = c.a; c.p01 = 6.0 / (dt * dt) * K.kgp; c.p11 = c.cq;
    c.u00 = -c.a; c.u01 = -c.bq; c.u10 = -(c.a * dt + c.bq); c.u11 = -(c.bq * dt + c.cq);
    return c;
}

// ------------------------------------------------------------------------------------------------
// shared: factors of the D chains and the position-position entries of their inverses.
// grid = (D, ceil(H / LR_COLS)), block = LR_COLS threads.  Every block factorises its joint's chain itself (H sequential 2 x 2
// steps on one thread, operands in LDS: cheaper than a launch of its own) and runs the recurrence of the inverse's diagonal blocks,
//     G_{H-1,H-1} = W_{H-1},      G_{t,t} = W_t + F_t G_{t+1,t+1} F_t^T,
// then thread c walks column c of the inverse UP from its diagonal block, G_{t,c} = -F_t G_{t+1,c} (t < c: above the diagonal the
// forward pass of a unit vector is zero, so the back substitution is this product alone), and writes the position-position entry to
// (t, c) and, by symmetry, to (c, t): c short steps instead of the 2 H of a substitution per column.
// ------------------------------------------------------------------------------------------------
#define LR_COLS 32
__global__ __launch_bounds__(LR_COLS) void gpmp2_chain_kernel(const double* __restrict__ diag_mean, double* __restrict__ rec_g,
                                                              double* __restrict__ G, int H, int D, GpConst K) {
    extern __shared__ double lds[];
    double* rec = lds;                                 // H x LR_REC
    double* gd = lds + (size_t)H * LR_REC;             // H x 2: first the damping (position, velocity), then column 0 of G_{t,t}
    const int i = blockIdx.x, c0 = blockIdx.y * LR_COLS, tid = threadIdx.x;
    const LrCoef C = lr_coef(K);
    const int dim = 2 * D;
    // the damping of this joint's position / velocity rows, staged (the sequential loop below must not wait for global memory)
    for (int t = tid; t < H; t += LR_COLS) {
        gd[2 * t] = K.trust ? K.delta * diag_mean[(size_t)t * dim + i] : K.delta;
        gd[2 * t + 1] = K.trust ? K.delta * diag_mean[(size_t)t * dim + D + i] : K.delta;
    }
    __syncthreads();
    if (tid == 0) {
        // block Thomas on the 2 x 2 chain of joint i:  S_0 = D_0,  W_t = S_t^-1,  F_t = W_t U,  S_{t+1} = D_{t+1} - U^T W_t U
        double w00 = 0.0, w01 = 0.0, w11 = 0.0;
#pragma unroll 4
        for (int t = 0; t < H; ++t) {          // (unrolled: the damping reads of later steps issue ahead of the dependent chain)
            const double first = (t == 0) ? 1.0 : 0.0, last = (t == H - 1) ? 1.0 : 0.0;
            const double dp = gd[2 * t], dv = gd[2 * t + 1];
            double s00 = (1.0 - last) * C.p00 + (1.0 - first) * C.a + dp + first * K.ks + last * K.kg;
            double s01 = (1.0 - last) * C.p01 + (1.0 - first) * C.bq;
            double s11 = (1.0 - last) * C.p11 + (1.0 - first) * C.cq + dv + first * K.ks + last * K.kg;
            if (t > 0) {
                // U^T W U with W symmetric: X = W U, then U^T X
                const double x00 = w00 * C.u00 + w01 * C.u10, x01 = w00 * C.u01 + w01 * C.u11;
                const double x10 = w01 * C.u00 + w11 * C.u10, x11 = w01 * C.u01 + w11 * C.u11;
                s00 -= C.u00 * x00 + C.u10 * x10;
                s01 -= C.u00 * x01 + C.u10 * x11;
                s11 -= C.u01 * x01 + C.u11 * x11;
            }
            const double id = lr_rcp(fma(s00, s11, -s01 * s01));
            w00 = s11 * id; w01 = -s01 * id; w11 = s00 * id;
            double* r = rec + (size_t)t * LR_REC;
            r[0] = w00; r[1] = w01; r[2] = w11;
            r[3] = w00 * C.u00 + w01 * C.u10; r[4] = w00 * C.u01 + w01 * C.u11;          // F = W U
            r[5] = w01 * C.u00 + w11 * C.u10; r[6] = w01 * C.u01 + w11 * C.u11;
            r[7] = 0.0;
        }
        // diagonal blocks of the inverse, bottom up; column 0 (the response to a unit POSITION entry) of each is kept
        double g00 = w00, g01 = w01, g11 = w11;
        gd[2 * (H - 1)] = g00; gd[2 * (H - 1) + 1] = g01;
#pragma unroll 4
        for (int t = H - 2; t >= 0; --t) {
            const double* r = rec + (size_t)t * LR_REC;
            const double f00 = r[3], f01 = r[4], f10 = r[5], f11 = r[6];
            // Y = F G (G symmetric), then Y F^T
            const double y00 = f00 * g00 + f01 * g01, y01 = f00 * g01 + f01 * g11;
            const double y10 = f10 * g00 + f11 * g01, y11 = f10 * g01 + f11 * g11;
            g00 = r[0] + (y00 * f00 + y01 * f01);
            g01 = r[1] + (y00 * f10 + y01 * f11);
            g11 = r[2] + (y10 * f10 + y11 * f11);
            gd[2 * t] = g00; gd[2 * t + 1] = g01;
        }
    }
    __syncthreads();
    if (blockIdx.y == 0)
        for (int e = tid; e < H * LR_REC; e += LR_COLS) rec_g[(size_t)i * H * LR_REC + e] = rec[e];
    const int c = c0 + tid;
    if (c >= H) return;
    double* Gi = G + (size_t)i * H * H;
    double y0 = gd[2 * c], y1 = gd[2 * c + 1];
    Gi[(size_t)c * H + c] = y0;
#pragma unroll 4
    for (int t = c - 1; t >= 0; --t) {
        const double* r = rec + (size_t)t * LR_REC;
        const double n0 = -(r[3] * y0 + r[4] * y1), n1 = -(r[5] * y0 + r[6] * y1);
        y0 = n0; y1 = n1;
        Gi[(size_t)t * H + c] = y0;
        Gi[(size_t)c * H + t] = y0;
    }
}

// ------------------------------------------------------------------------------------------------
// A0^-1 applied to a gradient: lane = (particle, joint).  FINAL = false: the gradient is g_rest (priors + GP factors,
// gpmp2.py:355-368 with the rows of cost_functions.py:291-314, :538-554); writes the position rows of u0 = A0^-1 g_rest (t major:
// upos[t][lane]) and the cost b^T K b of those factors per particle (gpmp2.py:493-495).  FINAL = true: the gradient is
// g_rest + V w (w: dense per field and waypoint, zero off the active rows); writes x += step * dtheta (gpmp2.py:326-331).
// Forward r_t = g_t - F_{t-1}^T r_{t-1}, z_t = W_t r_t (records to zbuf, t major, with the lane's x_t beside them); backward
// y_t = z_t - F_t y_{t+1}.
// ------------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ __launch_bounds__(64) void gpmp2_lr_sweep(float* __restrict__ x, const float* __restrict__ start, const float* __restrict__ goal,
                                                     const float* __restrict__ jac, const double* __restrict__ wdense,
                                                     const double* __restrict__ rec_g, double* __restrict__ zbuf, double* __restrict__ upos,
                                                     double* __restrict__ gpcost, int B, int H, int D, int F, GpConst K) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, dim = 2 * D;
    const int per = 64 / D;                                   // particles per wave
    const int stride = H * LR_REC + 2;                        // doubles between two joints' tables (+ 2: their records fall on different banks)
    for (int e = lane; e < D * H * (LR_REC / 2); e += 64) {   // the shared factors into LDS, 16 bytes at a time
        const int j = e / (H * (LR_REC / 2)), r = e - j * (H * (LR_REC / 2));
        reinterpret_cast<lr_d2*>(lds + (size_t)j * stride)[r] = reinterpret_cast<const lr_d2*>(rec_g + (size_t)j * H * LR_REC)[r];
    }
    __syncthreads();
    const int pl = lane / D, i = lane - pl * D;
    const int p = blockIdx.x * per + pl;
    const bool live = pl < per && p < B;
    const int pc = live ? p : 0;                              // idle lanes shadow particle 0 (loads only; nothing is stored)
    const size_t NL = (size_t)B * D;                          // lanes of the whole batch: the t-major arrays' row length
    const size_t gl = (size_t)pc * D + i;
    const LrCoef C = lr_coef(K);
    const double dt = K.dt;
    const float* xb = x + (size_t)pc * H * dim;
    const double* tab = lds + (size_t)i * stride;
    const float* jb = jac + (size_t)pc * H * (D + 1);
    const double* wb = wdense + (size_t)pc * H;
    // ---- forward.  Per step ~25 fp64 instructions on the wave's critical path (the sweep is bound by their issue, not by memory:
    //      everything it reads arrives through the rings): e_t and Qi e_t of factor (t, t + 1) once -- the row's share of factor
    //      (t - 1, t) is the previous step's Qi e, carried --, two fma per component of r_t, z_t = W_t r_t.
    float rp_[LR_PF], rv_[LR_PF], rh_[LR_PF];
    double rw_[LR_PF];
#pragma unroll
    for (int u = 0; u < LR_PF; ++u) {           // ring slot u <- waypoint 1 + u (what step u needs as its right neighbour)
        const int tn = (1 + u < H) ? 1 + u : H - 1;
        rp_[u] = xb[(size_t)tn * dim + i];
        rv_[u] = xb[(size_t)tn * dim + D + i];
        if (FINAL) {                            // (h w of waypoint u itself)
            const int tc = (u < H) ? u : H - 1;
            rh_[u] = jb[(size_t)tc * (D + 1) + i];
            rw_[u] = wb[tc];
        }
    }
    const double s_p = (double)start[(size_t)pc * dim + i], s_v = (double)start[(size_t)pc * dim + D + i];
    const double g_p = (double)goal[(size_t)pc * dim + i], g_v = (double)goal[(size_t)pc * dim + D + i];
    float pf = xb[i], vf = xb[D + i];
    double pcur = (double)pf, vcur = (double)vf;
    // the priors without a branch in the loop: row 0 takes +ks (start - x_0) -- carried in as the "previous factor" of step 0 (the
    // row's gradient is Phi^T q_t - q_{t-1}) --, row H - 1 takes +kg (goal - x_{H-1}) as the virtual q of its missing factor
    double qpl = -K.ks * (s_p - pcur), qvl = -K.ks * (s_v - vcur);
    double r0 = 0.0, r1 = 0.0, f00 = 0.0, f01 = 0.0, f10 = 0.0, f11 = 0.0, cost = 0.0;
    if (!FINAL) cost = K.ks * fma(s_p - pcur, s_p - pcur, (s_v - vcur) * (s_v - vcur));
    double* zr = zbuf + gl * LR_ZREC;
    const size_t zstep = NL * LR_ZREC;
    // (the ring slots must be (re)defined OUTSIDE any conditional: a slot loaded under `if (t < H)` reaches the next trip through a
    // phi, the copy that resolves it sits at the end of the defining block and waits for the load it has just issued -- the first
    // build of this kernel ran one memory round trip per step, 111 us.  Whole blocks of LR_PF steps carry no guard; the tail does.)
    auto fwd_step = [&](auto uc, int t) {
                constexpr int u = decltype(uc)::value;
                const float pnf = rp_[u], vnf = rv_[u];
                double hw = 0.0;
                if (FINAL) hw = (double)rh_[u] * rw_[u];
                {
                    const int tn = (t + 1 + LR_PF < H) ? t + 1 + LR_PF : H - 1;
                    rp_[u] = xb[(size_t)tn * dim + i];
                    rv_[u] = xb[(size_t)tn * dim + D + i];
                    if (FINAL) {
                        const int tc = (t + LR_PF < H) ? t + LR_PF : H - 1;
                        rh_[u] = jb[(size_t)tc * (D + 1) + i];
                        rw_[u] = wb[tc];
                    }
                }
                const double pn = (double)pnf, vn = (double)vnf;
                double qp, qv;
                if (t < H - 1) {        // factor (t, t + 1): e = x_{t+1} - Phi x_t; Qi e      (wave-uniform branch)
                    const double ep = pn - fma(dt, vcur, pcur), ev = vn - vcur;
                    qp = fma(C.bq, ev, C.a * ep);
                    qv = fma(C.cq, ev, C.bq * ep);
                    if (!FINAL) cost += fma(ep, qp, ev * qv);
                } else {                // the goal prior as the virtual q with Phi^T q = kg (goal - x)
                    const double ep = g_p - pcur, ev = g_v - vcur;
                    qp = K.kg * ep;
                    qv = fma(-dt, qp, K.kg * ev);
                    if (!FINAL) cost += K.kg * fma(ep, ep, ev * ev);
                }
                // this row takes Phi^T Qi e of its own factor and -Qi e of the previous one
                double gp = qp - qpl, gv = fma(dt, qp, qv) - qvl;
                qpl = qp; qvl = qv;
                if (FINAL) {
                    gp += hw;           // (row 0: h = 0 and w = 0)
                    for (int f = 1; f < F; ++f)          // further chained fields (rare: not prefetched)
                        gp = fma((double)jac[(((size_t)f * B + pc) * H + t) * (D + 1) + i], wdense[((size_t)f * B + pc) * H + t], gp);
                }
                const double* rc = tab + (size_t)t * LR_REC;
                const double a0 = fma(-f10, r1, fma(-f00, r0, gp)), a1 = fma(-f11, r1, fma(-f01, r0, gv));
                r0 = a0; r1 = a1;
                const double z0 = fma(rc[1], r1, rc[0] * r0), z1 = fma(rc[2], r1, rc[1] * r0);
                f00 = rc[3]; f01 = rc[4]; f10 = rc[5]; f11 = rc[6];
                if (live) {
                    zr[0] = z0; zr[1] = z1;
                    zr[2] = __longlong_as_double(((unsigned long long)__float_as_uint(vf) << 32) | __float_as_uint(pf));
                }
                zr += zstep;
                pcur = pn; vcur = vn; pf = pnf; vf = vnf;
    };
    int tb = 0;
    for (; tb + LR_PF <= H; tb += LR_PF) lr_static_for<0, LR_PF>([&](auto uc) { fwd_step(uc, tb + decltype(uc)::value); });
    lr_static_for<0, LR_PF>([&](auto uc) {
        if (tb + decltype(uc)::value < H) fwd_step(uc, tb + decltype(uc)::value);         // (wave-uniform)
    });
    if (!FINAL) {
        // the particle's share of the cost: the sum over its D lanes (a wave reduction would mix particles): through LDS
        double* red = lds + (size_t)D * stride;
        red[lane] = cost;
        __syncthreads();
        if (live && i == 0) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += red[lane + j];
            gpcost[p] = s;
        }
    }
    // ---- backward: records H - 1 .. 0 through the ring
    double q0[LR_PF], q1[LR_PF], q2[LR_PF];
#pragma unroll
    for (int u = 0; u < LR_PF; ++u) {
        const int t = (H - 1 - u >= 0) ? H - 1 - u : 0;
        const double* zr = zbuf + ((size_t)t * NL + gl) * LR_ZREC;
        q0[u] = zr[0]; q1[u] = zr[1]; q2[u] = zr[2];
    }
    double y0 = 0.0, y1 = 0.0;            // (y_H = 0: the first step takes z_{H-1} as it is)
    auto bwd_step = [&](auto uc, int t) {
                constexpr int u = decltype(uc)::value;
                const double z0 = q0[u], z1 = q1[u], xx = q2[u];
                {
                    const int tn = (t - LR_PF >= 0) ? t - LR_PF : 0;
                    const double* zq = zbuf + ((size_t)tn * NL + gl) * LR_ZREC;
                    q0[u] = zq[0]; q1[u] = zq[1]; q2[u] = zq[2];
                }
                const double* rc = tab + (size_t)t * LR_REC;
                const double a0 = fma(-rc[4], y1, fma(-rc[3], y0, z0)), a1 = fma(-rc[6], y1, fma(-rc[5], y0, z1));
                y0 = a0; y1 = a1;
                if (live) {
                    if (FINAL) {
                        const unsigned long long xu = __double_as_longlong(xx);
                        const double xp = (double)__uint_as_float((unsigned)xu), xv = (double)__uint_as_float((unsigned)(xu >> 32));
                        x[((size_t)p * H + t) * dim + i] = (float)(xp + K.step * y0);
                        x[((size_t)p * H + t) * dim + D + i] = (float)(xv + K.step * y1);
                    } else {
                        upos[(size_t)t * NL + gl] = y0;
                    }
                }
    };
    int kb = 0;
    for (; kb + LR_PF <= H; kb += LR_PF) lr_static_for<0, LR_PF>([&](auto uc) { bwd_step(uc, H - 1 - kb - decltype(uc)::value); });
    lr_static_for<0, LR_PF>([&](auto uc) {
        if (kb + decltype(uc)::value < H) bwd_step(uc, H - 1 - kb - decltype(uc)::value);
    });
}

// ------------------------------------------------------------------------------------------------
// per particle: the capacitance system of its active collision rows.  One workgroup of four waves.
//
// M (n x n, n <= 127) and the right-hand side as row n live in LDS as the lower triangle of an 8 x 8 array of 16 x 16 fp64 TILES
// (diagonal tiles full), each tile row-major with its columns XOR-swizzled by the row (element (r, c) at 16 r + (c ^ r)): the
// three access patterns of the factorisation -- a lane per row walking the columns, the matrix instruction's operand layout
// (lane (lk, li) -> element [li][4 kc + lk]) and its accumulator layout (rows lk + 4 q of column li) -- are all conflict free.
// Tile Cholesky, right looking, one tile column J at a time:
//     POTRF  wave 0, a lane per row of the diagonal tile, the row in registers; pivots and rank-1 updates through v_readlane;
//     TRSM   a lane per row of the tiles below it (forward substitution against L_JJ: its entries are uniform LDS reads);
//     GEMM   A_IK -= L_IJ L_KJ^T on the matrix cores, four v_mfma_f64_16x16x4_f64 per tile, tiles dealt to the four waves.
// Row n rides along as an ordinary row and comes out as y = L^-1 rhs; L^T w = y is a column-oriented sweep by one wave.
// (The first version of the round ran a left-looking column Cholesky on one wave out of a packed triangle: a particle with 116
// active rows -- the largest of the 2 048 at C4 -- took ~0.45 ms, and the kernel lasts as long as its slowest particle.)
// ------------------------------------------------------------------------------------------------
#define LR_TILE 256                                                          // doubles per tile
__device__ __forceinline__ int lr_tile(int I, int J) { return ((I * (I + 1)) >> 1) + J; }            // J <= I
__device__ __forceinline__ int lr_sw(int r, int c) { return (r << 4) + (c ^ r); }
__device__ __forceinline__ double lr_rsqrt(double x) {                      // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(fma(-x * y, y, 1.0), 0.5, 1.0);
    y = y * fma(fma(-x * y, y, 1.0), 0.5, 1.0);
    return y;
}

__global__ __launch_bounds__(256) void gpmp2_lr_cap(const float* __restrict__ jac, const double* __restrict__ upos,
                                                   const double* __restrict__ G, const double* __restrict__ gpcost,
                                                   double* __restrict__ wdense, float* __restrict__ costs_out, int B, int H, int D, int F,
                                                   int n_tiles_max, GpConst K) {
    extern __shared__ double lds[];
    // LDS: [ tiles | rhs / y / w (128) | 1 / l_kk (128) | h rows of the active set (LR_NMAX x 8 fp32) | waypoint and field of every
    //        active row (2 x 128 ints) | scratch ints ]
    double* Tl = lds;
    double* wv = lds + (size_t)n_tiles_max * LR_TILE;
    double* dinv = wv + 128;
    float* hb = reinterpret_cast<float*>(dinv + 128);
    const int hs = (D <= 8) ? 8 : MPB_MAX_DOF;                  // floats per row of h (the launcher sizes the LDS for it)
    int* tact = reinterpret_cast<int*>(hb + LR_NMAX * hs);
    int* fact = tact + 128;
    int* cnt = fact + 128;                                       // [0 .. 2 F): active rows of (field, 64-waypoint chunk)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const size_t NL = (size_t)B * D;
    // ---- 1. the active collision rows, compacted (field major, then waypoint); the collision part of the cost.  Chunk (f, base) is
    //         examined by wave (2 f + base / 64) mod 4: counts first, then positions
    const int nchunk = F * ((H + 63) >> 6);
    double cost = 0.0;
    float hrow[MPB_MAX_DOF + 1];
    bool act = false;
    int myt = 0, myf = 0;
    // (F (H - 1) <= 127: at most 8 chunks, two per wave; a wave keeps the rows of its SECOND chunk in the second set below)
    float hrow2[MPB_MAX_DOF + 1];
    bool act2 = false;
    int myt2 = 0, myf2 = 0;
    for (int ch = wave; ch < nchunk; ch += 4) {
        const int f = ch / ((H + 63) >> 6), base = (ch - f * ((H + 63) >> 6)) << 6;
        const int t = base + lane;
        const float* jb = jac + ((size_t)f * B + b) * H * (D + 1);
        float hr[MPB_MAX_DOF + 1];
        bool a = false;
#pragma unroll
        for (int j = 0; j <= MPB_MAX_DOF; ++j) {
            hr[j] = (t < H && t > 0 && j <= D) ? jb[(size_t)t * (D + 1) + j] : 0.f;      // row 0 takes no collision factor
            if (j < D) a = a || (hr[j] != 0.f);
        }
        if (t < H) {
            if (t > 0) cost += K.kc * (double)hr[D] * (double)hr[D];
            if (!a) wdense[((size_t)f * B + b) * H + t] = 0.0;          // (active rows get their w in step 6: every word is written once)
        }
        const unsigned long long m = __ballot(a);
        if (lane == 0) cnt[ch] = __popcll(m);
        if (ch < 4) {
#pragma unroll
            for (int j = 0; j <= MPB_MAX_DOF; ++j) hrow[j] = hr[j];
            act = a; myt = t; myf = f;
        } else {
#pragma unroll
            for (int j = 0; j <= MPB_MAX_DOF; ++j) hrow2[j] = hr[j];
            act2 = a; myt2 = t; myf2 = f;
        }
    }
    // the cost: the four waves' partial sums through LDS
    cost = wave_sum_f64(cost);
    if (lane == 0) dinv[wave] = cost;
    __syncthreads();
    if (costs_out != nullptr && tid == 0) costs_out[b] = (float)(((dinv[0] + dinv[1]) + (dinv[2] + dinv[3])) + gpcost[b]);
    int n = 0;
    for (int ch = 0; ch < nchunk; ++ch) n += cnt[ch];
    if (n == 0) return;                                          // (block-uniform)
    for (int ps = 0; ps < 2; ++ps) {
        const int ch = wave + 4 * ps;
        if (ch < nchunk) {
            int off = 0;
            for (int e = 0; e < ch; ++e) off += cnt[e];
            const bool a = ps ? act2 : act;
            const unsigned long long m = __ballot(a);
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (a) {
                tact[pos] = ps ? myt2 : myt;
                fact[pos] = ps ? myf2 : myf;
#pragma unroll
                for (int j = 0; j < MPB_MAX_DOF; ++j)
                    if (j < hs) hb[pos * hs + j] = (j < D) ? (ps ? hrow2[j] : hrow[j]) : 0.f;
                wv[pos] = (double)(ps ? hrow2[D] : hrow[D]);     // c of the row (until the right-hand side takes the slot)
            }
        }
    }
    __syncthreads();
    // ---- 2. right-hand side: c_a - h_a . u0[position rows of t_a]
    if (tid < n) {
        double sacc = wv[tid];
        const double* up = upos + (size_t)tact[tid] * NL + (size_t)b * D;
        for (int j = 0; j < D; ++j) sacc -= (double)hb[tid * hs + j] * up[j];
        wv[tid] = sacc;
    }
    __syncthreads();
    // ---- 3. the tiles: M[a][c] = [a == c] / kc + sum_i h_a,i h_c,i G_i(t_a, t_c); row n = the right-hand side; identity beyond.
    //         A tile per pass of the 256 threads, the NEXT tile's G entries (L2) fetched before the current tile's arithmetic
    const int TR = (n + 16) >> 4;                                // tile rows that hold rows 0 .. n
    const int ntl = (TR * (TR + 1)) >> 1;
    {
        const double ikc = 1.0 / K.kc;
        const int r = tid >> 4, c = tid & 15;
        auto fetch = [&](int I, int J, double (&gv)[MPB_MAX_DOF]) {
            int a = 16 * I + r, cc = 16 * J + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }              // (diagonal tiles are stored full: mirror)
            const bool on = a < n;
            const double* Gst = G + (size_t)tact[on ? a : 0] * H + tact[on ? cc : 0];
#pragma unroll
            for (int j = 0; j < MPB_MAX_DOF; ++j) gv[j] = (j < D) ? Gst[(size_t)j * H * H] : 0.0;
        };
        double gcur[MPB_MAX_DOF], gnxt[MPB_MAX_DOF];
        int I = 0, J = 0;
        fetch(0, 0, gcur);
        for (int q = 0; q < ntl; ++q) {
            int In = I, Jn = J + 1;
            if (Jn > In) { ++In; Jn = 0; }
            if (q + 1 < ntl) fetch(In, Jn, gnxt);
            int a = 16 * I + r, cc = 16 * J + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }
            double m;
            if (a < n) {
                m = (a == cc) ? ikc : 0.0;
#pragma unroll
                for (int j = 0; j < MPB_MAX_DOF; ++j)
                    if (j < D) m = fma((double)hb[a * hs + j] * (double)hb[cc * hs + j], gcur[j], m);
            } else if (a == n && cc < n) {
                m = wv[cc];
            } else {
                m = (a == cc) ? 1.0 : 0.0;
            }
            Tl[(size_t)lr_tile(I, J) * LR_TILE + lr_sw(r, c)] = m;
#pragma unroll
            for (int j = 0; j < MPB_MAX_DOF; ++j) gcur[j] = gnxt[j];
            I = In; J = Jn;
        }
    }
    __syncthreads();
    // ---- 4. tile Cholesky
    const int TC = (n + 15) >> 4;                                // tile columns with a pivot
    const int li = lane & 15, lk = lane >> 4;
    for (int J = 0; J < TC; ++J) {
        double* Djj = Tl + (size_t)lr_tile(J, J) * LR_TILE;
        if (wave == 0) {
            // POTRF: lane r < 16 holds row r of the diagonal tile
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Djj[lr_sw(r, c)];
            const int npiv = (n - 16 * J < 16) ? n - 16 * J : 16;          // pivots of this tile (the rest is padding / the rhs row)
            lr_static_for<0, 16>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                if (p < npiv) {                                              // (wave-uniform)
                    const double d = lr_readlane(a[p], p);
                    const double rs = lr_rsqrt(d);
                    const double l = a[p] * rs;
                    a[p] = (r == p) ? d * rs : l;
                    if (lane == 0) dinv[16 * J + p] = rs;
                    lr_static_for<p + 1, 16>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        a[c] = fma(-l, lr_readlane(l, c), a[c]);
                    });
                }
            });
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Djj[lr_sw(r, c)] = a[c];
            }
        }
        __syncthreads();
        // TRSM: rows of the tiles (I, J), I > J: x L_JJ^T = a by forward substitution, a lane per row
        {
            const int rows_below = 16 * (TR - J - 1);
            for (int rr = tid; rr < rows_below; rr += 256) {
                const int I = J + 1 + (rr >> 4), r = rr & 15;
                double* Tij = Tl + (size_t)lr_tile(I, J) * LR_TILE;
                double xv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) xv[c] = Tij[lr_sw(r, c)];
                // (column by column, every later column updated as soon as x_c is known: the dependent chain is 16 x (mul, fma), where
                // the dot-product form accumulated 120 fma one after the other)
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (16 * J + c < n) xv[c] *= dinv[16 * J + c];                  // (padding columns: the identity)
#pragma unroll
                    for (int j = c + 1; j < 16; ++j) xv[j] = fma(-xv[c], Djj[lr_sw(j, c)], xv[j]);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) Tij[lr_sw(r, c)] = xv[c];
            }
        }
        __syncthreads();
        // GEMM: A_IK -= L_IJ L_KJ^T for J < K <= I < TR on the matrix cores, tiles dealt to the waves
        {
            const int nb = TR - J - 1;
            const int ng = (nb * (nb + 1)) >> 1;
            int I = J + 1, Kt = J + 1;
            for (int g = 0; g < ng; ++g) {
                if ((g & 3) == wave) {
                    double* C = Tl + (size_t)lr_tile(I, Kt) * LR_TILE;
                    const double* A = Tl + (size_t)lr_tile(I, J) * LR_TILE;
                    const double* Bt = Tl + (size_t)lr_tile(Kt, J) * LR_TILE;
                    typedef double f64x4 __attribute__((ext_vector_type(4)));
                    f64x4 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = C[lr_sw(lk + 4 * q, li)];
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[lr_sw(li, 4 * kc + lk)], Bt[lr_sw(li, 4 * kc + lk)], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) C[lr_sw(lk + 4 * q, li)] = acc[q];
                }
                if (++Kt > I) { ++I; Kt = J + 1; }
            }
        }
        __syncthreads();
    }
    // ---- 5. L^T w = y (y = row n), column oriented, one wave: w_k = y_k / l_kk, then y_j -= l_kj w_k for j < k (lanes = j, j + 64)
    if (wave == 0) {
        const int In = n >> 4, rn = n & 15;
        double y[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = lane + 64 * ps;
            y[ps] = (j < n) ? Tl[(size_t)lr_tile(In, j >> 4) * LR_TILE + lr_sw(rn, j & 15)] : 0.0;
        }
        for (int k = n - 1; k >= 0; --k) {
            const double yk = (k < 64) ? lr_readlane(y[0], k) : lr_readlane(y[1], k - 64);
            const double wk = yk * dinv[k];
            if (lane == 0) wv[k] = wk;
            const double* Lk = Tl + (size_t)lr_tile(k >> 4, 0) * LR_TILE;       // tile row of k
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int j = lane + 64 * ps;
                if (j < k) y[ps] = fma(-Lk[(size_t)(j >> 4) * LR_TILE + lr_sw(k & 15, j & 15)], wk, y[ps]);
            }
        }
    }
    __syncthreads();
    // ---- 6. w to its waypoints
    if (tid < n) wdense[((size_t)fact[tid] * B + b) * H + tact[tid]] = wv[tid];
}

// ------------------------------------------------------------------------------------------------
// launcher (called by mpb_gpmp2_solve, mpb_gpmp2.hip)
// ------------------------------------------------------------------------------------------------
bool mpb_gpmp2_lr_ok(int H, int D, int n_fields) { return H >= 2 && D >= 1 && D <= MPB_MAX_DOF && n_fields >= 1 && n_fields * (H - 1) <= LR_NMAX; }

// doubles of workspace: shared tables (factor records, G) + per-batch arrays (sweep records, u0's position rows, w, the GP cost)
size_t mpb_gpmp2_lr_ws_doubles(int B, int H, int D) {
    const size_t NL = (size_t)B * D;
    return (size_t)D * H * LR_REC + (size_t)D * H * H + (size_t)H * NL * LR_ZREC + (size_t)H * NL + (size_t)MPB_GP_MAX_FIELDS * B * H + (size_t)B + 64;
}

int mpb_gpmp2_lr_launch(float* x, const float* start, const float* goal, const float* jac, const double* diag_mean, double* ws,
                        float* costs_out, int B, int H, int D, int n_fields, const GpConst& K, hipStream_t stream) {
    const size_t NL = (size_t)B * D;
    double* rec = ws;
    double* G = rec + (size_t)D * H * LR_REC;
    double* zbuf = G + (size_t)D * H * H;
    double* upos = zbuf + (size_t)H * NL * LR_ZREC;
    double* wdense = upos + (size_t)H * NL;
    double* gpcost = wdense + (size_t)MPB_GP_MAX_FIELDS * B * H;
    const size_t lds_chain = ((size_t)H * LR_REC + (size_t)H * 2) * sizeof(double);
    hipLaunchKernelGGL(gpmp2_chain_kernel, dim3(D, (H + LR_COLS - 1) / LR_COLS), dim3(LR_COLS), lds_chain, stream, diag_mean, rec, G, H, D, K);
    const int per = 64 / D;
    const size_t lds_sweep = ((size_t)D * (H * LR_REC + 2) + 64) * sizeof(double);
    const dim3 gs((B + per - 1) / per);
    hipLaunchKernelGGL(gpmp2_lr_sweep<false>, gs, dim3(64), lds_sweep, stream, x, start, goal, jac, wdense, rec, zbuf, upos, gpcost, B, H, D,
                       n_fields, K);
    const int n_max = n_fields * (H - 1);
    const int trm = (n_max + 16) >> 4, ntm = (trm * (trm + 1)) >> 1;             // tiles of the largest system the shape allows
    const size_t lds = ((size_t)ntm * LR_TILE + 256) * sizeof(double) + (size_t)LR_NMAX * (D <= 8 ? 8 : MPB_MAX_DOF) * sizeof(float) + (256 + 16) * sizeof(int);
    hipLaunchKernelGGL(gpmp2_lr_cap, dim3(B), dim3(256), lds, stream, jac, upos, G, gpcost, wdense, costs_out, B, H, D, n_fields, ntm, K);
    hipLaunchKernelGGL(gpmp2_lr_sweep<true>, gs, dim3(64), lds_sweep, stream, x, start, goal, jac, wdense, rec, zbuf, upos, gpcost, B, H, D,
                       n_fields, K);
    return MPB_OK;
}
