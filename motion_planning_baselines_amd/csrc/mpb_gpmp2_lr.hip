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
// Five launches per iteration, every one of them bound by the LATENCY of one particle's (or one joint's) dependent steps, not by
// the batch (a lone wave issues one instruction per four cycles, an fp64 one per eight: what counts is the instruction COUNT on
// the sequential path -- DESIGN.md section 6 has the per-kernel times on one particle):
//   gpmp2_chain_kernel    shared by all particles: the block-Thomas factors (W_t = S_t^-1, F_t = W_t U) of the D chains and the
//                         position-position entries G_i(s, t) of their inverses, D H^2 doubles (0.9 MB at C4: stays in L2).  One
//                         lane walks the factorisation and the diagonal recurrence; what does not depend on the previous step
//                         (D_t before, W_t / F_t / the recurrence's coefficients after) is done by all lanes and staged in LDS;
//   gpmp2_lr_gradient     g_rest (priors + GP factors) and its cost, a lane per (waypoint, joint): nothing in it is sequential,
//                         so it stays out of the sweeps (formed on the fly there it was 40 of a step's 60 instructions);
//   gpmp2_lr_sweep<false> u0: lane = (particle, joint) -- 9 particles per wave at D = 7 --, the factor table in LDS (records read
//                         one step ahead), H steps down and H up with two dependent fma per step; g and the z_t records travel
//                         through register rings 16 steps ahead (the first version of the round walked the chain inside the
//                         per-particle kernel with the factors read from L2 one step ahead: 512 L2 round trips in a row per
//                         particle, 0.91 ms at C4);
//   gpmp2_lr_cap          per particle, four waves: the active rows compacted by ballot, M (n_a x n_a) from the G table in 16 x 16
//                         tiles into LDS (three tiles of loads in flight), tile Cholesky (POTRF in registers by readlane, TRSM a
//                         lane per row, the trailing update on the fp64 MFMA), the right-hand side carried as row n_a, a blocked
//                         back substitution, w scattered to a dense per-waypoint vector;
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
#ifndef LR_PF
#define LR_PF 16                       // steps the sweeps read ahead (register rings)
#endif
#define LR_ZREC 2                      // doubles per (waypoint, lane) record of a sweep: z0, z1

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
#define LR_HMAX 128                    // waypoints (mpb_gpmp2_lr_ok: n_fields (H - 1) <= LR_NMAX)
#define LR_ORD 9                       // size classes of the capacitance systems' launch order: 0 rows, 1-16, ..., 113-128 (largest first)
#define LR_KK 10                       // doubles per waypoint of the chain kernel's coefficient table (nine used; even, so that pairs stay aligned)
__global__ __launch_bounds__(LR_COLS) void gpmp2_chain_kernel(const double* __restrict__ diag_mean, double* __restrict__ rec_g,
                                                              double* __restrict__ G, int H, int D, GpConst K) {
    extern __shared__ double lds[];
    double* rec = lds;                                 // H x LR_REC
    double* gd = lds + (size_t)H * LR_REC;             // H x 2: column 0 of G_{t,t}
    double* kk = gd + (size_t)H * 2;                   // H x LR_KK: first the chain's diagonal blocks D_t, then the coefficients of the diagonal recurrence
    const int i = blockIdx.x, c0 = blockIdx.y * LR_COLS, tid = threadIdx.x;
    const LrCoef C = lr_coef(K);
    const int dim = 2 * D;
#ifdef LR_T_CLK
    unsigned long long clk_[6]; clk_[0] = wall_clock64();
#define LR_CLK(k) clk_[k] = wall_clock64()
#else
#define LR_CLK(k)
#endif
    // ONE lane walks the two sequential recurrences below, and a wave issues one fp64 instruction per four cycles whatever the number
    // of its lanes at work: a step costs its instruction COUNT (measured: 50 instructions a step, 290 cycles), not the depth of its
    // dependent chain.  So everything that does not depend on the previous step is done by all lanes, before or after, and staged.
    // D_t = (GP blocks) + damping + start / goal prior, a lane per waypoint
    for (int t = tid; t < H; t += LR_COLS) {
        const double first = (t == 0) ? 1.0 : 0.0, last = (t == H - 1) ? 1.0 : 0.0;
        const double dp = K.trust ? K.delta * diag_mean[(size_t)t * dim + i] : K.delta;
        const double dv = K.trust ? K.delta * diag_mean[(size_t)t * dim + D + i] : K.delta;
        kk[t * LR_KK] = (1.0 - last) * C.p00 + (1.0 - first) * C.a + dp + first * K.ks + last * K.kg;
        kk[t * LR_KK + 1] = (1.0 - last) * C.p01 + (1.0 - first) * C.bq;
        kk[t * LR_KK + 2] = (1.0 - last) * C.p11 + (1.0 - first) * C.cq + dv + first * K.ks + last * K.kg;
    }
    __syncthreads();
    LR_CLK(1);
    if (tid == 0) {
        // block Thomas on the 2 x 2 chain of joint i:  S_0 = D_0,  S_{t+1} = D_{t+1} - U^T S_t^-1 U.  With S^-1 = adj(S) / det,
        // U^T S^-1 U = N / det where N = U^T adj(S) U is three fma chains with CONSTANT coefficients (products of U's entries):
        // 9 + 3 (S) + 2 (det) + 5 (reciprocal) instructions a step; S_t and 1 / det are kept, W_t and F_t follow in parallel below.
        const double c00a = C.u00 * C.u00, c00b = -2.0 * C.u00 * C.u10, c00c = C.u10 * C.u10;                  // N00 = c00a s11 + c00b s01 + c00c s00
        const double c01a = C.u00 * C.u01, c01b = -(C.u00 * C.u11 + C.u10 * C.u01), c01c = C.u10 * C.u11;      // N01
        const double c11a = C.u01 * C.u01, c11b = -2.0 * C.u01 * C.u11, c11c = C.u11 * C.u11;                  // N11
        double s00 = 0.0, s01 = 0.0, s11 = 0.0, idp = 0.0;        // (t = 0: the Schur term vanishes)
        double e00 = kk[0], e01 = kk[1], e11 = kk[2];
#pragma unroll 4
        for (int t = 0; t < H; ++t) {
            const double d00 = e00, d01 = e01, d11 = e11;
            const int tn = (t + 1 < H) ? t + 1 : t;           // the next block is read BEFORE this step's stores (the compiler cannot
            e00 = kk[tn * LR_KK]; e01 = kk[tn * LR_KK + 1]; e11 = kk[tn * LR_KK + 2];      // tell the two LDS arrays apart)
            const double n00 = fma(c00a, s11, fma(c00b, s01, c00c * s00));
            const double n01 = fma(c01a, s11, fma(c01b, s01, c01c * s00));
            const double n11 = fma(c11a, s11, fma(c11b, s01, c11c * s00));
            s00 = fma(-n00, idp, d00);
            s01 = fma(-n01, idp, d01);
            s11 = fma(-n11, idp, d11);
            idp = lr_rcp(fma(s00, s11, -s01 * s01));
            double* r = rec + (size_t)t * LR_REC;
            r[0] = s00; r[1] = s01; r[2] = s11; r[3] = idp;
        }
    }
    __syncthreads();
    LR_CLK(2);
    // W_t = adj(S_t) / det,  F_t = W_t U,  and the coefficients of  G_tt = W_t + F_t G_{t+1,t+1} F_t^T  written out in G's three entries
    for (int t = tid; t < H; t += LR_COLS) {
        double* r = rec + (size_t)t * LR_REC;
        const double idp = r[3];
        const double w00 = r[2] * idp, w01 = -r[1] * idp, w11 = r[0] * idp;
        const double f00 = w00 * C.u00 + w01 * C.u10, f01 = w00 * C.u01 + w01 * C.u11;
        const double f10 = w01 * C.u00 + w11 * C.u10, f11 = w01 * C.u01 + w11 * C.u11;
        r[0] = w00; r[1] = w01; r[2] = w11; r[3] = f00; r[4] = f01; r[5] = f10; r[6] = f11; r[7] = 0.0;
        double* k = kk + t * LR_KK;
        k[0] = f00 * f00; k[1] = 2.0 * f00 * f01; k[2] = f01 * f01;
        k[3] = f00 * f10; k[4] = fma(f00, f11, f01 * f10); k[5] = f01 * f11;
        k[6] = f10 * f10; k[7] = 2.0 * f10 * f11; k[8] = f11 * f11;
    }
    __syncthreads();
    LR_CLK(3);
    if (tid == 0) {
        // diagonal blocks of the inverse, bottom up; column 0 (the response to a unit POSITION entry) of each is kept.  This block
        // walks columns c0 .. c0 + 31 only, so the recurrence stops at c0 (the block of the last columns has the longest walks
        // and the shortest recurrence)
        double g00 = rec[(size_t)(H - 1) * LR_REC], g01 = rec[(size_t)(H - 1) * LR_REC + 1], g11 = rec[(size_t)(H - 1) * LR_REC + 2];
        gd[2 * (H - 1)] = g00; gd[2 * (H - 1) + 1] = g01;
        const int t0 = (H >= 2) ? H - 2 : 0;
        double w0 = rec[(size_t)t0 * LR_REC], w1 = rec[(size_t)t0 * LR_REC + 1], w2 = rec[(size_t)t0 * LR_REC + 2];
        double k0 = kk[t0 * LR_KK], k1 = kk[t0 * LR_KK + 1], k2 = kk[t0 * LR_KK + 2], k3 = kk[t0 * LR_KK + 3], k4 = kk[t0 * LR_KK + 4];
        double k5 = kk[t0 * LR_KK + 5], k6 = kk[t0 * LR_KK + 6], k7 = kk[t0 * LR_KK + 7], k8 = kk[t0 * LR_KK + 8];
#pragma unroll 4
        for (int t = H - 2; t >= c0; --t) {
            const double n00 = fma(k0, g00, w0) + fma(k2, g11, k1 * g01);
            const double n01 = fma(k3, g00, w1) + fma(k5, g11, k4 * g01);
            const double n11 = fma(k6, g00, w2) + fma(k8, g11, k7 * g01);
            const int tn = (t > 0) ? t - 1 : 0;
            w0 = rec[(size_t)tn * LR_REC]; w1 = rec[(size_t)tn * LR_REC + 1]; w2 = rec[(size_t)tn * LR_REC + 2];
            k0 = kk[tn * LR_KK]; k1 = kk[tn * LR_KK + 1]; k2 = kk[tn * LR_KK + 2]; k3 = kk[tn * LR_KK + 3]; k4 = kk[tn * LR_KK + 4];
            k5 = kk[tn * LR_KK + 5]; k6 = kk[tn * LR_KK + 6]; k7 = kk[tn * LR_KK + 7]; k8 = kk[tn * LR_KK + 8];
            g00 = n00; g01 = n01; g11 = n11;
            gd[2 * t] = g00; gd[2 * t + 1] = g01;
        }
    }
    __syncthreads();
    LR_CLK(4);
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
#ifdef LR_T_CLK
    LR_CLK(5);
    if (blockIdx.x == 0 && c == H - 1)
        printf("chain clk (10 ns): stage %llu fact %llu post %llu diag %llu cols %llu\n", clk_[1] - clk_[0], clk_[2] - clk_[1], clk_[3] - clk_[2],
               clk_[4] - clk_[3], clk_[5] - clk_[4]);
#endif
}

// ------------------------------------------------------------------------------------------------
// g_rest: the gradient without its collision part (start / goal priors and the GP factors, gpmp2.py:355-368 with the rows of
// cost_functions.py:291-314, :538-554), fp64, laid out like x ((B, H, 2D)), and the cost b^T K b of those factors per particle
// (gpmp2.py:493-495).  One wave per particle, lane = waypoint: nothing here depends on anything else, so it is kept OUT of the sweeps
// (round 6, second version: formed on the fly inside the sweep it was 40 of a step's 60 instructions on the chain every lane walks
// alone -- a sweep over one particle took 49 us whatever the batch size).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpmp2_lr_gradient(const float* __restrict__ x, const float* __restrict__ start, const float* __restrict__ goal,
                                                         const float* __restrict__ jac, double* __restrict__ g, double* __restrict__ gpcost,
                                                         int* __restrict__ ord, int B, int H, int D, int F, GpConst K) {
    __shared__ double red_c[4];
    __shared__ int red_n[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // The capacitance kernel lasts as long as its largest system (one particle with 116 active rows: 87 us; the median particle has
    // none), and two workgroups fit a CU: in batch order the largest system of C4 started in the third of four rounds.  So every
    // particle gets a size class here -- rows with a non-zero hinge value, LR_ORD classes of 16 --, one extra workgroup of the
    // sweep that follows sorts the particles by class, largest first (a counting sort in LDS; one atomic per particle on a
    // counter per class in global memory cost this kernel 19 us: 2 048 returning atomics on nine addresses), and gpmp2_lr_cap
    // takes them in that order.  (Scheduling only: the class is an estimate, nothing else reads it.)
    int n_est = 0;
    const int dim = 2 * D;
    const LrCoef C = lr_coef(K);
    const double dt = K.dt;
    double* gb = g + (size_t)b * H * dim;
    double cost = 0.0;
    // the particle's trajectory through LDS: every load of the workgroup is in flight at once (read in place, an element's six
    // neighbours were a round trip per trip of the loop below: 35 us at C4 for 52 MB of traffic)
    __shared__ float xb[LR_HMAX * 2 * MPB_MAX_DOF];
    {
        const float* xg = x + (size_t)b * H * dim;
        const int nx = H * dim;
        float tmp[(LR_HMAX * 2 * MPB_MAX_DOF + 255) / 256];
#pragma unroll
        for (int u = 0; u < (LR_HMAX * 2 * MPB_MAX_DOF + 255) / 256; ++u) {
            const int e = tid + 256 * u;
            tmp[u] = (e < nx) ? xg[e] : 0.f;
        }
        // (the size estimate's loads behind the trajectory's, ahead of the first wait)
        for (int r0 = 0; r0 < F * H; r0 += 256) {
            const int r = r0 + tid, f = r / H, t = r - f * H;
            const bool a = r < F * H && t > 0 && jac[(((size_t)f * B + b) * H + t) * (D + 1) + D] != 0.f;
            n_est += __popcll(__ballot(a));
        }
#pragma unroll
        for (int u = 0; u < (LR_HMAX * 2 * MPB_MAX_DOF + 255) / 256; ++u) {
            const int e = tid + 256 * u;
            if (e < nx) xb[e] = tmp[u];
        }
    }
    __syncthreads();
    // a lane per (waypoint, joint), consecutive lanes = consecutive joints of a row: the D positions (and the D velocities) of a
    // row are read and written as one piece (a lane per waypoint walking the joints touched 64 rows per instruction: 37 us at C4)
    for (int e = tid; e < H * D; e += 256) {
        const int t = e / D, j = e - t * D;
        const float* xt = xb + (size_t)t * dim;
        const double p = (double)xt[j], v = (double)xt[D + j];
        double gp = 0.0, gv = 0.0;
        if (t < H - 1) {        // factor (t, t + 1): e = x_{t+1} - Phi x_t; this row takes Phi^T Qi e
            const double ep = (double)xt[dim + j] - fma(dt, v, p), ev = (double)xt[dim + D + j] - v;
            const double qp = fma(C.bq, ev, C.a * ep), qv = fma(C.cq, ev, C.bq * ep);
            gp = qp;
            gv = fma(dt, qp, qv);
            cost += fma(ep, qp, ev * qv);
        }
        if (t > 0) {            // factor (t - 1, t): this row takes -Qi e
            const double pl = (double)xt[j - dim], vl = (double)xt[D + j - dim];
            const double ep = p - fma(dt, vl, pl), ev = v - vl;
            gp -= fma(C.bq, ev, C.a * ep);
            gv -= fma(C.cq, ev, C.bq * ep);
        }
        if (t == 0) {
            const double ep = (double)start[(size_t)b * dim + j] - p, ev = (double)start[(size_t)b * dim + D + j] - v;
            gp = fma(K.ks, ep, gp); gv = fma(K.ks, ev, gv);
            cost += K.ks * fma(ep, ep, ev * ev);
        }
        if (t == H - 1) {
            const double ep = (double)goal[(size_t)b * dim + j] - p, ev = (double)goal[(size_t)b * dim + D + j] - v;
            gp = fma(K.kg, ep, gp); gv = fma(K.kg, ev, gv);
            cost += K.kg * fma(ep, ep, ev * ev);
        }
#ifndef LR_T_GRAD_NOSTORE
        gb[(size_t)t * dim + j] = gp;
        gb[(size_t)t * dim + D + j] = gv;
#else
        if (gp == 1.2345 && gv == 2.3456) gb[0] = 0.0;
#endif
    }
    cost = wave_sum_f64(cost);
    if (lane == 0) { red_c[wave] = cost; red_n[wave] = n_est; }
    __syncthreads();
    if (tid == 0) {
        gpcost[b] = (red_c[0] + red_c[1]) + (red_c[2] + red_c[3]);
        const int n = red_n[0] + red_n[1] + red_n[2] + red_n[3];
        ord[b] = LR_ORD - 1 - min(LR_ORD - 1, (n + 15) >> 4);          // size class, 0 = largest (sorted by gpmp2_lr_sweep<false>'s extra workgroup)
    }
}

// ------------------------------------------------------------------------------------------------
// A0^-1 applied to a gradient: lane = (particle, joint) -- 9 particles per wave at D = 7 --, the shared factors in LDS.
// FINAL = false: the gradient is g_rest; writes the position rows of u0 = A0^-1 g_rest (t major: upos[t][lane]).
// FINAL = true: the gradient is g_rest + V w (w: dense per field and waypoint, zero off the active rows); writes x += step * dtheta
// (gpmp2.py:326-331).  Forward r_t = g_t - F_{t-1}^T r_{t-1}, z_t = W_t r_t (records to zbuf, t major); backward
// y_t = z_t - F_t y_{t+1}.  A lane walks its chain alone (the kernel lasts 2 H steps whatever the batch): a step is the ring
// hand-over of what it reads (16 steps ahead), four fma on the chain, four off it, one record store.
// ------------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ __launch_bounds__(64) void gpmp2_lr_sweep(float* __restrict__ x, const double* __restrict__ g, const float* __restrict__ jac,
                                                     const double* __restrict__ wdense, const double* __restrict__ rec_g,
                                                     double* __restrict__ zbuf, double* __restrict__ upos, int* __restrict__ ord, int B, int H, int D,
                                                     int F, GpConst K) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, dim = 2 * D;
    if (!FINAL && blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: the particles sorted by size class (gpmp2_lr_gradient), largest first -- the order gpmp2_lr_cap
        // takes them in.  A counting sort: class totals and positions by LDS atomics.
        int* cnt = reinterpret_cast<int*>(lds);            // [0, LR_ORD): totals, then running offsets
        if (lane < LR_ORD) cnt[lane] = 0;
        lr_wave_sync();
        for (int b0 = 0; b0 < B; b0 += 64)
            if (b0 + lane < B) atomicAdd(&cnt[ord[b0 + lane]], 1);
        lr_wave_sync();
        if (lane == 0) {
            int run = 0;
            for (int c = 0; c < LR_ORD; ++c) { const int n = cnt[c]; cnt[c] = run; run += n; }
        }
        lr_wave_sync();
        for (int b0 = 0; b0 < B; b0 += 64)
            if (b0 + lane < B) ord[B + atomicAdd(&cnt[ord[b0 + lane]], 1)] = b0 + lane;
        return;
    }
    const int per = 64 / D;                                   // particles per wave
    const int stride = H * LR_REC + 2;                        // doubles between two joints' tables (+ 2: their records fall on different banks)
    // the shared factors into LDS, 16 bytes at a time, EIGHT loads in flight per lane and joint (one load per trip with its index
    // division cost the kernel ~30 us of serialised L2 round trips before its first step: a sweep took 43 us whatever it did per step)
    for (int j = 0; j < D; ++j) {
        const lr_d2* src = reinterpret_cast<const lr_d2*>(rec_g + (size_t)j * H * LR_REC);
        lr_d2* dst = reinterpret_cast<lr_d2*>(lds + (size_t)j * stride);
        const int n2 = H * (LR_REC / 2);
        for (int e0 = 0; e0 < n2; e0 += 64 * 8) {
            lr_d2 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + lane + 64 * u;
                tmp[u] = src[e < n2 ? e : n2 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + lane + 64 * u;
                if (e < n2) dst[e] = tmp[u];
            }
        }
    }
    __syncthreads();
    const int pl = lane / D, i = lane - pl * D;
    const int p = blockIdx.x * per + pl;
    if (pl >= per || p >= B) return;                          // (no block barrier below: idle lanes leave, nothing is masked per step)
    const bool live = true;
    const int pc = p;
    const size_t NL = (size_t)B * D;                          // lanes of the whole batch: the t-major arrays' row length
    const size_t gl = (size_t)pc * D + i;
    const double* gb = g + (size_t)pc * H * dim;
    const double* tab = lds + (size_t)i * stride;
    const float* jb = jac + (size_t)pc * H * (D + 1);
    const double* wb = wdense + (size_t)pc * H;
    // ---- forward
    double rg0_[LR_PF], rg1_[LR_PF], rw_[LR_PF];
    float rh_[LR_PF];
#pragma unroll
    for (int u = 0; u < LR_PF; ++u) {           // ring slot u <- waypoint u
        const int tc = (u < H) ? u : H - 1;
        rg0_[u] = gb[(size_t)tc * dim + i];
        rg1_[u] = gb[(size_t)tc * dim + D + i];
        if (FINAL) {
            rh_[u] = jb[(size_t)tc * (D + 1) + i];
            rw_[u] = wb[tc];
        }
    }
    double r0 = 0.0, r1 = 0.0, f00 = 0.0, f01 = 0.0, f10 = 0.0, f11 = 0.0;
    // the joint's factor record of the CURRENT step, read from LDS one step ahead (R0 .. R3 = W00 W01 | W11 F00 | F01 F10 | F11 -):
    // read in the step that uses it, the LDS round trip sat on every step of the chain (~100 of a step's ~290 cycles)
    const lr_d2* tab2 = reinterpret_cast<const lr_d2*>(tab);
    lr_d2 R0 = tab2[0], R1 = tab2[1], R2 = tab2[2], R3 = tab2[3];
    double* zr = zbuf + gl * LR_ZREC;
    const size_t zstep = NL * LR_ZREC;
    // (the ring slots must be (re)defined OUTSIDE any conditional: a slot loaded under `if (t < H)` reaches the next trip through a
    // phi, the copy that resolves it sits at the end of the defining block and waits for the load it has just issued -- the first
    // build of this kernel ran one memory round trip per step.  Whole blocks of LR_PF steps carry no guard; the tail does.)
    auto fwd_step = [&](auto uc, int t) {
                constexpr int u = decltype(uc)::value;
#ifdef LR_T_NOLOAD     // (wrong-result timing switch, tuning builds only: the sweep without its ring loads)
                double gp = 1.0;
                const double gv = 0.5;
#else
                double gp = rg0_[u];
                const double gv = rg1_[u];
#endif
                if (FINAL) gp = fma((double)rh_[u], rw_[u], gp);          // (row 0: h = 0 and w = 0)
                {
                    const int tc = (t + LR_PF < H) ? t + LR_PF : H - 1;
                    rg0_[u] = gb[(size_t)tc * dim + i];
                    rg1_[u] = gb[(size_t)tc * dim + D + i];
                    if (FINAL) {
                        rh_[u] = jb[(size_t)tc * (D + 1) + i];
                        rw_[u] = wb[tc];
                    }
                }
                if (FINAL) {
                    for (int f = 1; f < F; ++f)          // further chained fields (rare: not prefetched)
                        gp = fma((double)jac[(((size_t)f * B + pc) * H + t) * (D + 1) + i], wdense[((size_t)f * B + pc) * H + t], gp);
                }
                const lr_d2* rn = tab2 + (size_t)((t + 1 < H) ? t + 1 : t) * (LR_REC / 2);
                const lr_d2 N0 = rn[0], N1 = rn[1], N2 = rn[2], N3 = rn[3];
                const double a0 = fma(-f10, r1, fma(-f00, r0, gp)), a1 = fma(-f11, r1, fma(-f01, r0, gv));
                r0 = a0; r1 = a1;
                const double z0 = fma(R0.y, r1, R0.x * r0), z1 = fma(R1.x, r1, R0.y * r0);
                f00 = R1.y; f01 = R2.x; f10 = R2.y; f11 = R3.x;
                R0 = N0; R1 = N1; R2 = N2; R3 = N3;
#ifndef LR_T_NOSTORE   // (wrong-result timing switch, tuning builds only)
                if (live) *reinterpret_cast<lr_d2*>(zr) = lr_d2{z0, z1};
#else
                if (live && t == 0) *reinterpret_cast<lr_d2*>(zr) = lr_d2{z0, z1};
#endif
                zr += zstep;
    };
    int tb = 0;
    for (; tb + LR_PF <= H; tb += LR_PF) lr_static_for<0, LR_PF>([&](auto uc) { fwd_step(uc, tb + decltype(uc)::value); });
    lr_static_for<0, LR_PF>([&](auto uc) {
        if (tb + decltype(uc)::value < H) fwd_step(uc, tb + decltype(uc)::value);         // (wave-uniform)
    });
    // ---- backward: records H - 1 .. 0 through the ring (FINAL: the lane's x beside them)
    lr_d2 q_[LR_PF];
    float xp_[LR_PF], xv_[LR_PF];
    const float* xb = x + (size_t)pc * H * dim;
#pragma unroll
    for (int u = 0; u < LR_PF; ++u) {
        const int t = (H - 1 - u >= 0) ? H - 1 - u : 0;
        q_[u] = *reinterpret_cast<const lr_d2*>(zbuf + ((size_t)t * NL + gl) * LR_ZREC);
        if (FINAL) { xp_[u] = xb[(size_t)t * dim + i]; xv_[u] = xb[(size_t)t * dim + D + i]; }
    }
    double y0 = 0.0, y1 = 0.0;            // (y_H = 0: the first step takes z_{H-1} as it is)
    R1 = tab2[(size_t)(H - 1) * (LR_REC / 2) + 1]; R2 = tab2[(size_t)(H - 1) * (LR_REC / 2) + 2]; R3 = tab2[(size_t)(H - 1) * (LR_REC / 2) + 3];
    auto bwd_step = [&](auto uc, int t) {
                constexpr int u = decltype(uc)::value;
                const lr_d2 zz = q_[u];
                float xp = 0.f, xv = 0.f;
                if (FINAL) { xp = xp_[u]; xv = xv_[u]; }
                {
                    const int tn = (t - LR_PF >= 0) ? t - LR_PF : 0;
                    q_[u] = *reinterpret_cast<const lr_d2*>(zbuf + ((size_t)tn * NL + gl) * LR_ZREC);
                    if (FINAL) { xp_[u] = xb[(size_t)tn * dim + i]; xv_[u] = xb[(size_t)tn * dim + D + i]; }
                }
                const lr_d2* rn = tab2 + (size_t)((t > 0) ? t - 1 : 0) * (LR_REC / 2);
                const lr_d2 N1 = rn[1], N2 = rn[2], N3 = rn[3];
                const double a0 = fma(-R2.x, y1, fma(-R1.y, y0, zz.x)), a1 = fma(-R3.x, y1, fma(-R2.y, y0, zz.y));
                y0 = a0; y1 = a1;
                R1 = N1; R2 = N2; R3 = N3;
                if (live) {
                    if (FINAL) {
                        x[((size_t)p * H + t) * dim + i] = (float)((double)xp + K.step * y0);
                        x[((size_t)p * H + t) * dim + D + i] = (float)((double)xv + K.step * y1);
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
                                                   double* __restrict__ wdense, float* __restrict__ costs_out, const int* __restrict__ ord,
                                                   int B, int H, int D, int F, int n_tiles_max, GpConst K) {
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the particle of this workgroup: largest systems first (gpmp2_lr_gradient, gpmp2_lr_sweep<false>)
    const int b = ord[B + blockIdx.x];
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
        // tile q of the lower triangle, row major: (I, J) with q = I (I + 1) / 2 + J
        auto tile_of = [](int q, int& I, int& J) {
            I = (int)((__builtin_sqrtf(8.f * (float)q + 1.f) - 1.f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= q) ++I;
            while (I * (I + 1) / 2 > q) --I;
            J = q - I * (I + 1) / 2;
        };
        auto fetch = [&](int q, double (&gv)[MPB_MAX_DOF]) {
            int I, J;
            tile_of(q < ntl ? q : ntl - 1, I, J);
            int a = 16 * I + r, cc = 16 * J + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }              // (diagonal tiles are stored full: mirror)
            const bool on = a < n;
            const double* Gst = G + (size_t)tact[on ? a : 0] * H + tact[on ? cc : 0];
#pragma unroll
            for (int j = 0; j < MPB_MAX_DOF; ++j) gv[j] = (j < D) ? Gst[(size_t)j * H * H] : 0.0;
        };
        auto emit = [&](int q, const double (&gv)[MPB_MAX_DOF]) {
            int I, J;
            tile_of(q, I, J);
            int a = 16 * I + r, cc = 16 * J + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }
            double m;
            if (a < n) {
                m = (a == cc) ? ikc : 0.0;
#pragma unroll
                for (int j = 0; j < MPB_MAX_DOF; ++j)
                    if (j < D) m = fma((double)hb[a * hs + j] * (double)hb[cc * hs + j], gv[j], m);
            } else if (a == n && cc < n) {
                m = wv[cc];
            } else {
                m = (a == cc) ? 1.0 : 0.0;
            }
            Tl[(size_t)lr_tile(I, J) * LR_TILE + lr_sw(r, c)] = m;
        };
        // THREE tiles' G entries (L2, ~1 us) in flight per thread (one tile ahead, the loop paid most of a round trip per tile: 28 us
        // of the 95 a particle with 116 active rows took)
        double g0[MPB_MAX_DOF], g1[MPB_MAX_DOF], g2[MPB_MAX_DOF];
        fetch(0, g0);
        fetch(1, g1);
        fetch(2, g2);
#ifdef LR_T_CAP_NOASM      // (wrong-result timing switch, tuning builds only: the first tile alone)
        const int ntl_run = 1;
#else
        const int ntl_run = ntl;
#endif
        for (int q = 0; q < ntl_run; q += 3) {
            emit(q, g0);
            fetch(q + 3, g0);
            if (q + 1 < ntl_run) { emit(q + 1, g1); fetch(q + 4, g1); }
            if (q + 2 < ntl_run) { emit(q + 2, g2); fetch(q + 5, g2); }
        }
    }
    __syncthreads();
    // ---- 4. tile Cholesky
    const int TC = (n + 15) >> 4;                                // tile columns with a pivot
    const int li = lane & 15, lk = lane >> 4;
#ifdef LR_T_CAP_NOCHOL     // (wrong-result timing switch, tuning builds only)
    for (int J = 0; J < 1; ++J) {
#else
    for (int J = 0; J < TC; ++J) {
#endif
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
    // ---- 5. L^T w = y (y = row n), one wave, a tile row at a time from the last: the 16 x 16 triangular system of the row's diagonal
    //         tile with its entries in registers (lane j < 16 holds L[16 K + k][16 K + j], k = 0 .. 15: sixteen steps of readlane, mul,
    //         fma), then y_j -= sum_k L[16 K + k][j] w_k for every j of the earlier tile rows (lanes = j, j + 64: sixteen independent
    //         fma per lane).  (Column by column over all n with the L reads inside the dependent loop: 21 us at n = 116.)
    if (wave == 0) {
        const int In = n >> 4, rn = n & 15;
        double y[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = lane + 64 * ps;
            y[ps] = (j < n) ? Tl[(size_t)lr_tile(In, j >> 4) * LR_TILE + lr_sw(rn, j & 15)] : 0.0;
        }
#ifdef LR_T_CAP_NOBACK     // (wrong-result timing switch, tuning builds only)
        for (int Kt = 0; Kt >= 0; --Kt) {
#else
        for (int Kt = TC - 1; Kt >= 0; --Kt) {
#endif
            const double* Dk = Tl + (size_t)lr_tile(Kt, Kt) * LR_TILE;
            const int jl = lane & 15;
            double Lc[16], di[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                Lc[k] = Dk[lr_sw(k, jl)];                                  // L[16 Kt + k][16 Kt + jl]
                di[k] = (16 * Kt + k < n) ? dinv[16 * Kt + k] : 0.0;      // (padding columns: w = 0)
            }
            // this tile row's y in lanes 0 .. 15 of a register of its own
            const int src = 16 * Kt + jl;
            double yt = (src < 64) ? __shfl(y[0], src & 63, 64) : __shfl(y[1], (src - 64) & 63, 64);
            double wk[16];
            lr_static_for<0, 16>([&](auto kc) {
                constexpr int k = 15 - decltype(kc)::value;
                wk[k] = lr_readlane(yt, k) * di[k];
                yt = fma(-Lc[k], wk[k], yt);                               // (lanes jl >= k: discarded)
            });
            if (lane < 16 && 16 * Kt + lane < n) {
                double wl = 0.0;
                lr_static_for<0, 16>([&](auto kc) { if (decltype(kc)::value == lane) wl = wk[decltype(kc)::value]; });
                wv[16 * Kt + lane] = wl;
            }
            // earlier tile rows: y_j -= sum_k L[16 Kt + k][j] w_k
            const double* Lrow = Tl + (size_t)lr_tile(Kt, 0) * LR_TILE;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int j = lane + 64 * ps;
                if (j < 16 * Kt) {
                    double acc = y[ps];
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = fma(-Lrow[(size_t)(j >> 4) * LR_TILE + lr_sw(k, j & 15)], wk[k], acc);
                    y[ps] = acc;
                }
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
    return (size_t)D * H * LR_REC + (size_t)D * H * H + (size_t)H * NL * LR_ZREC + (size_t)H * NL + (size_t)MPB_GP_MAX_FIELDS * B * H + (size_t)B + 64 +
           (size_t)B * H * 2 * D +         // ... and g_rest
           (size_t)B + 1;                  // ... and the size class of every particle + the launch order of the capacitance systems (2 B ints)
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
    double* grest = gpcost + B + 64;
    int* ord = reinterpret_cast<int*>(grest + (size_t)B * H * 2 * D);        // [0, B): size class of particle b; [B, 2 B): the particles, largest class first
    const size_t lds_chain = ((size_t)H * LR_REC + (size_t)H * 2 + (size_t)H * LR_KK) * sizeof(double);
    hipLaunchKernelGGL(gpmp2_chain_kernel, dim3(D, (H + LR_COLS - 1) / LR_COLS), dim3(LR_COLS), lds_chain, stream, diag_mean, rec, G, H, D, K);
    const int per = 64 / D;
    const size_t lds_sweep = (size_t)D * (H * LR_REC + 2) * sizeof(double);
    const dim3 gs((B + per - 1) / per);
    hipLaunchKernelGGL(gpmp2_lr_gradient, dim3(B), dim3(256), 0, stream, x, start, goal, jac, grest, gpcost, ord, B, H, D, n_fields, K);
    hipLaunchKernelGGL(gpmp2_lr_sweep<false>, dim3(gs.x + 1), dim3(64), lds_sweep, stream, x, grest, jac, wdense, rec, zbuf, upos, ord, B, H, D, n_fields, K);
    const int n_max = n_fields * (H - 1);
    const int trm = (n_max + 16) >> 4, ntm = (trm * (trm + 1)) >> 1;             // tiles of the largest system the shape allows
    const size_t lds = ((size_t)ntm * LR_TILE + 256) * sizeof(double) + (size_t)LR_NMAX * (D <= 8 ? 8 : MPB_MAX_DOF) * sizeof(float) + (256 + 16) * sizeof(int);
    hipLaunchKernelGGL(gpmp2_lr_cap, dim3(B), dim3(256), lds, stream, jac, upos, G, gpcost, wdense, costs_out, ord, B, H, D, n_fields, ntm, K);
    hipLaunchKernelGGL(gpmp2_lr_sweep<true>, gs, dim3(64), lds_sweep, stream, x, grest, jac, wdense, rec, zbuf, upos, ord, B, H, D, n_fields, K);
    return MPB_OK;
}
