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
// Six launches per iteration (DESIGN.md section 6 has their times and what the two earlier versions of the round -- chains walked
// inside the per-particle kernel, then block-Thomas sweeps with a lane per (particle, joint) -- taught):
//   gpmp2_pcr_setup        shared by all particles: the cyclic-reduction coefficients of the D chains (below) and the
//                          position-position entries G_i(s, t) of their inverses, D H^2 doubles (0.9 MB at C4: stays in L2);
//   gpmp2_lr_gradient      g_rest (priors + GP factors), joint major, its cost, and the particle's size class (rows with a hinge);
//   gpmp2_pcr_solve<false> u0 = A0^-1 g_rest, position and velocity rows, in place; one extra workgroup sorts the particles by class;
//   gpmp2_lr_cap           per particle, eight waves, largest systems first: the active rows compacted by ballot, M (n_a x n_a) from
//                          the G table in 16 x 16 tiles into LDS (six tiles of loads in flight), tile Cholesky (POTRF in registers
//                          by readlane, TRSM a lane per row, the trailing update on the fp64 MFMA), the right-hand side carried as
//                          row n_a, a blocked back substitution, w scattered to a dense per-waypoint vector;
//   gpmp2_pcr_solve<true>  A0^-1 V w for the particles WITH rows, step = step (u0 + that), joint major;
//   gpmp2_lr_apply         x += step, turned row-wise through LDS (particles without rows: step u0).
// Against the block elimination of rounds 1-5 (mpb_gpmp2.hip: one 16 x 16 fp64 Gauss-Jordan inverse per waypoint and particle, 127
// of them in a row; 352 MB of W_t records written and read back): ~1.2 MFLOP and ~2 600 dependent pivot steps per particle become
// ~0.2 MFLOP, none of it sequential, and 2 n_a short steps.  The block kernel stays for what this form does not take: chained
// fields whose rows exceed the LDS tile (F (H - 1) > 127), H > 128.
#include <stdlib.h>
#include <type_traits>

#include "mpb_common.h"
#include "mpb_gpmp2.h"

#define LR_NMAX 127                    // active rows a particle may have (with the right-hand side 128 rows: 36 tiles of 16 x 16 doubles, 72 KB of LDS)

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
// A0^-1 by PARALLEL CYCLIC REDUCTION (round 6, third version of the shared part).  The block-Thomas sweeps of the second version
// walked a chain of H 2 x 2 blocks one step after the other -- 2 H dependent steps per solve, a lane per (particle, joint), 228
// waves on 1 024 SIMDs: 31 + 47 us per iteration at C4 whatever the batch, and 31 us more for the factors and the inverse's entries
// (one lane per joint).  A0 is the same for every particle, so the MATRIX half of a cyclic reduction is shared: level l (stride
// h = 2^l) replaces equation t by  eq_t + alpha_t eq_{t-h} + gamma_t eq_{t+h}  with
//     alpha_t = -Lo_t Dg_{t-h}^-1,   gamma_t = -Up_t Dg_{t+h}^-1,
//     Dg'_t = Dg_t + alpha_t Up_{t-h} + gamma_t Lo_{t+h},   Lo'_t = alpha_t Lo_{t-h},   Up'_t = gamma_t Up_{t+h}
// (Lo / Up: the blocks that couple t to t - h / t + h; level 0: U^T / U), and after ceil(log2 H) levels every equation stands alone:
// y_t = Dg_t^-1 r_t.  gpmp2_pcr_setup computes alpha, gamma of every level and the final inverses once per joint (a thread per
// waypoint, neighbours through LDS: 7 levels at H = 128); a SOLVE is then, per level, r_t += alpha_t r_{t-h} + gamma_t r_{t+h} for
// all t at once -- a wave per chain, lane = waypoints t and t + 64, the right-hand sides exchanged through LDS, PCR_NP chains per
// wave sharing every coefficient read.  3.5 x the arithmetic of a Thomas sweep, none of it sequential.  Stability: every level is
// a block Gaussian elimination of an SPD matrix in a symmetric order.
// ------------------------------------------------------------------------------------------------
#define LR_HMAX 128                    // waypoints (mpb_gpmp2_lr_ok: n_fields (H - 1) <= LR_NMAX)
#define LR_ORD 9                       // size classes of the capacitance systems' launch order: 0 rows, 1-16, ..., 113-128 (largest first)
// (measured at C4, waves x chains: 8 x 4 -- 23.1 / 19.7 us for the two solves --, 8 x 2 -- 24.0 / 18.9 --, 16 x 2 -- 21.4 / 15.2: the solves are
// bound by the latency of a level's LDS round trips, not by LDS bandwidth -- dropping the last level's exchange changed nothing --, and
// four waves per SIMD hide more of it than two)
#ifndef PCR_NP
#define PCR_NP 2                       // chains a wave solves at once (they share every coefficient read)
#endif
#ifndef PCR_WAVES
#define PCR_WAVES 16
#endif
#define PCR_THREADS (64 * PCR_WAVES)
static inline int pcr_levels(int H) { int L = 0; for (int h = 1; h < H; h <<= 1) ++L; return L; }
// coefficient table of a joint, 16-byte entries: [level][q][t], q = 0: (alpha00, alpha01), 1: (alpha10, alpha11), 2: (gamma00, gamma01),
// 3: (gamma10, gamma11) -- a lane's reads of consecutive t fall on consecutive banks --, then [2][t]: the rows of Dg^-1 of the last level
static inline size_t pcr_coef_entries(int H) { return (size_t)(4 * pcr_levels(H) + 2) * H; }

// PCR_NP right-hand sides through the L levels.  r[k][e]: chain k at waypoint lane (e = 0) / lane + 64 (e = 1); on return the solution.
// rb: this wave's PCR_NP x H exchange rows (one buffer: the LDS instructions of a wave execute in order, the reads of a level are all
// issued before its successor's writes).
__device__ __forceinline__ void pcr_solve(const lr_d2* __restrict__ coef, lr_d2* __restrict__ rb, int H, int L, int lane, lr_d2 (&r)[PCR_NP][2]) {
    const int tt[2] = {lane, lane + 64};
    const bool vv[2] = {lane < H, lane + 64 < H};
    for (int l = 0, h = 1; l < L; ++l, h <<= 1) {
        const lr_d2* cl = coef + (size_t)(4 * l) * H;
        if (h == 64) {
            // the last level of 64 < H <= 128: waypoint t's partner t + 64 is the lane's own other element (t - 64 and t + 128 do not
            // exist: their coefficients are zero) -- no exchange through LDS (a seventh of the solves' LDS traffic, which bounds them)
            const int t0 = vv[0] ? tt[0] : 0, t1 = vv[1] ? tt[1] : 0;
            const lr_d2 c0 = cl[2 * H + t0], c1 = cl[3 * H + t0], a0 = cl[t1], a1 = cl[H + t1];
#pragma unroll
            for (int k = 0; k < PCR_NP; ++k) {
                const lr_d2 ra = r[k][0], rq = vv[1] ? r[k][1] : lr_d2{0.0, 0.0};
                r[k][0] = lr_d2{fma(c0.y, rq.y, fma(c0.x, rq.x, ra.x)), fma(c1.y, rq.y, fma(c1.x, rq.x, ra.y))};
                r[k][1] = lr_d2{fma(a0.y, ra.y, fma(a0.x, ra.x, rq.x)), fma(a1.y, ra.y, fma(a1.x, ra.x, rq.y))};
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            if (vv[0]) rb[k * H + tt[0]] = r[k][0];
            if (vv[1]) rb[k * H + tt[1]] = r[k][1];
        }
        lr_wave_sync();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = vv[e] ? tt[e] : 0;
            const lr_d2 a0 = cl[t], a1 = cl[H + t], c0 = cl[2 * H + t], c1 = cl[3 * H + t];
            const int tm = (t - h >= 0) ? t - h : t, tp = (t + h < H) ? t + h : t;       // (out of range: alpha / gamma are zero)
#pragma unroll
            for (int k = 0; k < PCR_NP; ++k) {
                const lr_d2 rm = rb[k * H + tm], rp = rb[k * H + tp];
                const double nx = fma(c0.y, rp.y, fma(c0.x, rp.x, fma(a0.y, rm.y, fma(a0.x, rm.x, r[k][e].x))));
                const double ny = fma(c1.y, rp.y, fma(c1.x, rp.x, fma(a1.y, rm.y, fma(a1.x, rm.x, r[k][e].y))));
                r[k][e] = lr_d2{nx, ny};
            }
        }
        lr_wave_sync();
    }
    const lr_d2* dl = coef + (size_t)(4 * L) * H;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int t = vv[e] ? tt[e] : 0;
        const lr_d2 i0 = dl[t], i1 = dl[H + t];
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            const lr_d2 v = r[k][e];
            r[k][e] = lr_d2{fma(i0.y, v.y, i0.x * v.x), fma(i1.y, v.y, i1.x * v.x)};
        }
    }
}

// a joint's table from global memory into LDS, eight 16-byte loads in flight per thread
__device__ __forceinline__ void pcr_stage(const lr_d2* __restrict__ src, lr_d2* __restrict__ dst, int n, int tid) {
    for (int e0 = 0; e0 < n; e0 += PCR_THREADS * 8) {
        lr_d2 tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + tid + PCR_THREADS * u;
            tmp[u] = src[e < n ? e : n - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + tid + PCR_THREADS * u;
            if (e < n) dst[e] = tmp[u];
        }
    }
}

struct Pcr22 { double a, b, c, d; };          // [[a, b], [c, d]]
__device__ __forceinline__ Pcr22 pcr_inv(const Pcr22& m) {
    const double id = lr_rcp(fma(m.a, m.d, -m.b * m.c));
    return Pcr22{m.d * id, -m.b * id, -m.c * id, m.a * id};
}
__device__ __forceinline__ Pcr22 pcr_mul(const Pcr22& x, const Pcr22& y) {
    return Pcr22{fma(x.b, y.c, x.a * y.a), fma(x.b, y.d, x.a * y.b), fma(x.d, y.c, x.c * y.a), fma(x.d, y.d, x.c * y.b)};
}

// ------------------------------------------------------------------------------------------------
// shared: the reduction coefficients of the D chains (to coef_g, by the blocks with blockIdx.y = 0) and the position-position entries
// G_i(s, t) of their inverses (D H^2 doubles, 0.9 MB at C4: stays in L2) -- column c is the solve of a unit position entry at c, the
// columns split over the gridDim.y blocks of a joint, PCR_NP per wave and pass; written as ROW c (G is symmetric: a lane per t
// then stores consecutive words).  Every block computes its joint's coefficients itself (7 levels of a thread per waypoint: cheaper
// than a launch of its own).  grid = (D, NY), block = PCR_THREADS; LDS: the table | 12 H doubles of exchange | the waves' rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PCR_THREADS) void gpmp2_pcr_setup(const double* __restrict__ diag_mean, double* __restrict__ coef_g,
                                                               double* __restrict__ G, int H, int D, int L, GpConst K) {
    extern __shared__ double lds[];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_coef = (4 * L + 2) * H;
    lr_d2* coef = reinterpret_cast<lr_d2*>(lds);
    double* xch = lds + 2 * (size_t)n_coef;                                  // H x 12: Dg, Lo, Up of every waypoint at the current level
    lr_d2* rb = reinterpret_cast<lr_d2*>(xch + 12 * (size_t)H) + (size_t)wave * PCR_NP * H;
    const LrCoef C = lr_coef(K);
    const int dim = 2 * D, t = tid;
    Pcr22 Dg{0, 0, 0, 0}, Lo{0, 0, 0, 0}, Up{0, 0, 0, 0};
    if (t < H) {
        // D_t = (GP blocks) + damping + start / goal prior (gpmp2.py:355-368; the damping is delta x the batch mean of the diagonal: Q9)
        const double first = (t == 0) ? 1.0 : 0.0, last = (t == H - 1) ? 1.0 : 0.0;
        const double dp = K.trust ? K.delta * diag_mean[(size_t)t * dim + i] : K.delta;
        const double dv = K.trust ? K.delta * diag_mean[(size_t)t * dim + D + i] : K.delta;
        Dg.a = (1.0 - last) * C.p00 + (1.0 - first) * C.a + dp + first * K.ks + last * K.kg;
        Dg.b = Dg.c = (1.0 - last) * C.p01 + (1.0 - first) * C.bq;
        Dg.d = (1.0 - last) * C.p11 + (1.0 - first) * C.cq + dv + first * K.ks + last * K.kg;
        if (t > 0) Lo = Pcr22{C.u00, C.u10, C.u01, C.u11};                  // block (t, t - 1) = U^T
        if (t < H - 1) Up = Pcr22{C.u00, C.u01, C.u10, C.u11};              // block (t, t + 1) = U
    }
    for (int l = 0, h = 1; l < L; ++l, h <<= 1) {
        if (t < H) {
            double* w = xch + 12 * (size_t)t;
            w[0] = Dg.a; w[1] = Dg.b; w[2] = Dg.c; w[3] = Dg.d;
            w[4] = Lo.a; w[5] = Lo.b; w[6] = Lo.c; w[7] = Lo.d;
            w[8] = Up.a; w[9] = Up.b; w[10] = Up.c; w[11] = Up.d;
        }
        __syncthreads();
        if (t < H) {
            Pcr22 al{0, 0, 0, 0}, ga{0, 0, 0, 0};
            Pcr22 nDg = Dg, nLo{0, 0, 0, 0}, nUp{0, 0, 0, 0};
            if (t - h >= 0) {
                const double* m = xch + 12 * (size_t)(t - h);
                const Pcr22 Dm{m[0], m[1], m[2], m[3]}, Lm{m[4], m[5], m[6], m[7]}, Um{m[8], m[9], m[10], m[11]};
                const Pcr22 x = pcr_mul(Lo, pcr_inv(Dm));
                al = Pcr22{-x.a, -x.b, -x.c, -x.d};
                const Pcr22 du = pcr_mul(al, Um);
                nDg.a += du.a; nDg.b += du.b; nDg.c += du.c; nDg.d += du.d;
                nLo = pcr_mul(al, Lm);
            }
            if (t + h < H) {
                const double* q = xch + 12 * (size_t)(t + h);
                const Pcr22 Dp{q[0], q[1], q[2], q[3]}, Lp{q[4], q[5], q[6], q[7]}, Uq{q[8], q[9], q[10], q[11]};
                const Pcr22 x = pcr_mul(Up, pcr_inv(Dp));
                ga = Pcr22{-x.a, -x.b, -x.c, -x.d};
                const Pcr22 dl = pcr_mul(ga, Lp);
                nDg.a += dl.a; nDg.b += dl.b; nDg.c += dl.c; nDg.d += dl.d;
                nUp = pcr_mul(ga, Uq);
            }
            lr_d2* cl = coef + (size_t)(4 * l) * H;
            cl[t] = lr_d2{al.a, al.b}; cl[H + t] = lr_d2{al.c, al.d}; cl[2 * H + t] = lr_d2{ga.a, ga.b}; cl[3 * H + t] = lr_d2{ga.c, ga.d};
            // (the reduced diagonal block is symmetric up to rounding: kept so)
            const double off = 0.5 * (nDg.b + nDg.c);
            Dg = Pcr22{nDg.a, off, off, nDg.d}; Lo = nLo; Up = nUp;
        }
        __syncthreads();
    }
    if (t < H) {
        const Pcr22 inv = pcr_inv(Dg);
        coef[(size_t)(4 * L) * H + t] = lr_d2{inv.a, inv.b};
        coef[(size_t)(4 * L + 1) * H + t] = lr_d2{inv.c, inv.d};
    }
    __syncthreads();
    if (blockIdx.y == 0) {
        lr_d2* dst = reinterpret_cast<lr_d2*>(coef_g) + (size_t)i * n_coef;
        for (int e = tid; e < n_coef; e += PCR_THREADS) dst[e] = coef[e];
    }
    // the columns of this block
    const int cpb = (H + gridDim.y - 1) / gridDim.y;
    const int c_lo = blockIdx.y * cpb, c_hi = min(H, c_lo + cpb);
    double* Gi = G + (size_t)i * H * H;
    for (int c0 = c_lo + wave * PCR_NP; c0 < c_hi; c0 += PCR_WAVES * PCR_NP) {
        lr_d2 r[PCR_NP][2];
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            r[k][0] = lr_d2{(lane == c0 + k) ? 1.0 : 0.0, 0.0};
            r[k][1] = lr_d2{(lane + 64 == c0 + k) ? 1.0 : 0.0, 0.0};
        }
        pcr_solve(coef, rb, H, L, lane, r);
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            if (c0 + k < c_hi) {
                if (lane < H) Gi[(size_t)(c0 + k) * H + lane] = r[k][0].x;
                if (lane + 64 < H) Gi[(size_t)(c0 + k) * H + lane + 64] = r[k][1].x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// g_rest: the gradient without its collision part (start / goal priors and the GP factors, gpmp2.py:355-368 with the rows of
// cost_functions.py:291-314, :538-554), fp64, JOINT MAJOR -- (D, B, H) pairs (position, velocity): the right-hand side of a chain
// is 2 KB in a row for gpmp2_pcr_solve --, the cost b^T K b of those factors per particle (gpmp2.py:493-495), and the particle's size
// class for the launch order of the capacitance systems.  A workgroup per particle.
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
    // solve that follows sorts the particles by class, largest first (a counting sort in LDS; one atomic per particle on a
    // counter per class in global memory cost this kernel 19 us: 2 048 returning atomics on nine addresses), and gpmp2_lr_cap
    // takes them in that order.  (Scheduling only: the class is an estimate, nothing else reads it.)
    int n_est = 0;
    const int dim = 2 * D;
    const LrCoef C = lr_coef(K);
    const double dt = K.dt;
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
        const int j = e / H, t = e - j * H;              // consecutive threads = consecutive waypoints of a joint: the stores below are whole lines
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
        reinterpret_cast<lr_d2*>(g)[((size_t)j * B + b) * H + t] = lr_d2{gp, gv};
    }
    cost = wave_sum_f64(cost);
    if (lane == 0) { red_c[wave] = cost; red_n[wave] = n_est; }
    __syncthreads();
    if (tid == 0) {
        gpcost[b] = (red_c[0] + red_c[1]) + (red_c[2] + red_c[3]);
        const int n = red_n[0] + red_n[1] + red_n[2] + red_n[3];
        ord[b] = LR_ORD - 1 - min(LR_ORD - 1, (n + 15) >> 4);          // size class, 0 = largest (sorted by gpmp2_pcr_solve<false>'s extra workgroup)
    }
}

// ------------------------------------------------------------------------------------------------
// A0^-1 applied to a gradient.  grid = (D, NG [+ 1]), block = PCR_THREADS: a block stages ITS joint's coefficient table (61 KB at
// H = 128) and its waves take the particles of group blockIdx.y, PCR_NP at a time.  g: joint major, (D, B, H) pairs (position,
// velocity) -- a chain's right-hand side is 2 KB in a row (gpmp2_lr_gradient writes it so).
// FINAL = false: the gradient is g_rest; writes the position rows of u0 = A0^-1 g_rest, joint major (upos[j][b][t]).  Its extra
// row of blocks (blockIdx.y = NG; block 0 of it works) sorts the particles by size class for gpmp2_lr_cap.
// FINAL = true: the gradient is g_rest + V w (w: dense per field and waypoint, zero off the active rows); x += step * dtheta
// (gpmp2.py:326-331).
// ------------------------------------------------------------------------------------------------
template <bool FINAL>
__global__ __launch_bounds__(PCR_THREADS) void gpmp2_pcr_solve(double* __restrict__ g, const float* __restrict__ jac,
                                                               const double* __restrict__ wdense, const double* __restrict__ coef_g,
                                                               double* __restrict__ dth, int* __restrict__ ord, int B, int H, int D, int F,
                                                               int L, int NG, GpConst K) {
    extern __shared__ double lds[];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!FINAL && (int)blockIdx.y == NG) {
        // the extra row: the particles sorted by size class (gpmp2_lr_gradient), largest first -- the order gpmp2_lr_cap takes them
        // in.  A counting sort: class totals and positions by LDS atomics.
        if (j != 0) return;
        int* cnt = reinterpret_cast<int*>(lds);            // [0, LR_ORD): totals, then running offsets
        if (tid < LR_ORD) cnt[tid] = 0;
        __syncthreads();
        for (int b0 = 0; b0 < B; b0 += PCR_THREADS)
            if (b0 + tid < B) atomicAdd(&cnt[ord[b0 + tid]], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int c = 0; c < LR_ORD; ++c) { const int n = cnt[c]; cnt[c] = run; run += n; }
            ord[2 * B] = cnt[LR_ORD - 1];          // particles with collision rows: the first so many of the list (the last class has none)
        }
        __syncthreads();
        for (int b0 = 0; b0 < B; b0 += PCR_THREADS)
            if (b0 + tid < B) ord[B + atomicAdd(&cnt[ord[b0 + tid]], 1)] = b0 + tid;
        return;
    }
    // FINAL: only the particles WITH collision rows (the head of the sorted list): the others' step is u0 itself (gpmp2_lr_apply)
    const int n_part = FINAL ? ord[2 * B] : B;
    const int per_block = (n_part + NG - 1) / NG;
    const int p_lo = blockIdx.y * per_block, p_hi = min(n_part, p_lo + per_block);
    if (p_lo >= p_hi) return;                                                 // (block-uniform, before any barrier)
    const int n_coef = (4 * L + 2) * H;
    lr_d2* coef = reinterpret_cast<lr_d2*>(lds);
    lr_d2* rb = coef + n_coef + (size_t)wave * PCR_NP * H;
    pcr_stage(reinterpret_cast<const lr_d2*>(coef_g) + (size_t)j * n_coef, coef, n_coef, tid);
    __syncthreads();
    const int tt[2] = {lane, lane + 64};
    const bool vv[2] = {lane < H, lane + 64 < H};
    lr_d2* g2 = reinterpret_cast<lr_d2*>(g);
    for (int p0 = p_lo + wave * PCR_NP; p0 < p_hi; p0 += PCR_WAVES * PCR_NP) {
        lr_d2 r[PCR_NP][2];
        int pk[PCR_NP];
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            const int a = (p0 + k < p_hi) ? p0 + k : p_hi - 1;               // (beyond the group: a valid chain, not stored)
            const int p = FINAL ? ord[B + a] : a;
            pk[k] = p;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                lr_d2 v = lr_d2{0.0, 0.0};
                if (vv[e]) {
                    if (FINAL) {
                        for (int f = 0; f < F; ++f) {
                            const double w = wdense[((size_t)f * B + p) * H + tt[e]];
                            if (w != 0.0) v.x = fma((double)jac[(((size_t)f * B + p) * H + tt[e]) * (D + 1) + j], w, v.x);     // (row 0: w = 0)
                        }
                    } else {
                        v = g2[((size_t)j * B + p) * H + tt[e]];
                    }
                }
                r[k][e] = v;
            }
        }
        pcr_solve(coef, rb, H, L, lane, r);
#pragma unroll
        for (int k = 0; k < PCR_NP; ++k) {
            if (p0 + k < p_hi) {
                const int p = pk[k];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (vv[e]) {
                        lr_d2* slot = g2 + ((size_t)j * B + p) * H + tt[e];
                        if (FINAL) {
                            // the step of (particle, joint) = step (u0 + A0^-1 V w), joint major over the rows gpmp2_lr_cap's w went through:
                            // gpmp2_lr_apply adds it to x row by row (written to x from here -- 4 bytes every 8 D -- the kernel took 66 us
                            // instead of 21)
                            const lr_d2 u0 = *slot;
                            reinterpret_cast<float2*>(dth)[((size_t)j * B + p) * H + tt[e]] =
                                make_float2((float)(K.step * (u0.x + r[k][e].x)), (float)(K.step * (u0.y + r[k][e].y)));
                        } else {
                            *slot = r[k][e];                                   // u0 over g_rest (position AND velocity rows)
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// x += step * dtheta (gpmp2.py:326-331): the steps are joint major ((D, B, H) pairs of floats, gpmp2_pcr_solve<true>; for a particle
// without collision rows: step * u0, gpmp2_pcr_solve<false>'s pairs), x is (B, H, 2 D): a workgroup per particle turns its rows through
// LDS, so both sides move in whole lines.
// ------------------------------------------------------------------------------------------------
template <int NTHR>
__device__ __forceinline__ void lr_apply_rows(float* __restrict__ x, const float2* __restrict__ dth, const double* __restrict__ u0, bool has_rows,
                                              int b, int B, int H, int D, double step, float* __restrict__ xs, int tid) {
    const int dim = 2 * D, nx = H * dim;
    float* xg = x + (size_t)b * nx;
    constexpr int NT = (LR_HMAX * 2 * MPB_MAX_DOF + NTHR - 1) / NTHR;
    float tmp[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int e = tid + NTHR * u;
        tmp[u] = (e < nx) ? xg[e] : 0.f;
    }
    constexpr int ND = (LR_HMAX * MPB_MAX_DOF + NTHR - 1) / NTHR;
    float2 dv[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int e = tid + NTHR * u, j = e / H, t = e - j * H;
        dv[u] = make_float2(0.f, 0.f);
        if (e < H * D) {
            if (has_rows) {
                dv[u] = dth[((size_t)j * B + b) * H + t];
            } else {            // no collision rows: w = 0 and the step is u0's (gpmp2_pcr_solve<true> skips the particle)
                const lr_d2 v = reinterpret_cast<const lr_d2*>(u0)[((size_t)j * B + b) * H + t];
                dv[u] = make_float2((float)(step * v.x), (float)(step * v.y));
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int e = tid + NTHR * u;
        if (e < nx) xs[e] = tmp[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ND; ++u) {
        const int e = tid + NTHR * u, j = e / H, t = e - j * H;
        if (e < H * D) {
            xs[t * dim + j] += dv[u].x;
            xs[t * dim + D + j] += dv[u].y;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int e = tid + NTHR * u;
        if (e < nx) xg[e] = xs[e];
    }
}
// (the particles WITH collision rows: the others were stepped by their capacitance workgroup, which had nothing else to do)
__global__ __launch_bounds__(256) void gpmp2_lr_apply(float* __restrict__ x, const float2* __restrict__ dth, const double* __restrict__ u0,
                                                      const int* __restrict__ ord, int B, int H, int D, double step) {
    __shared__ float xs[LR_HMAX * 2 * MPB_MAX_DOF];
    const int a = blockIdx.x;
    if (a >= ord[2 * B]) return;                               // (block-uniform: the head of the sorted list)
    lr_apply_rows<256>(x, dth, u0, true, ord[B + a], B, H, D, step, xs, threadIdx.x);
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
#ifndef LR_ASM_RING
#define LR_ASM_RING 6                                                        // tiles of M whose G entries are in flight
#endif
#define CAP_THREADS 512                                                      // eight waves: the assembly, the TRSM rows and the trailing update spread over all of them
#define CAP_SLOTS (CAP_THREADS / 256)                                        // tiles of M assembled per pass
#define LR_TILE 256                                                          // doubles per tile
__device__ __forceinline__ int lr_tile(int I, int J) { return ((I * (I + 1)) >> 1) + J; }            // J <= I
__device__ __forceinline__ int lr_sw(int r, int c) { return (r << 4) + (c ^ r); }
__device__ __forceinline__ double lr_rsqrt(double x) {                      // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(fma(-x * y, y, 1.0), 0.5, 1.0);
    y = y * fma(fma(-x * y, y, 1.0), 0.5, 1.0);
    return y;
}

template <int DMAX>        // DMAX: 8 or MPB_MAX_DOF -- the joints a row of h is unrolled over (D <= DMAX)
#ifndef CAP_WPE
#define CAP_WPE 4
#endif
__global__ __launch_bounds__(CAP_THREADS, CAP_WPE) void gpmp2_lr_cap(const float* __restrict__ jac, const double* __restrict__ upos,
                                                   const double* __restrict__ G, const double* __restrict__ gpcost,
                                                   double* __restrict__ wdense, float* __restrict__ costs_out, const int* __restrict__ ord,
                                                   float* __restrict__ x, int B, int H, int D, int F, int n_tiles_max, GpConst K) {
    extern __shared__ double lds[];
    // LDS: [ tiles | rhs / y / w (128) | 1 / l_kk (128) | h rows of the active set (LR_NMAX x 8 fp32) | waypoint and field of every
    //        active row (2 x 128 ints) | scratch ints ]
    double* Tl = lds;
    double* wv = lds + (size_t)n_tiles_max * LR_TILE;
    double* dinv = wv + 128;
    float* hb = reinterpret_cast<float*>(dinv + 128);
    constexpr int hs = DMAX;                  // floats per row of h (the launcher sizes the LDS for it)
    int* tact = reinterpret_cast<int*>(hb + LR_NMAX * hs);
    int* fact = tact + 128;
    int* cnt = fact + 128;                                       // [0 .. 2 F): active rows of (field, 64-waypoint chunk)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the particle of this workgroup: largest systems first (gpmp2_lr_gradient, gpmp2_pcr_solve<false>)
    const int b = ord[B + blockIdx.x];
#ifdef LR_T_CLK
    unsigned long long cclk_[7], cph_[3] = {0, 0, 0}; cclk_[0] = wall_clock64();
#define CAP_CLK(k) cclk_[k] = wall_clock64()
#else
#define CAP_CLK(k)
#endif
    // a particle of the last size class has no hinge anywhere (every c_t is zero, so every h_t is): its collision cost is zero, w = 0,
    // and its step is u0's.  This workgroup takes that step, x += step u0, and is done -- no compaction (its jac rows are not even
    // read), and gpmp2_lr_apply is left with the particles that have rows.  (More than half of C4's particles: with the compaction
    // in front of the test the kernel did not get below 41 us however small the systems had become.)
    if (ord[b] == LR_ORD - 1) {                                  // (block-uniform)
        if (costs_out != nullptr && tid == 0) costs_out[b] = (float)gpcost[b];
        lr_apply_rows<CAP_THREADS>(x, nullptr, upos, false, b, B, H, D, K.step, reinterpret_cast<float*>(lds), tid);
        return;
    }
    // ---- 1. the active collision rows, compacted (field major, then waypoint); the collision part of the cost.  Chunk (f, base) is
    //         examined by wave (2 f + base / 64) mod 4: counts first, then positions
    const int nchunk = F * ((H + 63) >> 6);
    double cost = 0.0;
    float hrow[DMAX + 1];
    bool act = false;
    int myt = 0, myf = 0;
    // (F (H - 1) <= 127: at most 8 chunks, two per wave; a wave keeps the rows of its SECOND chunk in the second set below)
    float hrow2[DMAX + 1];
    bool act2 = false;
    int myt2 = 0, myf2 = 0;
    for (int ch = wave; ch < nchunk && wave < 4; ch += 4) {      // (waves 0-3: the compaction is four waves' work)
        const int f = ch / ((H + 63) >> 6), base = (ch - f * ((H + 63) >> 6)) << 6;
        const int t = base + lane;
        const float* jb = jac + ((size_t)f * B + b) * H * (D + 1);
        float hr[DMAX + 1];
        bool a = false;
#pragma unroll
        for (int j = 0; j <= DMAX; ++j) {
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
            for (int j = 0; j <= DMAX; ++j) hrow[j] = hr[j];
            act = a; myt = t; myf = f;
        } else {
#pragma unroll
            for (int j = 0; j <= DMAX; ++j) hrow2[j] = hr[j];
            act2 = a; myt2 = t; myf2 = f;
        }
    }
    // the cost: the four waves' partial sums through LDS
    cost = wave_sum_f64(cost);
    if (lane == 0 && wave < 4) dinv[wave] = cost;
    __syncthreads();
    if (costs_out != nullptr && tid == 0) costs_out[b] = (float)(((dinv[0] + dinv[1]) + (dinv[2] + dinv[3])) + gpcost[b]);
    int n = 0;
    for (int ch = 0; ch < nchunk; ++ch) n += cnt[ch];
    if (n == 0) return;                                          // (block-uniform; hinges without a gradient: on gpmp2_pcr_solve<true>'s list, stepped there)
    for (int ps = 0; ps < 2; ++ps) {
        const int ch = wave + 4 * ps;
        if (ch < nchunk && wave < 4) {
            int off = 0;
            for (int e = 0; e < ch; ++e) off += cnt[e];
            const bool a = ps ? act2 : act;
            const unsigned long long m = __ballot(a);
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (a) {
                tact[pos] = ps ? myt2 : myt;
                fact[pos] = ps ? myf2 : myf;
#pragma unroll
                for (int j = 0; j < DMAX; ++j)
                    if (j < hs) hb[pos * hs + j] = (j < D) ? (ps ? hrow2[j] : hrow[j]) : 0.f;
                wv[pos] = (double)(ps ? hrow2[D] : hrow[D]);     // c of the row (until the right-hand side takes the slot)
            }
        }
    }
    __syncthreads();
    CAP_CLK(1);
    // ---- 2. right-hand side: c_a - h_a . u0[position rows of t_a]
    if (tid < n) {
        double sacc = wv[tid];
        const lr_d2* up = reinterpret_cast<const lr_d2*>(upos) + (size_t)b * H + tact[tid];        // joint major pairs (position, velocity): u0[j][b][t]
        for (int j = 0; j < D; ++j) sacc -= (double)hb[tid * hs + j] * up[(size_t)j * B * H].x;
        wv[tid] = sacc;
    }
    __syncthreads();
    CAP_CLK(2);
    // ---- 3. the tiles: M[a][c] = [a == c] / kc + sum_i h_a,i h_c,i G_i(t_a, t_c); row n = the right-hand side; identity beyond.
    //         A tile per pass of the 256 threads, the NEXT tile's G entries (L2) fetched before the current tile's arithmetic
    const int TR = (n + 16) >> 4;                                // tile rows that hold rows 0 .. n
    const int ntl = (TR * (TR + 1)) >> 1;
    {
        const double ikc = 1.0 / K.kc;
        const int r = (tid & 255) >> 4, c = tid & 15, slot = tid >> 8;          // CAP_SLOTS tiles per pass: tile q goes to slot q mod CAP_SLOTS
        // tile q of the lower triangle, row major: (I, J) with q = I (I + 1) / 2 + J.  The ring walks q in order: the fetches' and the
        // emits' (I, J) are stepped, not solved from q (a square root and two correction loops per call were ~80 of a tile's ~400
        // instructions)
        int fI = 0, fJ = 0, fq = 0, eI = 0, eJ = 0;
        auto step = [](int& I, int& J) { if (++J > I) { ++I; J = 0; } };
        for (int k = 0; k < slot; ++k) {                          // (this slot's first tile; with fewer tiles than slots: the last one)
            if (fq + 1 < ntl) { ++fq; step(fI, fJ); }
            step(eI, eJ);
        }
        auto fetch = [&](double (&gv)[DMAX]) {             // tile fq (beyond the last tile: the last one again)
            int a = 16 * fI + r, cc = 16 * fJ + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }              // (diagonal tiles are stored full: mirror)
            const bool on = a < n;
            const double* Gst = G + (size_t)tact[on ? a : 0] * H + tact[on ? cc : 0];
#pragma unroll
            for (int j = 0; j < DMAX; ++j) gv[j] = (j < D) ? Gst[(size_t)j * H * H] : 0.0;
#pragma unroll
            for (int k = 0; k < CAP_SLOTS; ++k)
                if (fq + 1 < ntl) { ++fq; step(fI, fJ); }
        };
        auto emit = [&](const double (&gv)[DMAX]) {        // the next tile in order
            const int I = eI, J = eJ;
#pragma unroll
            for (int k = 0; k < CAP_SLOTS; ++k) step(eI, eJ);
            int a = 16 * I + r, cc = 16 * J + c;
            if (cc > a) { const int t_ = a; a = cc; cc = t_; }
            double m;
            if (a < n) {
                m = (a == cc) ? ikc : 0.0;
#pragma unroll
                for (int j = 0; j < DMAX; ++j)
                    if (j < D) m = fma((double)hb[a * hs + j] * (double)hb[cc * hs + j], gv[j], m);
            } else if (a == n && cc < n) {
                m = wv[cc];
            } else {
                m = (a == cc) ? 1.0 : 0.0;
            }
            Tl[(size_t)lr_tile(I, J) * LR_TILE + lr_sw(r, c)] = m;
        };
        // RING tiles' G entries (L2, ~1 us) in flight per thread.  (One tile ahead, the loop paid most of a round trip per tile:
        // 28 us of the 95 a particle with 116 active rows took.  And a ring slot must be refilled OUTSIDE any conditional -- a fetch
        // under `if (q + 1 < ntl)` reaches the next trip through a phi whose copy waits for the load it has just issued: the
        // three-slot ring of the second version ran one tile ahead in effect.  A fetch beyond the last tile re-reads the last one.)
        constexpr int RING = (DMAX <= 8) ? LR_ASM_RING : LR_ASM_RING / 2;          // (twelve joints: half the ring, or the tile's registers spill)
        double gr[RING][DMAX];
        lr_static_for<0, RING>([&](auto uc) { fetch(gr[decltype(uc)::value]); });
#ifdef LR_T_CAP_NOASM      // (wrong-result timing switch, tuning builds only: the first tile alone)
        const int ntl_run = 1;
#else
        const int ntl_run = ntl;
#endif
        for (int q = slot; q < ntl_run; q += CAP_SLOTS * RING) {
            lr_static_for<0, RING>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (q + CAP_SLOTS * u < ntl_run) emit(gr[u]);
                fetch(gr[u]);
            });
        }
    }
    __syncthreads();
    CAP_CLK(3);
    // ---- 4. tile Cholesky
    const int TC = (n + 15) >> 4;                                // tile columns with a pivot
    const int li = lane & 15, lk = lane >> 4;
#ifdef LR_T_CAP_NOCHOL     // (wrong-result timing switch, tuning builds only)
    for (int J = 0; J < 1; ++J) {
#else
    for (int J = 0; J < TC; ++J) {
#endif
        double* Djj = Tl + (size_t)lr_tile(J, J) * LR_TILE;
#ifdef LR_T_CLK
        const unsigned long long ck0_ = wall_clock64();
#endif
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
#ifdef LR_T_CLK
        const unsigned long long ck1_ = wall_clock64();
#endif
        // TRSM: rows of the tiles (I, J), I > J: x L_JJ^T = a by forward substitution, a lane per row, column by column -- every later
        // column updated as soon as x_c is known: the dependent chain is 16 x (mul, fma), where the dot-product form accumulated 120
        // fma one after the other.  The entries of L_JJ are the same for every row: lane l keeps ROW l mod 16 of the tile (and
        // 1 / l_ll) in registers and the chain takes L[j][c] by v_readlane (read from LDS where they are used -- 120 broadcast
        // reads -- the compiler waited for 81 LDS round trips inside the chain: 2.6 of the 4.8 us a tile column of the factorisation
        // took; fetched a column ahead into registers instead, compiler and scheduler hoisted all 120 to the top and spilled ~250
        // registers whatever fences stood between the columns)
        {
            const int rows_below = 16 * (TR - J - 1);
            double Lrow[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Lrow[c] = Djj[lr_sw(lane & 15, c)];
            const double dv = (16 * J + (lane & 15) < n) ? dinv[16 * J + (lane & 15)] : 1.0;      // (padding columns: the identity)
            for (int rr = tid; rr < rows_below; rr += CAP_THREADS) {
                const int I = J + 1 + (rr >> 4), r = rr & 15;
                double* Tij = Tl + (size_t)lr_tile(I, J) * LR_TILE;
                double xv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) xv[c] = Tij[lr_sw(r, c)];
                lr_static_for<0, 16>([&](auto cc_) {
                    constexpr int c = decltype(cc_)::value;
                    xv[c] *= lr_readlane(dv, c);
                    lr_static_for<c + 1, 16>([&](auto jj_) {
                        constexpr int j = decltype(jj_)::value;
                        xv[j] = fma(-xv[c], lr_readlane(Lrow[c], j), xv[j]);
                    });
                });
#pragma unroll
                for (int c = 0; c < 16; ++c) Tij[lr_sw(r, c)] = xv[c];
            }
        }
        __syncthreads();
#ifdef LR_T_CLK
        const unsigned long long ck2_ = wall_clock64();
#endif
        // GEMM: A_IK -= L_IJ L_KJ^T for J < K <= I < TR on the matrix cores, tiles dealt to the waves
        {
            const int nb = TR - J - 1;
            const int ng = (nb * (nb + 1)) >> 1;
            int I = J + 1, Kt = J + 1;
            for (int g = 0; g < ng; ++g) {
                if ((g & (CAP_THREADS / 64 - 1)) == wave) {
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
#ifdef LR_T_CLK
        { const unsigned long long ck3_ = wall_clock64(); cph_[0] += ck1_ - ck0_; cph_[1] += ck2_ - ck1_; cph_[2] += ck3_ - ck2_; }
#endif
    }
    CAP_CLK(4);
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
    CAP_CLK(5);
    // ---- 6. w to its waypoints
    if (tid < n) wdense[((size_t)fact[tid] * B + b) * H + tact[tid]] = wv[tid];
#ifdef LR_T_CLK
    CAP_CLK(6);
    if (blockIdx.x == 0 && tid == 0)
        printf("cap clk (10 ns) n %d: compact %llu rhs %llu assemble %llu cholesky %llu (potrf %llu trsm %llu update %llu) back %llu scatter %llu\n", n,
               cclk_[1] - cclk_[0], cclk_[2] - cclk_[1], cclk_[3] - cclk_[2], cclk_[4] - cclk_[3], cph_[0], cph_[1], cph_[2], cclk_[5] - cclk_[4],
               cclk_[6] - cclk_[5]);
#endif
}

// ------------------------------------------------------------------------------------------------
// launcher (called by mpb_gpmp2_solve, mpb_gpmp2.hip)
// ------------------------------------------------------------------------------------------------
bool mpb_gpmp2_lr_ok(int H, int D, int n_fields) { return H >= 2 && D >= 1 && D <= MPB_MAX_DOF && n_fields >= 1 && n_fields * (H - 1) <= LR_NMAX; }

// doubles of workspace: shared tables (cyclic-reduction coefficients, G) + per-batch arrays (the steps, w, the GP cost, g_rest / u0, the order)
size_t mpb_gpmp2_lr_ws_doubles(int B, int H, int D) {
    const size_t NL = (size_t)B * D;
    return 2 * (size_t)D * pcr_coef_entries(H) + (size_t)D * H * H + (size_t)H * NL + (size_t)MPB_GP_MAX_FIELDS * B * H + (size_t)B + 64 +
           (size_t)B * H * 2 * D +         // ... and g_rest
           (size_t)B + 1;                  // ... and the size class of every particle + the launch order of the capacitance systems (2 B ints)
}

int mpb_gpmp2_lr_launch(float* x, const float* start, const float* goal, const float* jac, const double* diag_mean, double* ws,
                        float* costs_out, int B, int H, int D, int n_fields, const GpConst& K, hipStream_t stream) {
    const size_t NL = (size_t)B * D;
    const int L = pcr_levels(H);
    const size_t n_coef = pcr_coef_entries(H);
    double* coef = ws;                                                   // D tables of n_coef 16-byte entries
    double* G = coef + 2 * (size_t)D * n_coef;
    double* dth = G + (size_t)D * H * H;                                 // the steps of the particles with collision rows: (D, B, H) float pairs
    double* wdense = dth + (size_t)H * NL;
    double* gpcost = wdense + (size_t)MPB_GP_MAX_FIELDS * B * H;
    double* grest = gpcost + B + 64;
    int* ord = reinterpret_cast<int*>(grest + (size_t)B * H * 2 * D);        // [0, B): size class of particle b; [B, 2 B): the particles, largest class first
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const size_t lds_rows = (size_t)PCR_WAVES * PCR_NP * H * sizeof(lr_d2);
    const size_t lds_setup = n_coef * sizeof(lr_d2) + 12 * (size_t)H * sizeof(double) + lds_rows;
    const size_t lds_solve = n_coef * sizeof(lr_d2) + lds_rows;
    const int NY = (H + 15) / 16;                                                // sixteen columns per block (one pass of eight of its waves: 12 us where a full pass of 32 columns took 15)
    hipLaunchKernelGGL(gpmp2_pcr_setup, dim3(D, NY), dim3(PCR_THREADS), lds_setup, stream, diag_mean, coef, G, H, D, L, K);
    hipLaunchKernelGGL(gpmp2_lr_gradient, dim3(B), dim3(256), 0, stream, x, start, goal, jac, grest, gpcost, ord, B, H, D, n_fields, K);
    // particle groups: one block per CU over the D joints, at least one pass of the waves per group
    int NG = n_cu / D;
    if (NG < 1) NG = 1;
    if (NG > (B + PCR_WAVES * PCR_NP - 1) / (PCR_WAVES * PCR_NP)) NG = (B + PCR_WAVES * PCR_NP - 1) / (PCR_WAVES * PCR_NP);
    hipLaunchKernelGGL(gpmp2_pcr_solve<false>, dim3(D, NG + 1), dim3(PCR_THREADS), lds_solve, stream, grest, jac, wdense, coef, dth, ord, B, H, D,
                       n_fields, L, NG, K);
    const int n_max = n_fields * (H - 1);
    const int trm = (n_max + 16) >> 4, ntm = (trm * (trm + 1)) >> 1;             // tiles of the largest system the shape allows
    const size_t lds = ((size_t)ntm * LR_TILE + 256) * sizeof(double) + (size_t)LR_NMAX * (D <= 8 ? 8 : MPB_MAX_DOF) * sizeof(float) + (256 + 16) * sizeof(int);
    if (D <= 8) hipLaunchKernelGGL(gpmp2_lr_cap<8>, dim3(B), dim3(CAP_THREADS), lds, stream, jac, grest, G, gpcost, wdense, costs_out, ord, x, B, H, D, n_fields, ntm, K);
    else hipLaunchKernelGGL(gpmp2_lr_cap<MPB_MAX_DOF>, dim3(B), dim3(CAP_THREADS), lds, stream, jac, grest, G, gpcost, wdense, costs_out, ord, x, B, H, D, n_fields, ntm, K);
    hipLaunchKernelGGL(gpmp2_pcr_solve<true>, dim3(D, NG), dim3(PCR_THREADS), lds_solve, stream, grest, jac, wdense, coef, dth, ord, B, H, D,
                       n_fields, L, NG, K);
    hipLaunchKernelGGL(gpmp2_lr_apply, dim3(B), dim3(256), 0, stream, x, reinterpret_cast<const float2*>(dth), grest, ord, B, H, D, K.step);
    return MPB_OK;
}
