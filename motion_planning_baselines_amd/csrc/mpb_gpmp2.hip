// mpb_gpmp2.hip -- GPMP2 Gauss-Newton step without the dense (A, b, K).
//
// The reference (gpmp2.py:308-368, cost_functions.py:107-144,191-231,291-314,538-554) stacks a dense
// A (B,M,N), K (B,M,M) with N = 2D*H, forms A^T K A (B,N,N) and runs a dense Cholesky.  The factor graph
// is a chain, so A^T K A is block tridiagonal with 2D x 2D blocks (tests/golden: |outside band| == 0):
//
//   diag block t : [t=0] K_s + [t<H-1] Phi^T Qi Phi + [t>0] Qi + [t=H-1] K_g
//                  + [t>0] (1/sigma_c^2) h_t h_t^T (position rows/cols) + damping
//   block (t,t+1): U = -Phi^T Qi                                  (constant)
//   rhs block t  : [t=0] K_s (mu_s - x_0) + [t<H-1] Phi^T Qi e_t - [t>0] Qi e_{t-1}
//                  + [t=H-1] K_g (goal - x_{H-1}) + [t>0] (1/sigma_c^2) h_t c_t
//   with e_t = x_{t+1} - Phi x_t (gp_factor.py:52-56), c_t the collision cost of waypoint t and
//   h_t = -d c_t / d q_t (field_factor.py:54).
//
// Solve per particle by block elimination (block Thomas):
//   S_0 = D_0, r_0 = g_0;   W_t = S_t^-1, z_t = W_t r_t;   S_{t+1} = D_{t+1} - U^T W_t U;  r_{t+1} = g_{t+1} - U^T z_t
//   dtheta_{H-1} = z_{H-1};  dtheta_t = z_t - W_t U dtheta_{t+1}
// One wave per particle; the 2D x 2D blocks live in LDS as 16 x 16 fp64 tiles; W_t, z_t go to a
// caller-provided workspace.  All arithmetic is fp64: the weights reach 1/sigma^2 = 1e10 (gpmp2.py:32-35)
// and fp32 Cholesky at that conditioning is not reproducible; storage stays fp32.
#include "mpb_common.h"
#include "mpb_geom.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define GP_N 16            // padded block size (2D <= 16)
#define GP_LD 17           // LDS leading dimension (fp64 words)
#define GP_MAXH MPB_MAX_H

// ------------------------------------------------------------------------------------------------
// linearisation of the collision factor: jac[b][t][0..D) = h_t = -d c_t/d q, jac[b][t][D] = c_t
// (t = 0 is excluded from the collision factor: traj_range [1, None]).  One wave per particle, lane = waypoint.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpmp2_linearize_kernel(const float* __restrict__ x, const float* __restrict__ geom,
                                                              float* __restrict__ jac_all, int B, int H, int D, int n_interp) {
    __shared__ unsigned gridw[MPB_GRID_MAX_CELLS];              // broad-phase grid of the field being linearised
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool dead = b >= B;                                   // such waves still take part in the block barriers
    const int dim = 2 * D;
    // one (h_t, c_t) set per chained collision field f (the reference stacks one block of H-1 rows per field,
    // cost_functions.py:107-144), scaled by sqrt(s_f) so that the solve kernel only has to sum kc h h^T, kc h c, kc c^2
    int fidx = 0;
    for (const float* gp = geom; gp != nullptr; gp = geom_next(gp), ++fidx) {
        const GeomView G = geom_view(gp);
        const bool ug = grid_usable_grad(G);
        __syncthreads();
        if (ug) grid_stage(G, gridw, otab, threadIdx.x, blockDim.x);
        __syncthreads();
        if (dead) continue;
        const float rs = __builtin_amdgcn_sqrtf(G.fscale);
        float* jac = jac_all + (size_t)fidx * B * H * (D + 1);
        // n_interp > 0 (CostComposite.get_linear_system with n_interpolated_points, cost_functions.py:115-119;
        // field_factor.py:42-54): the Jacobian row of support point t is d/dq_t of the summed cost of the
        // INTERPOLATED trajectory, i.e. its own gradient plus (1-a) * grad of every interior point of segment
        // (t, t+1) plus a * grad of every interior point of segment (t-1, t); the error c_t stays the support
        // point's own.  Lane t evaluates the interior points of ITS segment once and hands the a-weighted part
        // to lane t+1 (carry across 64-waypoint chunks).
        float carry[MPB_MAX_DOF];
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) carry[i] = 0.f;
        for (int base = 0; base < H; base += 64) {
            const int t = base + lane;
            const bool active = t < H;
            const float* row = x + ((size_t)b * H + (active ? t : 0)) * dim;
            float q[MPB_MAX_DOF], dq[MPB_MAX_DOF], gnext[MPB_MAX_DOF];
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) {
                q[i] = (i < D) ? row[i] : 0.f;
                dq[i] = 0.f;
                gnext[i] = 0.f;
            }
            float c = 0.f;
            if (active && t >= 1) c = ug ? waypoint_cost_grid_grad(G, gridw, otab, q, dq) : waypoint_cost<true>(G, q, dq);
            if (n_interp > 0) {
                if (active && t + 1 < H) {
                    float qn[MPB_MAX_DOF];
#pragma unroll
                    for (int i = 0; i < MPB_MAX_DOF; ++i) qn[i] = (i < D) ? row[dim + i] : 0.f;
                    for (int k = 1; k <= n_interp; ++k) {
                        const float al = (float)k / (float)(n_interp + 1);
                        float qi[MPB_MAX_DOF], dqi[MPB_MAX_DOF];
#pragma unroll
                        for (int i = 0; i < MPB_MAX_DOF; ++i) qi[i] = q[i] + al * (qn[i] - q[i]);
                        if (ug) waypoint_cost_grid_grad(G, gridw, otab, qi, dqi);
                        else waypoint_cost<true>(G, qi, dqi);
#pragma unroll
                        for (int i = 0; i < MPB_MAX_DOF; ++i) {
                            dq[i] = fmaf(1.f - al, dqi[i], dq[i]);
                            gnext[i] = fmaf(al, dqi[i], gnext[i]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i) {
                    const float up = __shfl_up(gnext[i], 1, 64);
                    dq[i] += (lane == 0) ? carry[i] : up;
                    carry[i] = __shfl(gnext[i], 63, 64);
                }
            }
            if (active) {
                float* o = jac + ((size_t)b * H + t) * (D + 1);
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i)
                    if (i < D) o[i] = -rs * dq[i];
                o[D] = rs * c;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// local SUM over particles of diag(A^T K A) (quirk Q9 needs its batch mean).  grid = H, block = 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpmp2_diag_kernel(const float* __restrict__ jac, double* __restrict__ diag_sum,
                                                         int B, int H, int D, int F, double dt, double ks, double kgp,
                                                         double kg, double kc) {
    const int t = blockIdx.x;
    const int dim = 2 * D;
    __shared__ double red[4][MPB_MAX_DOF];
    double acc[MPB_MAX_DOF];
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) acc[i] = 0.0;
    for (int f = 0; f < F; ++f)
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            const float* o = jac + (((size_t)f * B + b) * H + t) * (D + 1);
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i)
                if (i < D) acc[i] += (double)o[i] * (double)o[i];
        }
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) acc[i] = wave_sum_f64(acc[i]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) red[threadIdx.x >> 6][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < dim) {
        const int i = threadIdx.x;
        const bool pos = i < D;
        // constant part of the diagonal (identical for every particle)
        double c = 0.0;
        if (t == 0) c += ks;
        if (t < H - 1) c += pos ? 12.0 / (dt * dt * dt) * kgp : 4.0 / dt * kgp;   // Phi^T Qi Phi
        if (t > 0) c += pos ? 12.0 / (dt * dt * dt) * kgp : 4.0 / dt * kgp;       // Qi
        if (t == H - 1) c += kg;
        double s = c * (double)B;
        if (pos && t > 0) {
            double h2 = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) h2 += red[w][i < MPB_MAX_DOF ? i : 0];
            s += kc * h2;
        }
        diag_sum[(size_t)t * dim + i] = s;
    }
}

__global__ void gpmp2_scale_kernel(const double* __restrict__ in, double* __restrict__ out, int n, double s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * s;
}

// ------------------------------------------------------------------------------------------------
// block-tridiagonal solve + update, one wave per particle.
//
// U = -Phi^T Qi is (2x2) (x) I_D, so F_t = S_t^-1 U is a combination of column blocks of W_t = S_t^-1 and
//   S_{t+1} = D_{t+1} - U^T W_t U,   z_t = W_t r_t,   r_{t+1} = g_{t+1} - U^T z_t,
//   dtheta_t = z_t - W_t (U dtheta_{t+1}).
// The only dense operation per waypoint is the SPD inverse W_t: blocked Gauss-Jordan (no pivoting: the
// pivot blocks of an SPD matrix are SPD) on a 16 x 16 fp64 tile with 4 x 4 pivot blocks; each block step is
// one v_mfma_f64_16x16x4_f64 rank-4 update of the whole tile, ping-ponging between two LDS buffers
// (4 wave-level synchronisations per inverse).
// ------------------------------------------------------------------------------------------------
// 1/x in fp64: v_rcp_f64 (about 2^-26 accurate) + two Newton steps; the IEEE division hipcc emits costs
// ~40 instructions and there are 16 of them per waypoint in the pivot-block inverses
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// broadcast of a double held by lane `l` (compile-time constant after unrolling): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

struct GpConst {
    double dt, ks, kgp, kg, kc, delta, step;
    int trust;
};

#define GP_WS_PER_T (GP_N * GP_N + GP_N)   // workspace doubles per waypoint: W_t (16x16 row-major) + z_t

// DT > 0: the number of degrees of freedom is a compile-time constant (loop bounds, pivot-block count and the
// position/velocity index tests fold away); DT == 0: generic.
template <int DT, bool MULTI>
__global__ __launch_bounds__(64) void gpmp2_solve_kernel(float* __restrict__ x, const float* __restrict__ start,
                                                         const float* __restrict__ goal, const float* __restrict__ jac,
                                                         const double* __restrict__ diag_mean, double* __restrict__ work,
                                                         float* __restrict__ costs_out, int B, int H, int Drt, int Frt, GpConst K) {
    const int D = DT ? DT : Drt;
    const int F = MULTI ? Frt : 1;       // MULTI == false: one collision field, the field loops fold away
    __shared__ double Sb[1][GP_N * GP_LD];  // W_t for the matvec / next-tile reads (the inverse itself runs in registers)
    __shared__ double xs[2][GP_N];          // x_t, x_{t+1} (fp64 copies)
    __shared__ double rv[GP_N];             // r_t
    __shared__ double zv[GP_N];             // z_t / scratch vector
    __shared__ double dth[GP_N];            // dtheta_{t+1} during the backward pass
    __shared__ double hv[MPB_MAX_FIELDS][GP_N];   // per field: collision Jacobian h_t (D values) and cost c_t at [D]
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int dim = 2 * D;
    const double dt = K.dt;
    // 2x2 GP coefficient matrices (Kronecker with I_D)
    const double a = 12.0 / (dt * dt * dt) * K.kgp, bq = -6.0 / (dt * dt) * K.kgp, cq = 4.0 / dt * K.kgp;  // Qi
    const double p00 = a, p01 = 6.0 / (dt * dt) * K.kgp, p11 = cq;                                            // Phi^T Qi Phi
    // U = -Phi^T Qi = -[[a, bq],[a dt + bq, bq dt + cq]]   (U[c][e]: c,e in {pos, vel})
    const double u00 = -a, u01 = -bq, u10 = -(a * dt + bq), u11 = -(bq * dt + cq);
    double* wW = work + (size_t)b * H * GP_WS_PER_T;
    float* xb = x + (size_t)b * H * dim;
    const float* jb = jac + (size_t)b * H * (D + 1);
    double cost = 0.0;
    // element ownership for the 16x16 tile = the C/D layout of v_mfma_f64_16x16x4_f64: lane (lk, li) holds rows
    // lk + 4q (q = 0..3) of column li.  The tile stays in registers from its assembly through the whole inverse.
    const int li = lane & 15, lk = lane >> 4;
    f64x4 Snext = {0.0, 0.0, 0.0, 0.0};  // -(U^T W_{t-1} U) on entry to step t > 0, same layout
    double rcarry = 0.0;  // lane < dim: r_t contribution carried from step t-1 (gnext - U^T z)

    // per-element constants of the S assembly (element q of this lane: row lk + 4q, column li)
    double asm_g1[4], asm_g2[4], asm_dg[4], asm_pp[4], asm_id[4];
    bool asm_in[4];
    int asm_hi[4];
    const int asm_di = (li < dim) ? li : 0;    // the diagonal element of column li is row li
    const int asm_hj = (li < D) ? li : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q, j = li;
        const bool in = i < dim && j < dim;
        const bool ip = i < D, jp = j < D;
        const int ii = ip ? i : i - D, jj = jp ? j : j - D;
        const bool same = in && ii == jj;
        asm_in[q] = in;
        asm_g1[q] = same ? (ip ? (jp ? p00 : p01) : (jp ? p01 : p11)) : 0.0;   // Phi^T Qi Phi block (t < H-1)
        asm_g2[q] = same ? (ip ? (jp ? a : bq) : (jp ? bq : cq)) : 0.0;         // Qi block (t > 0)
        asm_dg[q] = (in && i == j) ? 1.0 : 0.0;
        asm_pp[q] = (in && ip && jp) ? 1.0 : 0.0;
        asm_hi[q] = (i < D) ? i : 0;
        asm_id[q] = (i == j) ? 1.0 : 0.0;
    }

    // per-element constants of the next-tile product -(U^T W U)
    double nt_c[4][4];
    int nt_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q, j = li;
        const bool in = i < dim && j < dim;
        const bool ip = i < D, jp = j < D;
        const int ii = in ? (ip ? i : i - D) : 0, jj = in ? (jp ? j : j - D) : 0;
        const double uca0 = ip ? u00 : u01, uca1 = ip ? u10 : u11;   // U[c][a], c = 0,1
        const double ueb0 = jp ? u00 : u01, ueb1 = jp ? u10 : u11;   // U[e][b], e = 0,1
        const double m = in ? -1.0 : 0.0;
        nt_c[q][0] = m * uca0 * ueb0;   // W[ii][jj]
        nt_c[q][1] = m * uca0 * ueb1;   // W[ii][jj + D]
        nt_c[q][2] = m * uca1 * ueb0;   // W[ii + D][jj]
        nt_c[q][3] = m * uca1 * ueb1;   // W[ii + D][jj + D]
        nt_off[q] = ii * GP_LD + jj;
    }

    // software prefetch: rows t+1 of x and of the Jacobian are loaded one step ahead so that their global
    // latency hides behind the inverse of step t instead of sitting on the sequential critical path
    float xr0 = (lane < dim) ? xb[lane] : 0.f;
    float xr1 = (lane < dim && H > 1) ? xb[dim + lane] : 0.f;
    float jr = 0.f;                                                   // row 0 takes no collision factor
    float jr1 = (lane <= D && H > 1) ? jb[(D + 1) + lane] : 0.f;
    double dm0 = K.trust ? diag_mean[asm_di] : 0.0;
    double dm1 = (K.trust && H > 1) ? diag_mean[dim + asm_di] : 0.0;
    for (int t = 0; t < H; ++t) {
        // ---- x_t, x_{t+1}, h_t from the prefetched registers; issue the loads of step t+1
        if (lane < dim) {
            xs[0][lane] = (double)xr0;
            xs[1][lane] = (t + 1 < H) ? (double)xr1 : 0.0;
        }
        if (lane <= D) {
            hv[0][lane] = (t > 0) ? (double)jr : 0.0;
            for (int f = 1; f < F; ++f)      // further chained fields: not prefetched (the single-field path stays lean)
                hv[f][lane] = (t > 0) ? (double)jb[(size_t)f * B * H * (D + 1) + t * (D + 1) + lane] : 0.0;
        }
        const float xr2 = (lane < dim && t + 2 < H) ? xb[(t + 2) * dim + lane] : 0.f;
        const float jr2 = (lane <= D && t + 2 < H) ? jb[(t + 2) * (D + 1) + lane] : 0.f;
        const double dm2 = (K.trust && t + 2 < H) ? diag_mean[(size_t)(t + 2) * dim + asm_di] : 0.0;
        wave_sync();
        // ---- GP error of factor t: e = x_{t+1} - Phi x_t
        double e_i = 0.0;
        if (lane < dim && t + 1 < H) {
            const bool pos = lane < D;
            e_i = pos ? xs[1][lane] - (xs[0][lane] + dt * xs[0][lane + D]) : xs[1][lane] - xs[0][lane];
        }
        double qe_i = 0.0, pqe_i = 0.0;
        {
            const double e_partner = __shfl(e_i, (lane < D) ? lane + D : lane - D, 64);
            if (lane < dim && t + 1 < H) {
                const bool pos = lane < D;
                const double ep = pos ? e_i : e_partner, ev = pos ? e_partner : e_i;
                const double qp = a * ep + bq * ev, qv = bq * ep + cq * ev;      // Qi e
                qe_i = pos ? qp : qv;
                pqe_i = pos ? qp : dt * qp + qv;                                 // Phi^T (Qi e)
                cost += pos ? ep * qp : ev * qv;
            }
        }
        // ---- S = D_t (+ Schur term carried in registers for t > 0); padding rows/cols = identity.  Branch-free: the
        //      per-element coefficients (asm_*) were fixed before the loop, only the t-dependent selects remain
        f64x4 T;
        {
            const double first = (t == 0) ? 1.0 : 0.0, notfirst = 1.0 - first, notlast = (t < H - 1) ? 1.0 : 0.0;
            double hj[MPB_MAX_FIELDS];
#pragma unroll
            for (int f = 0; f < MPB_MAX_FIELDS; ++f) hj[f] = (f < F) ? hv[f][asm_hj] * (K.kc * notfirst) : 0.0;
            // damping of the diagonal element of this lane's column (prefetched one step ahead, like x and h)
            const double dg = (K.trust ? K.delta * dm0 : K.delta) + first * K.ks + (1.0 - notlast) * K.kg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double v = (t > 0) ? Snext[q] : 0.0;
                v = fma(notlast, asm_g1[q], v);
                v = fma(notfirst, asm_g2[q], v);
                v = fma(asm_dg[q], dg, v);
#pragma unroll
                for (int f = 0; f < MPB_MAX_FIELDS; ++f)
                    if (f < F) v = fma(asm_pp[q] * hj[f], hv[f][asm_hi[q]], v);
                T[q] = asm_in[q] ? v : asm_id[q];
            }
        }
        // ---- r_t
        double gnext = 0.0;
        if (lane < dim) {
            double r = (t > 0) ? rcarry : 0.0;
            if (t == 0) {
                const double es = (double)start[(size_t)b * dim + lane] - xs[0][lane];
                r += K.ks * es;
                cost += K.ks * es * es;
            }
            if (t == H - 1) {
                const double eg = (double)goal[(size_t)b * dim + lane] - xs[0][lane];
                r += K.kg * eg;
                cost += K.kg * eg * eg;
            }
            if (t < H - 1) r += pqe_i;
            if (t > 0 && lane < D)
                for (int f = 0; f < F; ++f) r += K.kc * hv[f][lane] * hv[f][D];
            gnext = -qe_i;                                         // contribution of factor t to g_{t+1}
            rv[lane] = r;
        }
        if (lane == 0 && t > 0)
            for (int f = 0; f < F; ++f) cost += K.kc * hv[f][D] * hv[f][D];
        wave_sync();
        // ---- W = S^-1 : blocked Gauss-Jordan with 4x4 pivot blocks, entirely in registers.  Per block step K:
        //        D = (-A[:,K]) * (Pinv * A'[K,:]) + C_in,  A'[K,K] := I,  C_in := A with columns K zeroed,
        //      one v_mfma_f64_16x16x4_f64, then rows K := Pinv * A'[K,:] (this lane's own B operand).
        //      Lane maps (f64 16x16x4): A-op lane l -> [i = l&15][k = l>>4]; B-op [k = l>>4][j = l&15]; C/D: column
        //      l&15, rows (l>>4) + 4*reg.  Nothing goes through LDS:
        //        * the pivot block is broadcast with v_readlane (row k0+r lives in register kb of lanes (r, .));
        //        * the B operand needs rows K of column li = register kb of lanes (m, li): four shuffles;
        //        * the A operand -A[li][k0+lk] is this lane's OWN register kb up to a sign: the working matrix of
        //          the Gauss-Jordan inverse of a symmetric matrix satisfies M[a][b] = s M[b][a] with s = -1 when
        //          exactly one of a, b belongs to an already processed block, +1 otherwise (induction over the
        //          block steps), and T[kb] = M[k0+lk][li].
        {
            const double rowsel[4] = {lk == 0 ? 1.0 : 0.0, lk == 1 ? 1.0 : 0.0, lk == 2 ? 1.0 : 0.0, lk == 3 ? 1.0 : 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (4 * kb < dim) {
                    const int k0 = 4 * kb;
                    const double tk = T[kb];
                    // pivot block (symmetric: upper triangle) from lanes (r, k0 + c)
                    const double a00 = readlane_f64(tk, 0 * 16 + k0 + 0), a01 = readlane_f64(tk, 0 * 16 + k0 + 1);
                    const double b00 = readlane_f64(tk, 0 * 16 + k0 + 2), b01 = readlane_f64(tk, 0 * 16 + k0 + 3);
                    const double a11 = readlane_f64(tk, 1 * 16 + k0 + 1);
                    const double b10 = readlane_f64(tk, 1 * 16 + k0 + 2), b11 = readlane_f64(tk, 1 * 16 + k0 + 3);
                    const double c00 = readlane_f64(tk, 2 * 16 + k0 + 2), c01 = readlane_f64(tk, 2 * 16 + k0 + 3);
                    const double c11 = readlane_f64(tk, 3 * 16 + k0 + 3);
                    // SPD 4x4 inverse in 2x2 blocks: X = A^-1 B, S = C - B^T X,
                    //   P^-1 = [[A^-1 + X S^-1 X^T, -X S^-1], [-(X S^-1)^T, S^-1]]      (2 reciprocals, ~50 fma)
                    double pv[4][4];
                    {
                        const double ia = fast_rcp(fma(a00, a11, -a01 * a01));
                        const double i00 = a11 * ia, i01 = -a01 * ia, i11 = a00 * ia;              // A^-1
                        const double x00 = fma(i00, b00, i01 * b10), x01 = fma(i00, b01, i01 * b11);   // X = A^-1 B
                        const double x10 = fma(i01, b00, i11 * b10), x11 = fma(i01, b01, i11 * b11);
                        const double s00 = c00 - fma(b00, x00, b10 * x10), s01 = c01 - fma(b00, x01, b10 * x11);
                        const double s11 = c11 - fma(b01, x01, b11 * x11);                          // S = C - B^T X
                        const double is = fast_rcp(fma(s00, s11, -s01 * s01));
                        const double t00 = s11 * is, t01 = -s01 * is, t11 = s00 * is;              // S^-1
                        const double y00 = -fma(x00, t00, x01 * t01), y01 = -fma(x00, t01, x01 * t11);  // -X S^-1
                        const double y10 = -fma(x10, t00, x11 * t01), y11 = -fma(x10, t01, x11 * t11);
                        pv[0][0] = i00 - fma(y00, x00, y01 * x01);                                  // A^-1 + X S^-1 X^T
                        pv[0][1] = pv[1][0] = i01 - fma(y00, x10, y01 * x11);
                        pv[1][1] = i11 - fma(y10, x10, y11 * x11);
                        pv[0][2] = pv[2][0] = y00; pv[0][3] = pv[3][0] = y01;
                        pv[1][2] = pv[2][1] = y10; pv[1][3] = pv[3][1] = y11;
                        pv[2][2] = t00; pv[2][3] = pv[3][2] = t01; pv[3][3] = t11;
                    }
                    // B operand (Pinv * A'[K,:])[lk][li]; lane-dependent choices are 0/1 multipliers, not selects
                    // (hipcc lowers such selects to trees of exec-mask branches)
                    const bool jin = (li >= k0) && (li < k0 + 4);
                    const double notj = jin ? 0.0 : 1.0;
                    double bop = 0.0;
                    {
                        double am[4];
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            am[m] = fma(notj, __shfl(tk, m * 16 + li, 64), (li - k0 == m) ? 1.0 : 0.0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double rowdot = pv[r][0] * am[0];
#pragma unroll
                            for (int m = 1; m < 4; ++m) rowdot = fma(pv[r][m], am[m], rowdot);
                            bop = fma(rowsel[r], rowdot, bop);
                        }
                    }
                    const double aop = (li < k0) ? tk : -tk;
                    f64x4 cin;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cin[q] = notj * T[q];
                    T = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, cin, 0, 0, 0);
                    T[kb] = bop;                                    // rows of the pivot block
                }
            }
        }
        // W_t to LDS once (z = W r, the next Schur tile) and to the workspace straight from the registers
        double* Wl = Sb[0];
#pragma unroll
        for (int q = 0; q < 4; ++q) Wl[(lk + 4 * q) * GP_LD + li] = T[q];
        wave_sync();
        const double* W = Wl;
        // ---- z = W r ; store W_t, z_t
        double* wt = wW + (size_t)t * GP_WS_PER_T;
        double zi = 0.0;
        if (lane < dim) {
            for (int j = 0; j < dim; ++j) zi = fma(W[lane * GP_LD + j], rv[j], zi);
            zv[lane] = zi;
            wt[GP_N * GP_N + lane] = zi;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) wt[(lk + 4 * q) * GP_N + li] = T[q];
        wave_sync();
        if (t < H - 1) {
            // ---- next tile: -(U^T W U), block (a,b) (i',j') = -sum_{c,e} U[c][a] U[e][b] W[i'+cD][j'+eD]; the four
            //      coefficient products and the element offsets are per-lane constants (nt_*), so 4 fma per element
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* Wq = W + nt_off[q];
                double v = nt_c[q][0] * Wq[0];
                v = fma(nt_c[q][1], Wq[D], v);
                v = fma(nt_c[q][2], Wq[D * GP_LD], v);
                v = fma(nt_c[q][3], Wq[D * GP_LD + D], v);
                Snext[q] = v;
            }
            // r_{t+1} carry = gnext - U^T z
            if (lane < dim) {
                const bool ip = lane < D;
                const int ii = ip ? lane : lane - D;
                const double zp = zv[ii], zvv = zv[ii + D];
                rcarry = gnext - (ip ? u00 * zp + u10 * zvv : u01 * zp + u11 * zvv);
            }
            wave_sync();                                           // W / zv reads done before the next step overwrites them
        }
        xr0 = xr1; xr1 = xr2; jr = jr1; jr1 = jr2; dm0 = dm1; dm1 = dm2;
    }
    // ---- backward substitution and update: dtheta_t = z_t - W_t (U dtheta_{t+1}).  Row `lane` of W_{t-1}, z_{t-1}
    //      and x_{t-1} are fetched while step t runs: the workspace (B*H*2.2 KB) does not stay in cache, and an
    //      un-prefetched global round trip per waypoint would sit on the sequential critical path
    double wrow[GP_N], wnext[GP_N], zc = 0.0, zn = 0.0;
    float xc = 0.f, xn = 0.f;
    const bool rowlane = lane < dim;
    const int rl = rowlane ? lane : 0;          // idle lanes shadow row 0: unconditional loads, no exec-mask branches
    {
        const double* wt = wW + (size_t)(H - 1) * GP_WS_PER_T;
#pragma unroll
        for (int j = 0; j < GP_N; ++j) wrow[j] = wt[rl * GP_N + j];
        zc = wt[GP_N * GP_N + rl];
        xc = xb[(H - 1) * dim + rl];
    }
    for (int t = H - 1; t >= 0; --t) {
        if (t > 0) {
            const double* wt = wW + (size_t)(t - 1) * GP_WS_PER_T;
#pragma unroll
            for (int j = 0; j < GP_N; ++j) wnext[j] = wt[rl * GP_N + j];
            zn = wt[GP_N * GP_N + rl];
            xn = xb[(t - 1) * dim + rl];
        }
        double d = zc;
        if (rowlane && t < H - 1) {
#pragma unroll
            for (int j = 0; j < GP_N; ++j)
                if (j < dim) d -= wrow[j] * zv[j];                               // zv holds U dtheta_{t+1}
        }
        wave_sync();
        if (rowlane) {
            dth[lane] = d;
            xb[t * dim + lane] = (float)((double)xc + K.step * d);
        }
        wave_sync();
        if (rowlane) {   // v = U dtheta_t for the next (earlier) waypoint
            const bool ip = lane < D;
            const int ii = ip ? lane : lane - D;
            const double dp = dth[ii], dv = dth[ii + D];
            zv[lane] = ip ? u00 * dp + u01 * dv : u10 * dp + u11 * dv;
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < GP_N; ++j) wrow[j] = wnext[j];
        zc = zn;
        xc = xn;
    }
    cost = wave_sum_f64(cost);
    if (costs_out != nullptr && lane == 0) costs_out[b] = (float)cost;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
static bool gp_shape_ok(int B, int H, int D) { return B >= 0 && H >= 2 && H <= GP_MAXH && D >= 1 && D <= MPB_MAX_DOF; }

extern "C" size_t mpb_gpmp2_workspace_bytes(int B, int H, int D) {
    if (!gp_shape_ok(B, H, D)) return 0;
    const size_t jac = (size_t)MPB_MAX_FIELDS * B * H * (D + 1) * sizeof(float);   // one (h, c) set per chained field
    const size_t diag = 2 * (size_t)H * 2 * D * sizeof(double);
    const size_t fz = (size_t)B * H * GP_WS_PER_T * sizeof(double);
    return ((jac + 255) / 256) * 256 + ((diag + 255) / 256) * 256 + fz;
}

struct GpWork {
    float* jac;
    double* diag_sum;
    double* diag_mean;
    double* fz;
};
static GpWork gp_carve(void* ws, int B, int H, int D) {
    GpWork w;
    char* p = (char*)ws;
    w.jac = (float*)p;
    p += (((size_t)MPB_MAX_FIELDS * B * H * (D + 1) * sizeof(float)) + 255) / 256 * 256;
    w.diag_sum = (double*)p;
    w.diag_mean = w.diag_sum + (size_t)H * 2 * D;
    p += ((2 * (size_t)H * 2 * D * sizeof(double)) + 255) / 256 * 256;
    w.fz = (double*)p;
    return w;
}

extern "C" int mpb_gpmp2_linearize(const float* x, const float* geom, void* workspace, int B, int H, int D, int n_interp,
                                   void* stream) {
    if (!x || !geom || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_linearize: null pointer");
    if (!gp_shape_ok(B, H, D) || n_interp < 0 || n_interp > 64) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_linearize: bad shape");
    if (B == 0) return MPB_OK;
    GpWork w = gp_carve(workspace, B, H, D);
    hipLaunchKernelGGL(gpmp2_linearize_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, geom, w.jac, B, H, D,
                       n_interp);
    return mpb_check_launch("mpb_gpmp2_linearize");
}

extern "C" int mpb_gpmp2_diag(void* workspace, double* diag_sum_out, int B, int H, int D, int n_fields, float dt,
                              float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll, void* stream) {
    if (!workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_diag: null pointer");
    if (!gp_shape_ok(B, H, D) || n_fields < 1 || n_fields > MPB_MAX_FIELDS) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_diag: bad shape");
    GpWork w = gp_carve(workspace, B, H, D);
    double* out = diag_sum_out ? diag_sum_out : w.diag_sum;
    hipLaunchKernelGGL(gpmp2_diag_kernel, dim3(H), dim3(256), 0, (hipStream_t)stream, w.jac, out, B, H, D, n_fields,
                       (double)dt,
                       1.0 / ((double)sigma_start * sigma_start), 1.0 / ((double)sigma_gp * sigma_gp),
                       1.0 / ((double)sigma_goal * sigma_goal), 1.0 / ((double)sigma_coll * sigma_coll));
    return mpb_check_launch("mpb_gpmp2_diag");
}

extern "C" int mpb_gpmp2_solve(float* x, const float* start, const float* goal, const double* diag_mean, void* workspace,
                               float* costs_out, int B, int H, int D, int n_fields, float dt, float sigma_start,
                               float sigma_gp, float sigma_goal, float sigma_coll, float delta, int trust_region,
                               float step_size, void* stream) {
    if (!x || !start || !goal || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_solve: null pointer");
    if (!gp_shape_ok(B, H, D) || n_fields < 1 || n_fields > MPB_MAX_FIELDS) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_solve: bad shape");
    if (B == 0) return MPB_OK;
    GpWork w = gp_carve(workspace, B, H, D);
    GpConst K;
    K.dt = dt;
    K.ks = 1.0 / ((double)sigma_start * sigma_start);
    K.kgp = 1.0 / ((double)sigma_gp * sigma_gp);
    K.kg = 1.0 / ((double)sigma_goal * sigma_goal);
    K.kc = 1.0 / ((double)sigma_coll * sigma_coll);
    K.delta = delta;
    K.step = step_size;
    K.trust = trust_region;
    const double* dm = trust_region ? (diag_mean ? diag_mean : w.diag_mean) : nullptr;
#define GP_LAUNCH(DT)                                                                                                   \
    if (n_fields == 1)                                                                                                  \
        hipLaunchKernelGGL((gpmp2_solve_kernel<DT, false>), dim3(B), dim3(64), 0, (hipStream_t)stream, x, start, goal,  \
                           w.jac, dm, w.fz, costs_out, B, H, D, n_fields, K);                                           \
    else                                                                                                                \
        hipLaunchKernelGGL((gpmp2_solve_kernel<DT, true>), dim3(B), dim3(64), 0, (hipStream_t)stream, x, start, goal,   \
                           w.jac, dm, w.fz, costs_out, B, H, D, n_fields, K)
    switch (D) {
        case 2: GP_LAUNCH(2); break;
        case 3: GP_LAUNCH(3); break;
        case 7: GP_LAUNCH(7); break;
        default: GP_LAUNCH(0); break;
    }
#undef GP_LAUNCH
    return mpb_check_launch("mpb_gpmp2_solve");
}

extern "C" int mpb_gpmp2_step(float* x, const float* start, const float* goal, const float* geom, void* workspace,
                              float* costs_out, int B, int H, int D, float dt, float sigma_start, float sigma_gp,
                              float sigma_goal, float sigma_coll, float delta, int trust_region, float step_size,
                              int n_iters, int n_interp, int n_fields, void* stream) {
    if (!x || !start || !goal || !geom || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_step: null pointer");
    if (!gp_shape_ok(B, H, D) || n_iters < 0 || n_fields < 1 || n_fields > MPB_MAX_FIELDS)
        return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_step: bad shape");
    if (B == 0) return MPB_OK;
    GpWork w = gp_carve(workspace, B, H, D);
    for (int it = 0; it < n_iters; ++it) {
        int rc = mpb_gpmp2_linearize(x, geom, workspace, B, H, D, n_interp, stream);
        if (rc) return rc;
        if (trust_region) {
            rc = mpb_gpmp2_diag(workspace, nullptr, B, H, D, n_fields, dt, sigma_start, sigma_gp, sigma_goal, sigma_coll, stream);
            if (rc) return rc;
            const int n = H * 2 * D;
            hipLaunchKernelGGL(gpmp2_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w.diag_sum,
                               w.diag_mean, n, 1.0 / (double)B);
        }
        rc = mpb_gpmp2_solve(x, start, goal, nullptr, workspace, costs_out, B, H, D, n_fields, dt, sigma_start, sigma_gp, sigma_goal,
                             sigma_coll, delta, trust_region, step_size, stream);
        if (rc) return rc;
    }
    return mpb_check_launch("mpb_gpmp2_step");
}
