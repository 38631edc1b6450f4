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
// Two waves per particle: the elimination runs from BOTH ends of the chain towards the middle row m = (H-1)/2 (the
// bottom-up wave is the same recursion on the reversed chain, whose coupling block is U^T), the merge row collects
// both Schur complements and both right-hand-side carries, and the substitution runs outwards from it in both
// directions at once -- half the sequential depth.  The 2D x 2D blocks are 16 x 16 fp64 tiles held in registers
// (MFMA C layout); W_t (upper triangle: it is symmetric) and z_t go to a caller-provided workspace.  All arithmetic is fp64: the weights reach 1/sigma^2 = 1e10 (gpmp2.py:32-35)
// and fp32 Cholesky at that conditioning is not reproducible; storage stays fp32.
#include <stdlib.h>
#include <string.h>

#include "mpb_common.h"
#include "mpb_geom.h"
#include "mpb_gpmp2.h"
static_assert(MPB_GP_MAX_FIELDS == MPB_MAX_FIELDS, "mpb_gpmp2.h mirrors mpb_geom.h");

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define GP_N 16            // padded block size (2D <= 16)
#define GP_LD 17           // LDS leading dimension (fp64 words)
#define GP_MAXH MPB_MAX_H

// ------------------------------------------------------------------------------------------------
// linearisation of the collision factor: jac[b][t][0..D) = h_t = -d c_t/d q, jac[b][t][D] = c_t
// (t = 0 is excluded from the collision factor: traj_range [1, None]).  One wave per particle, lane = waypoint.
// ------------------------------------------------------------------------------------------------
// MODEL: compile-time robot model whose gradient walk the kernel runs (0: the table-driven walk); the geometry's tag is
// re-checked on the device and a mismatch poisons the rows (NaN) instead of mis-reading the buffer.
template <int MODEL>
__device__ __forceinline__ float gp_point_grad(const GeomView& G, bool ug, const unsigned* gridw, const float4* otab,
                                               const float (&q)[MPB_MAX_DOF], float (&dq)[MPB_MAX_DOF]) {
    if (MODEL == PandaModel::ID) {
        if (G.model == PandaModel::ID && ug) return waypoint_cost_grid_grad_model<PandaModel>(G, gridw, otab, q, dq);
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = __uint_as_float(0x7FC00000u);
        return __uint_as_float(0x7FC00000u);
    }
    return ug ? waypoint_cost_grid_grad(G, gridw, otab, q, dq) : waypoint_cost<true>(G, q, dq);
}

// INTERP: the interpolated-Jacobian path exists in the instantiation (its arrays cost ~40 registers the plain
// linearisation does not need); WPE: waves per SIMD the register allocation aims at.
template <int MODEL, bool INTERP, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void gpmp2_linearize_kernel(const float* __restrict__ x, const float* __restrict__ geom,
                                                              float* __restrict__ jac_all, int B, int H, int D, int n_interp_rt) {
    const int n_interp = INTERP ? n_interp_rt : 0;
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
            load_row_prefix<MPB_MAX_DOF>(row, D, true, q);       // (rows of 2D floats: always 8-byte aligned)
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) {
                dq[i] = 0.f;
                gnext[i] = 0.f;
            }
            float c = 0.f;
            if (active && t >= 1) c = gp_point_grad<MODEL>(G, ug, gridw, otab, q, dq);
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
                        gp_point_grad<MODEL>(G, ug, gridw, otab, qi, dqi);
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
                if (D == 7) {             // 8 floats per row, 32-byte aligned: two 16-byte stores
                    float4* o4 = reinterpret_cast<float4*>(o);
                    o4[0] = make_float4(-rs * dq[0], -rs * dq[1], -rs * dq[2], -rs * dq[3]);
                    o4[1] = make_float4(-rs * dq[4], -rs * dq[5], -rs * dq[6], rs * c);
                } else {
#pragma unroll
                    for (int i = 0; i < MPB_MAX_DOF; ++i)
                        if (i < D) o[i] = -rs * dq[i];
                    o[D] = rs * c;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// local SUM over particles of diag(A^T K A) (quirk Q9 needs its batch mean).  grid = H, block = 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpmp2_diag_kernel(const float* __restrict__ jac, double* __restrict__ diag_sum,
                                                         double* __restrict__ diag_mean, int B, int H, int D, int F, double dt,
                                                         double ks, double kgp, double kg, double kc) {
    const int t = blockIdx.x;
    const int dim = 2 * D;
    __shared__ double red[4][MPB_MAX_DOF];
    double acc[MPB_MAX_DOF];
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) acc[i] = 0.0;
    for (int f = 0; f < F; ++f)
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            const float* o = jac + (((size_t)f * B + b) * H + t) * (D + 1);
            if (D == 7) {            // a row is 8 floats, 32-byte aligned: two 16-byte loads instead of seven 4-byte ones
                const float4 a = reinterpret_cast<const float4*>(o)[0], c = reinterpret_cast<const float4*>(o)[1];
                acc[0] += (double)a.x * (double)a.x; acc[1] += (double)a.y * (double)a.y; acc[2] += (double)a.z * (double)a.z;
                acc[3] += (double)a.w * (double)a.w; acc[4] += (double)c.x * (double)c.x; acc[5] += (double)c.y * (double)c.y;
                acc[6] += (double)c.z * (double)c.z;
            } else {
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i)
                    if (i < D) acc[i] += (double)o[i] * (double)o[i];
            }
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
        if (diag_mean != nullptr) diag_mean[(size_t)t * dim + i] = s * (1.0 / (double)B);   // (one GPU: the batch mean of quirk Q9 at once)
    }
}

// ------------------------------------------------------------------------------------------------
// block-tridiagonal solve + update, two waves per particle (one when H < 4).
//
// U = -Phi^T Qi is (2x2) (x) I_D, so F_t = S_t^-1 U is a combination of column blocks of W_t = S_t^-1 and
//   S_{t+1} = D_{t+1} - U^T W_t U,   z_t = W_t r_t,   r_{t+1} = g_{t+1} - U^T z_t,
//   dtheta_t = z_t - W_t (U dtheta_{t+1})
// for the top-down wave (rows 0 .. m); the bottom-up wave (rows H-1 .. m+1) runs the same formulas with U^T in place
// of U and t-1 in place of t+1.  Row m: S_m = D_m - U^T W_{m-1} U - U W_{m+1} U^T, r_m likewise, dtheta_m = S_m^-1 r_m.
// The only dense operation per waypoint is the SPD inverse W_t: blocked Gauss-Jordan (no pivoting: the
// pivot blocks of an SPD matrix are SPD) on a 16 x 16 fp64 tile with 2 x 2 pivot blocks; each block step is
// one v_mfma_f64_16x16x4_f64 update of the whole tile, operands and result in registers.
// ------------------------------------------------------------------------------------------------
// 1/x in fp64: v_rcp_f64 (2^-24 accurate on gfx950) + ONE Newton step = at most 20 x 2^-53 relative error, 1.1 x 2^-53 on
// average (scripts/rcp_accuracy.hip, profiles/r03_rcp_accuracy.txt; a second step brings the maximum to 1.0 x 2^-53).  The
// IEEE division hipcc emits costs ~40 instructions and there are 7 reciprocals per waypoint in the pivot-block inverses,
// each on the sequential chain of the block step: the second Newton step (two dependent fp64 fma) was 4 % of the whole
// iteration at C4.  What the first step leaves reaches the result below the resolution of its fp32 storage: against two
// steps the updated trajectories differ by 5e-11 of the step with the trust region and 7e-8 (one fp32 ulp) without it
// (scripts/cmp_rcp_newton.py, B = 512, H = 128, D = 7).  -DGP_RCP_NEWTON=2 restores the second step.
#ifndef GP_RCP_NEWTON
#define GP_RCP_NEWTON 1
#endif
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int i = 0; i < GP_RCP_NEWTON; ++i) r = fma(fma(-x, r, 1.0), r, r);
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

// workspace doubles per waypoint: the upper triangle of the symmetric W_t (16x16 padded, row-major packed) + z_t.
// The substitution pass is bound by this traffic at large B, so only the triangle is kept (152 instead of 272 words).
#define GP_TRI (GP_N * (GP_N + 1) / 2)
#define GP_WS_PER_T (GP_TRI + GP_N)
__device__ __forceinline__ int gp_tri(int i, int j) { return i * GP_N - ((i * (i - 1)) >> 1) + (j - i); }   // i <= j

// DT > 0: the number of degrees of freedom is a compile-time constant (loop bounds, pivot-block count and the
// position/velocity index tests fold away); DT == 0: generic.
// two waves per SIMD: the kernel wants ~250 VGPRs (per-lane coefficient tables of the tile assembly in fp64), and
// the compiler's default of one wave per SIMD leaves the matrix-core and LDS latencies of the pivot steps exposed
// (measured at C4: 0.94 ms/iter with one wave, 0.63 with two; three or four only fit with spills and are slower)
#ifdef MPB_GP_STAMPS   // diagnostic build only: cycles per phase of the elimination loop, wave 0 of block 0
__device__ unsigned long long gp_phase_cycles[8];
extern "C" int mpb_debug_read_gp_phases(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(gp_phase_cycles), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : 3;
}
#define GP_STAMP(i)                                                                       \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long t_;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                \
        ph[i] += t_ - tlast;                                                              \
        tlast = t_;                                                                       \
    } while (0)
#else
#define GP_STAMP(i)
#endif
#ifndef MPB_GP_WAVES
#define MPB_GP_WAVES 2
#endif
#ifndef GP_PF
#define GP_PF 4            // register stages of the substitution pass's W_t prefetch ring (9 VGPRs each; C4: 4 stages 0.455 ms, 8: 0.476, 12: 0.488)
#endif
// SM: the collision factors enter by Sherman-Morrison instead of being assembled into S_t (round 5).  S_t = R_t + kc h h^T
// with R_t (GP blocks, Schur carry, damping, start / goal priors) well conditioned and the rank-1 collision term up to
// (sigma_gp / sigma_coll)^2 times stiffer: the explicit Gauss-Jordan inverse of S_t resolves the stiff direction to
// kappa^2 u -- step error 2e-6 at a precision ratio of 1e8, 8e-3 at 1e10, where dense fp64 Cholesky has 4e-7.  Here the
// tile that is inverted is R_t alone, with r_rest = (gradient without the collision part) as column 14 and h as column 15:
// the row operations turn them into z0 = R^-1 r_rest and y = R^-1 h for free, and
//     s = h^T y,  g = kc / (1 + kc s),   W = R^-1 - g y y^T,   z = W (r_rest + kc c h) = z0 + g (c - h^T z0) y
// -- no cancellation: kc c h never meets R^-1 on its own (z0 + kc c y - ... would lose log10(kc s) digits).  Further chained
// fields are applied one after the other the same way on the current W (y = W h_f by a matvec from LDS).  Matches the dense
// fp64 Cholesky's accuracy at every ratio (scripts/gpmp2_sm_prototype.py: 6e-9 against 1.9e-4 at 1e10, 7e-7 against O(1) at 1e12).
template <int DT, bool MULTI, bool SM>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(MPB_GP_WAVES))) void gpmp2_solve_kernel(float* __restrict__ x, const float* __restrict__ start,
                                                          const float* __restrict__ goal, const float* __restrict__ jac,
                                                          const double* __restrict__ diag_mean, double* __restrict__ work,
                                                          float* __restrict__ costs_out, int B, int H, int Drt, int Frt, int split, GpConst K) {
    const int D = DT ? DT : Drt;
    const int F = MULTI ? Frt : 1;       // MULTI == false: one collision field, the field loops fold away
    // every wave owns a private set of the per-step LDS vectors; the ex_* words are the hand-over at the merge row
    __shared__ double Sb_[2][GP_N * GP_LD];  // W_t for the matvec / next-tile reads (the inverse itself runs in registers)
    __shared__ double zv_[2][GP_N];          // z_t / scratch vector
    __shared__ double dth_[2][GP_N];         // dtheta of the previous row during the substitution pass
    __shared__ double ex_S[4][64];           // wave 1 -> wave 0: its last Schur tile (C layout)
    __shared__ double ex_r[GP_N];            // wave 1 -> wave 0: its last r carry
    __shared__ double ex_d[GP_N];            // wave 0 -> wave 1: dtheta of the merge row
    __shared__ double ex_cost;
    const int lane = threadIdx.x & 63;
    const int dir = threadIdx.x >> 6;        // 0: rows 0 .. m (top down, then the merge row m); 1: rows H-1 .. m+1 (bottom up)
    const int b = blockIdx.x;
    const int dim = 2 * D;
    const bool aug = DT ? (2 * DT <= 14) : (dim <= 14);   // room for the right-hand side as a column of the 16 x 16 tile
    const int m = split ? (H - 1) >> 1 : H - 1;   // merge row; split == 0 (64-thread block): plain top-down sweep, nothing to merge
    const int nst = dir ? (H - 1 - m) : m;   // plain elimination steps of this wave (wave 0 adds the merge step)
    const int nrows = nst + 1;               // rows this wave touches: its own and, for wave 1, the merge row as neighbour
    double* zv = zv_[dir];
    double* dth = dth_[dir];
    const double dt = K.dt;
    // 2x2 GP coefficient matrices (Kronecker with I_D)
    const double a = 12.0 / (dt * dt * dt) * K.kgp, bq = -6.0 / (dt * dt) * K.kgp, cq = 4.0 / dt * K.kgp;  // Qi
    const double p00 = a, p01 = 6.0 / (dt * dt) * K.kgp, p11 = cq;                                            // Phi^T Qi Phi
    // U = -Phi^T Qi = -[[a, bq],[a dt + bq, bq dt + cq]]   (U[c][e]: c,e in {pos, vel}).  The bottom-up wave runs the
    // same recursion on the reversed chain, whose super-diagonal block is U^T: it swaps the off-diagonal coefficients
    const double u00 = -a, u11 = -(bq * dt + cq);
    const double u01 = dir ? -(a * dt + bq) : -bq, u10 = dir ? -bq : -(a * dt + bq);
    double* wW = work + (size_t)b * H * GP_WS_PER_T;
    float* xb = x + (size_t)b * H * dim;
    const float* jb = jac + (size_t)b * H * (D + 1);
    double cost = 0.0;
    // element ownership for the 16x16 tile = the C/D layout of v_mfma_f64_16x16x4_f64: lane (lk, li) holds rows
    // lk + 4q (q = 0..3) of column li.  The tile stays in registers from its assembly through the whole inverse.
    const int li = lane & 15, lk = lane >> 4;
    f64x4 Snext = {0.0, 0.0, 0.0, 0.0};  // -(U^T W U) of the previous step, same layout
    double rcarry = 0.0;  // lane < dim: r contribution carried from the previous step (gnext - U^T z)

    // per-element constants of the S assembly (element q of this lane: row lk + 4q, column li)
    double asm_g1[4], asm_g2[4], asm_dg[4], asm_pp[4], asm_idpad[4];
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
        asm_g1[q] = same ? (ip ? (jp ? p00 : p01) : (jp ? p01 : p11)) : 0.0;   // Phi^T Qi Phi block (t < H-1)
        asm_g2[q] = same ? (ip ? (jp ? a : bq) : (jp ? bq : cq)) : 0.0;         // Qi block (t > 0)
        asm_dg[q] = (in && i == j) ? 1.0 : 0.0;
        asm_pp[q] = (in && ip && jp) ? 1.0 : 0.0;
        asm_hi[q] = (i < D) ? i : 0;
        asm_idpad[q] = (!in && i == j) ? 1.0 : 0.0;   // the identity of the padding rows / columns
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
        const double mm = in ? -1.0 : 0.0;
        nt_c[q][0] = mm * uca0 * ueb0;   // W[ii][jj]
        nt_c[q][1] = mm * uca0 * ueb1;   // W[ii][jj + D]
        nt_c[q][2] = mm * uca1 * ueb0;   // W[ii + D][jj]
        nt_c[q][3] = mm * uca1 * ueb1;   // W[ii + D][jj + D]
        nt_off[q] = ii * GP_LD + jj;
    }

    // packed-triangle offsets: the elements this lane stores (row lk + 4q <= column li), and row `lane` of W for the
    // substitution pass (element (r, j) lives at (min, max))
    int tri_st[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tri_st[q] = (lk + 4 * q <= li) ? gp_tri(lk + 4 * q, li) : 0;
    // k-th row of this wave's sweep
    const int t_first = dir ? H - 1 : 0, t_inc = dir ? -1 : 1;
    // software prefetch: the rows of x, of the Jacobian and of the damping two steps ahead are loaded so that their
    // global latency hides behind the inverse of the current step instead of sitting on the sequential critical path
    float xr0 = (lane < dim) ? xb[t_first * dim + lane] : 0.f;
    float xr1 = (lane < dim && nrows > 1) ? xb[(t_first + t_inc) * dim + lane] : 0.f;
    float jr = (lane <= D) ? jb[t_first * (D + 1) + lane] : 0.f;
    float jr1 = (lane <= D && nrows > 1) ? jb[(t_first + t_inc) * (D + 1) + lane] : 0.f;
    double dm0 = K.trust ? diag_mean[(size_t)t_first * dim + asm_di] : 0.0;
    double dm1 = (K.trust && nrows > 1) ? diag_mean[(size_t)(t_first + t_inc) * dim + asm_di] : 0.0;
    const int ksteps = dir ? nst : nst + 1;              // wave 0 also runs the merge step
#ifdef MPB_GP_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    double z_last = 0.0, x_last = 0.0;                   // z and x of the last row this wave eliminated (the merge row for wave 0)
    for (int k = 0; k < ksteps; ++k) {
        const int t = t_first + t_inc * k;               // actual row
        const bool merge = (dir == 0) && (k == m);       // wave-uniform
        if (merge) __syncthreads();                      // wave 1 has published ex_S / ex_r (it passes this barrier after its loop)
        // ---- x_t, the neighbour row, h_t from the prefetched registers (lane < dim: x element `lane`; lane <= D: h_t and
        //      c_t); issue the loads of step k+2.  Everything another lane needs of them travels through the lane
        //      crossbar (ds_bpermute on the fp32 values) -- no LDS staging, no write -> read round trip on the chain
        const double x0 = (double)xr0;
        float hf[MPB_MAX_FIELDS];
        hf[0] = (t > 0) ? jr : 0.f;                      // row 0 takes no collision factor
#pragma unroll
        for (int f = 1; f < MPB_MAX_FIELDS; ++f)         // further chained fields: not prefetched (the single-field path stays lean)
            hf[f] = (f < F && t > 0 && lane <= D) ? jb[(size_t)f * B * H * (D + 1) + t * (D + 1) + lane] : 0.f;
        const int t2 = t + 2 * t_inc;
        const bool has2 = k + 2 < nrows;
        const float xr2 = (lane < dim && has2) ? xb[t2 * dim + lane] : 0.f;
        const float jr2 = (lane <= D && has2) ? jb[t2 * (D + 1) + lane] : 0.f;
        const double dm2 = (K.trust && has2) ? diag_mean[(size_t)t2 * dim + asm_di] : 0.0;
        GP_STAMP(0);
        // ---- GP factor between this row and the neighbour: e = x_hi - Phi x_lo (hi = the later of the two rows).
        //      Its gradient splits into Phi^T Qi e (row lo) and -Qi e (row hi): one part is this row's, the other is
        //      carried to the neighbour.  The merge row receives both of its factors through the carries.
        double own_i = 0.0, gnext = 0.0;
        if (!merge) {
            const int partner = (lane < D) ? lane + D : lane - D;    // position <-> velocity of the same dof
            // (earlier / later of the two rows chosen on the fp32 values: two selects instead of eight on the doubles)
            const float xlo_f = dir ? xr1 : xr0, xhi_f = dir ? xr0 : xr1;
            const double lo_p = (double)__shfl(xlo_f, partner, 64), hi_p = (double)__shfl(xhi_f, partner, 64);   // partner element
            if (lane < dim) {
                const bool pos = lane < D;
                const double lo_o = (double)xlo_f, hi_o = (double)xhi_f;          // own element of the two rows
                const double ep = pos ? hi_o - (lo_o + dt * lo_p) : hi_p - (lo_p + dt * lo_o);
                const double ev = pos ? hi_p - lo_p : hi_o - lo_o;
                const double qp = a * ep + bq * ev, qv = bq * ep + cq * ev;      // Qi e
                const double qe_i = pos ? qp : qv;
                const double pqe_i = pos ? qp : dt * qp + qv;                    // Phi^T (Qi e)
                cost += pos ? ep * qp : ev * qv;
                own_i = dir ? -qe_i : pqe_i;
                gnext = dir ? pqe_i : -qe_i;
            }
        }
        GP_STAMP(1);
        // ---- S = D_t (+ Schur term carried in registers); padding rows/cols = identity.  Branch-free: the
        //      per-element coefficients (asm_*) were fixed before the loop, only the t-dependent selects remain
        f64x4 T;
        double r = 0.0;
        {
            const double first = (t == 0) ? 1.0 : 0.0, notfirst = 1.0 - first, notlast = (t < H - 1) ? 1.0 : 0.0;
            // damping of the diagonal element of this lane's column (prefetched, like x and h)
            const double dg = (K.trust ? K.delta * dm0 : K.delta) + first * K.ks + (1.0 - notlast) * K.kg;
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = (k > 0) ? Snext[q] : 0.0;
                if (merge && split) v[q] += ex_S[q][lane];
                v[q] = fma(notlast, asm_g1[q], v[q]);
                v[q] = fma(notfirst, asm_g2[q], v[q]);
                v[q] = fma(asm_dg[q], dg, v[q]);
            }
            // ---- r_t (lane < dim)
            r = (k > 0) ? rcarry : 0.0;
            if (merge && split && lane < dim) r += ex_r[lane];
            if (t == 0 && lane < dim) {
                const double es = (double)start[(size_t)b * dim + lane] - x0;
                r += K.ks * es;
                cost += K.ks * es * es;
            }
            if (t == H - 1 && lane < dim) {
                const double eg = (double)goal[(size_t)b * dim + lane] - x0;
                r += K.kg * eg;
                cost += K.kg * eg * eg;
            }
            r += own_i;
            // collision factor(s): kc h h^T on the position block, kc h c on the right-hand side
#pragma unroll
            for (int f = 0; f < MPB_MAX_FIELDS; ++f) {
                if (f < F) {
                    const double cf = (double)__shfl(hf[f], D, 64);                           // c_t of this field
                    if (!SM) {
                        const double hcol = (double)__shfl(hf[f], asm_hj, 64) * (K.kc * notfirst);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            v[q] = fma(asm_pp[q] * hcol, (double)__shfl(hf[f], asm_hi[q], 64), v[q]);
                        if (t > 0 && lane < D) r += K.kc * (double)hf[f] * cf;
                    }
                    if (t > 0 && lane == 0) cost += K.kc * cf * cf;
                }
            }
            // padding rows / columns = identity.  Outside the 2D x 2D block every term above is an exact zero (its
            // coefficient is one: asm_g1, asm_g2, asm_dg, asm_pp and the next-tile coefficients all vanish there), so the
            // tile is v plus the padding's diagonal ones -- one add per element instead of a select pair
#pragma unroll
            for (int q = 0; q < 4; ++q) T[q] = v[q] + asm_idpad[q];
            if (lane >= dim) r = 0.0;
            // the right-hand side rides along as column 14 of the tile (free whenever 2D <= 14): the row operations of the
            // Gauss-Jordan steps below turn it into z = S^-1 r -- the update of the whole 16 x 16 tile is one MFMA per block
            // step whatever its columns hold -- so z costs no instructions of its own (it used to be 14 fma with two
            // v_readlane and an LDS read each, behind the tile's trip through LDS).  Row 14 of the tile collects
            // meaningless values on the way (its A operand is read "by symmetry" from this column); nothing reads it.
            if (aug) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (lanes that do not hold an element of column 14 fetch lane 63, whose r is zero; column 14 itself
                    // is zero before -- it lies outside the block -- so the insertion is an add, not a select)
                    const int row = lk + 4 * q;
                    const double rq = __shfl(r, (li == 14 && row < dim) ? row : 63, 64);
                    T[q] += rq;
                    // (SM) h of the first field as column 15 (lane 63 holds no Jacobian entry: its hf is zero, as row 0's is)
                    if (SM) T[q] += (double)__shfl(hf[0], (li == 15 && row < D) ? row : 63, 64);
                }
            }
        }
        GP_STAMP(2);
        // ---- W = S^-1 : blocked Gauss-Jordan with 2x2 pivot blocks, entirely in registers.  Per block step K:
        //        D = (-A[:,K]) * (Pinv * A'[K,:]) + C_in,  A'[K,K] := I,  C_in := A with columns K zeroed,
        //      one v_mfma_f64_16x16x4_f64, then rows K := Pinv * A'[K,:] (this lane's own B operand).
        //      Lane maps (f64 16x16x4): A-op lane l -> [i = l&15][k = l>>4]; B-op [k = l>>4][j = l&15]; C/D: column
        //      l&15, rows (l>>4) + 4*reg.  Nothing goes through LDS:
        //        * rows k0, k0+1 are lanes lk = 2*half, 2*half+1 of register q = kb2 >> 1; the rank-2 update uses
        //          those two of the four k-slots of the MFMA (the operands of the other two are zero);
        //        * the pivot block is broadcast with v_readlane, the B operand needs rows K of column li: two shuffles;
        //        * the A operand -A[li][k0+r] is this lane's OWN register q up to a sign: the working matrix of
        //          the Gauss-Jordan inverse of a symmetric matrix satisfies M[a][b] = s M[b][a] with s = -1 when
        //          exactly one of a, b belongs to an already processed block, +1 otherwise (induction over the
        //          block steps), and T[q] = M[4q+lk][li].
        //      Against 4x4 pivots (round-1 first version) this trades 7 instead of 4 MFMAs per tile for a pivot-block
        //      inverse of 6 instead of ~60 fp64 operations that every lane repeats.  Lane masks are 0/1 fp64 factors:
        //      v_cndmask pairs in their place measured 8 % slower for the whole iteration.
        {
#pragma unroll
            for (int kb2 = 0; kb2 < 8; ++kb2) {
#ifdef GP_T_GJ_STEPS   // (tuning builds: only the first GP_T_GJ_STEPS block steps -- wrong results, timing only)
                if (2 * kb2 < dim && kb2 < GP_T_GJ_STEPS) {
#else
                if (2 * kb2 < dim) {
#endif
                    const int k0 = 2 * kb2, q = kb2 >> 1, half = kb2 & 1;
                    const double tk = T[q];
                    const double p00 = readlane_f64(tk, (2 * half) * 16 + k0), p01 = readlane_f64(tk, (2 * half) * 16 + k0 + 1);
                    const double p11 = readlane_f64(tk, (2 * half + 1) * 16 + k0 + 1);
                    const double id = fast_rcp(fma(p00, p11, -p01 * p01));
                    const bool jin = (li >= k0) && (li < k0 + 2);
                    const double notj = jin ? 0.0 : 1.0;
                    // (the two row fetches through v_permlane32_swap / v_permlane16_swap instead of the LDS crossbar: measured 2 %
                    // slower, twice -- with the reciprocal chain as it was and as it is now; round 5: only the EXCHANGE between lane rows
                    // 2 half and 2 half + 1 that the update really needs, one v_permlane16_swap per dword + selects: +2.8 % -- the fetches
                    // ride the LDS pipe for free, their replacements are vector instructions in a kernel at half of its fp64 issue peak)
                    const double am0 = fma(notj, __shfl(tk, (2 * half) * 16 + li, 64), (li == k0) ? 1.0 : 0.0);
                    const double am1 = fma(notj, __shfl(tk, (2 * half + 1) * 16 + li, 64), (li == k0 + 1) ? 1.0 : 0.0);
                    const double sel0 = (lk == 2 * half) ? 1.0 : 0.0, sel1 = (lk == 2 * half + 1) ? 1.0 : 0.0;
                    // this lane's row of Pinv * A'[K,:] with the reciprocal of the determinant factored out: everything but the
                    // last product is ready before the reciprocal is (Pinv = adj(P) / det: rows (p11, -p01) and (-p01, p00)),
                    // so ONE multiply stands between the reciprocal and the matrix instruction instead of five dependent
                    // operations (i00 = p11 id, sel0 i00 + sel1 i01, ... ): the block step is a latency chain
                    const double u0 = fma(sel0, p11, -(sel1 * p01)), u1 = fma(sel1, p00, -(sel0 * p01));
                    const double bop = fma(u0, am0, u1 * am1) * id;
                    const double act = sel0 + sel1;
                    const double aop = act * ((li < k0) ? tk : -tk);
                    f64x4 cin;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) cin[qq] = notj * T[qq];
                    T = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, cin, 0, 0, 0);
                    T[q] = fma(act, bop, (1.0 - act) * T[q]);       // rows of the pivot block := bop (act is 0 or 1: exact)
                }
            }
        }
        GP_STAMP(3);
        // W_t to LDS once (rows for z = W r, elements for the next Schur tile) and to the workspace straight from the
        // registers.  One wave: its LDS accesses execute in order, the fence only pins the compiler's order.
        double* Wl = Sb_[dir];
#pragma unroll
        for (int q = 0; q < 4; ++q) Wl[(lk + 4 * q) * GP_LD + li] = T[q];
        wave_sync();
        const double* W = Wl;
        // ---- z = W r (r_j broadcast with v_readlane); store W_t, z_t
        double* wt = wW + (size_t)t * GP_WS_PER_T;
        double zi = 0.0;
#ifdef GP_T_NO_Z       // (tuning builds: no z = W r -- wrong results, timing only)
        zi = r * W[(lane < dim ? lane : 0) * GP_LD];
        (void)wt;
#else
        {
            const int rowl = (lane < dim) ? lane : 0;
            if (aug) {
                zi = (lane < dim) ? W[rowl * GP_LD + 14] : 0.0;      // column 14 of the tile: z = S^-1 r (see the assembly)
            } else if (DT) {
#pragma unroll
                for (int j = 0; j < 2 * DT; ++j) zi = fma(W[rowl * GP_LD + j], readlane_f64(r, j), zi);
            } else {
                for (int j = 0; j < dim; ++j) zi = fma(W[rowl * GP_LD + j], readlane_f64(r, j), zi);
            }
        }
#endif
        if (SM) {
            // ---- the collision factors by Sherman-Morrison (see the kernel's header): zi is z0 = R^-1 r_rest, T / Wl hold R^-1
            const int rowl = (lane < dim) ? lane : 0;
#pragma unroll
            for (int f = 0; f < MPB_MAX_FIELDS; ++f) {
                if (f < F) {
                    double yi = 0.0;                                   // y = W h_f, element `lane` (zero beyond the block)
                    if (f == 0 && aug) {
                        yi = W[rowl * GP_LD + 15];
                    } else if (DT) {
#pragma unroll
                        for (int j = 0; j < DT; ++j) yi = fma(W[rowl * GP_LD + j], (double)readlane_f32(hf[f], j), yi);
                    } else {
                        for (int j = 0; j < D; ++j) yi = fma(W[rowl * GP_LD + j], (double)readlane_f32(hf[f], j), yi);
                    }
                    if (lane >= dim) yi = 0.0;
                    if (lane < GP_N) zv[lane] = yi;
                    const double hd = (lane < D) ? (double)hf[f] : 0.0;
                    // h^T y and h^T z over the first 16-lane row (D <= 8): four DPP steps, then lane 0's total
                    double py = hd * yi, pz = hd * zi;
                    py += dpp_f64<0xB1>(py); pz += dpp_f64<0xB1>(pz);
                    py += dpp_f64<0x4E>(py); pz += dpp_f64<0x4E>(pz);
                    py += dpp_f64<0x141>(py); pz += dpp_f64<0x141>(pz);
                    py += dpp_f64<0x140>(py); pz += dpp_f64<0x140>(pz);
                    const double sdot = readlane_f64(py, 0), hz = readlane_f64(pz, 0);
                    const double cf = (double)readlane_f32(hf[f], D);
                    const double gfac = K.kc * fast_rcp(fma(K.kc, sdot, 1.0));
                    zi = fma(gfac * (cf - hz), yi, zi);
                    wave_sync();
                    const double ycol = (li < dim) ? -gfac * zv[li] : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        T[q] = fma(ycol, zv[lk + 4 * q], T[q]);       // W -= g y y^T (rows beyond the block: y = 0)
                        Wl[(lk + 4 * q) * GP_LD + li] = T[q];
                    }
                    wave_sync();
                }
            }
        }
#ifndef GP_T_SKIP_STORE   // (tuning builds: elimination without the workspace traffic)
        if (lane < dim) wt[GP_TRI + lane] = zi;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lk + 4 * q <= li) wt[tri_st[q]] = T[q];
#endif
        z_last = zi;
        x_last = x0;
        if (!merge) {
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
            // carry to the neighbour's r = gnext - U^T z (z of the partner lane through the crossbar)
            {
                const bool ip = lane < D;
                const double zpart = __shfl(zi, ip ? lane + D : lane - D, 64);
                const double zp = ip ? zi : zpart, zvv = ip ? zpart : zi;
                rcarry = gnext - (ip ? u00 * zp + u10 * zvv : u01 * zp + u11 * zvv);
            }
            wave_sync();                                           // W reads stay ahead of the next step's tile write
        }
        GP_STAMP(4);
        xr0 = xr1; xr1 = xr2; jr = jr1; jr1 = jr2; dm0 = dm1; dm1 = dm2;
        GP_STAMP(5);
    }
#ifdef MPB_GP_STAMPS
    if (b == 0 && dir == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) gp_phase_cycles[i] = ph[i];
#endif
    // ---- hand-over at the merge row: wave 1 publishes its last Schur tile and r carry (first barrier, taken by wave 0
    //      at the top of its merge step); wave 0 publishes dtheta_m = z_m and updates x_m (second barrier).
    if (dir) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ex_S[q][lane] = Snext[q];
        if (lane < dim) ex_r[lane] = rcarry;
        __syncthreads();
    } else {
        if (lane < dim) {
            ex_d[lane] = z_last;                                        // dtheta_m = z_m
            xb[m * dim + lane] = (float)(x_last + K.step * z_last);
        }
    }
    __syncthreads();
    // ---- substitution away from the merge row and update: dtheta_t = z_t - W_t (U dtheta_prev), prev = the row one
    //      step closer to the merge row.  Row `lane` of the next W, z and x are fetched while the current step runs:
    //      the workspace (B*H*2.2 KB) does not stay in cache, and an un-prefetched global round trip per waypoint
    //      would sit on the sequential critical path
#ifdef GP_T_SKIP_SUBST    // (tuning builds: elimination only)
    if (costs_out != nullptr && dir == 0 && lane == 0) costs_out[b] = (float)cost;
    return;
#endif
    const bool rowlane = lane < dim;
    const int rl = rowlane ? lane : 0;          // idle lanes shadow row 0: unconditional loads, no exec-mask branches
    if (rowlane) {   // v = U dtheta_m
        const bool ip = lane < D;
        const int ii = ip ? lane : lane - D;
        const double dp = ex_d[ii], dv = ex_d[ii + D];
        zv[lane] = ip ? u00 * dp + u01 * dv : u10 * dp + u11 * dv;
    }
    wave_sync();
    if (nst > 0) {
        // The W_t records (152 doubles per waypoint and particle) do not stay in cache (352 MB at C4) and every step needs
        // its own.  Round 1 had each lane gather its row of the packed triangle straight from global memory, one step
        // ahead: 16 load instructions per step touching ~14 cache lines each -- the pass ran at 0.21 of the solve's
        // 0.60 ms (measured by elimination), bound by the address coalescer and one HBM latency per waypoint.  Now the
        // record is fetched as it lies (two 16-byte loads per lane, 1.2 KB contiguous), GP_PF steps ahead through a
        // ring of register stages (the elimination's registers are dead here), dropped into the wave's LDS tile and
        // the row is gathered from there.
        constexpr int NW = DT ? 2 * DT : GP_N;                 // entries of a row of W actually used
        int tri_ld[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) tri_ld[j] = gp_tri(min(rl, j), max(rl, j));
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 sa[GP_PF], sb[GP_PF];
        float xst[GP_PF];
        constexpr int REC2 = GP_WS_PER_T / 2;                  // 76 16-byte words
        auto fetch = [&](int k, d2& a0, d2& b0, float& xdst) {
            const int t = t_first + t_inc * k;
            const d2* rec = reinterpret_cast<const d2*>(wW + (size_t)t * GP_WS_PER_T);
            a0 = rec[lane];
            b0 = rec[64 + (lane < REC2 - 64 ? lane : 0)];
            xdst = xb[t * dim + rl];
        };
#pragma unroll
        for (int u = 0; u < GP_PF; ++u)
            if (nst - 1 - u >= 0) fetch(nst - 1 - u, sa[u], sb[u], xst[u]);
        double* slot = Sb_[dir];                               // the wave's tile buffer (272 doubles), free in this pass
        for (int kk = nst - 1; kk >= 0; kk -= GP_PF) {
#pragma unroll
            for (int u = 0; u < GP_PF; ++u) {
                const int k = kk - u;
                if (k < 0) break;                              // wave-uniform
                const int t = t_first + t_inc * k;
                reinterpret_cast<d2*>(slot)[lane] = sa[u];
                if (lane < REC2 - 64) reinterpret_cast<d2*>(slot)[64 + lane] = sb[u];
                const float xc = xst[u];
                if (k - GP_PF >= 0) fetch(k - GP_PF, sa[u], sb[u], xst[u]);        // this stage's next occupant
                wave_sync();
                double d = slot[GP_TRI + rl];
                if (rowlane) {
#pragma unroll
                    for (int j = 0; j < NW; ++j)
                        if (j < dim) d -= slot[tri_ld[j]] * zv[j];                 // zv holds U dtheta_prev
                }
                wave_sync();
                if (rowlane) {
                    dth[lane] = d;
                    xb[t * dim + lane] = (float)((double)xc + K.step * d);
                }
                wave_sync();
                if (rowlane) {   // v = U dtheta_t for the next row
                    const bool ip = lane < D;
                    const int ii = ip ? lane : lane - D;
                    const double dp = dth[ii], dv = dth[ii + D];
                    zv[lane] = ip ? u00 * dp + u01 * dv : u10 * dp + u11 * dv;
                }
                wave_sync();
            }
        }
    }
    cost = wave_sum_f64(cost);
    if (dir && lane == 0) ex_cost = cost;
    __syncthreads();
    if (costs_out != nullptr && dir == 0 && lane == 0) costs_out[b] = (float)(split ? cost + ex_cost : cost);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
static bool gp_shape_ok(int B, int H, int D) { return B >= 0 && H >= 2 && H <= GP_MAXH && D >= 1 && D <= MPB_MAX_DOF; }
// the block elimination keeps a 2D x 2D block in ONE 16 x 16 matrix-core tile: D <= 8; beyond that (D <= 12) only the low-rank form applies
#define GP_BLOCK_MAX_DOF 8

extern "C" size_t mpb_gpmp2_workspace_bytes(int B, int H, int D) {
    if (!gp_shape_ok(B, H, D)) return 0;
    const size_t jac = (size_t)MPB_MAX_FIELDS * B * H * (D + 1) * sizeof(float);   // one (h, c) set per chained field
    const size_t diag = 2 * (size_t)H * 2 * D * sizeof(double);
    // the elimination records of the block form -- or the tables and sweep records of the low-rank form (mpb_gpmp2_lr.hip), whichever is larger
    size_t fz = (size_t)B * H * GP_WS_PER_T * sizeof(double);
    const size_t lr = mpb_gpmp2_lr_ws_doubles(B, H, D) * sizeof(double);
    if (fz < lr) fz = lr;
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

extern "C" int mpb_gpmp2_linearize(const float* x, const float* geom, int geom_flags, void* workspace, int B, int H, int D,
                                   int n_interp, void* stream) {
    if (!x || !geom || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_linearize: null pointer");
    // (waypoint rows are read as 8-byte pieces, the Jacobian rows written as 16-byte pieces)
    if (((uintptr_t)x & 15u) || ((uintptr_t)workspace & 255u) || ((uintptr_t)geom & 15u))
        return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_linearize: x / geom must be 16-byte aligned and the workspace 256-byte aligned");
    if (!gp_shape_ok(B, H, D) || n_interp < 0 || n_interp > 64) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_linearize: bad shape");
    if (B == 0) return MPB_OK;
    GpWork w = gp_carve(workspace, B, H, D);
    // geom_flags (mpb_geom_flags of the host copy): low byte = compile-time robot model of EVERY chained field, bit 8 = all
    // of them grid-backed -- the model kernel needs both
#ifndef GP_LIN_WPE
#define GP_LIN_WPE 3
#endif
    const bool model = (geom_flags & 0xFF) == PandaModel::ID && (geom_flags & 0x100) && D == PandaModel::N_DOF;
#define GP_LIN(MODEL, INTERP, WPE)                                                                                           \
    hipLaunchKernelGGL((gpmp2_linearize_kernel<MODEL, INTERP, WPE>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, \
                       geom, w.jac, B, H, D, n_interp)
    if (model && n_interp == 0) GP_LIN(PandaModel::ID, false, GP_LIN_WPE);
    else if (model) GP_LIN(PandaModel::ID, true, 2);
    else if (n_interp == 0) GP_LIN(0, false, 2);
    else GP_LIN(0, true, 2);
#undef GP_LIN
    return mpb_check_launch("mpb_gpmp2_linearize");
}

extern "C" int mpb_gpmp2_diag(void* workspace, double* diag_sum_out, int B, int H, int D, int n_fields, float dt,
                              float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll, void* stream) {
    if (!workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_diag: null pointer");
    if (!gp_shape_ok(B, H, D) || n_fields < 1 || n_fields > MPB_MAX_FIELDS) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_diag: bad shape");
    GpWork w = gp_carve(workspace, B, H, D);
    double* out = diag_sum_out ? diag_sum_out : w.diag_sum;
    // workspace mode (one GPU): the LOCAL mean goes to the workspace in the same pass (what mpb_gpmp2_solve reads when it is
    // given no diag_mean); with diag_sum_out the host all-reduces the sums over the shards and passes the global mean itself
    hipLaunchKernelGGL(gpmp2_diag_kernel, dim3(H), dim3(256), 0, (hipStream_t)stream, w.jac, out,
                       diag_sum_out ? (double*)nullptr : w.diag_mean, B, H, D, n_fields, (double)dt,
                       1.0 / ((double)sigma_start * sigma_start), 1.0 / ((double)sigma_gp * sigma_gp),
                       (sigma_goal > 0.f ? 1.0 / ((double)sigma_goal * sigma_goal) : 0.0), 1.0 / ((double)sigma_coll * sigma_coll));
    return mpb_check_launch("mpb_gpmp2_diag");
}

extern "C" int mpb_gpmp2_solve(float* x, const float* start, const float* goal, const double* diag_mean, void* workspace,
                               float* costs_out, int B, int H, int D, int n_fields, float dt, float sigma_start,
                               float sigma_gp, float sigma_goal, float sigma_coll, float delta, int trust_region,
                               float step_size, void* stream) {
    if (!x || !start || !goal || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_solve: null pointer");
    if (((uintptr_t)x & 15u) || ((uintptr_t)workspace & 255u))
        return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_solve: x must be 16-byte aligned and the workspace 256-byte aligned");
    if (!gp_shape_ok(B, H, D) || n_fields < 1 || n_fields > MPB_MAX_FIELDS) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_solve: bad shape");
    if (B == 0) return MPB_OK;
    GpWork w = gp_carve(workspace, B, H, D);
    GpConst K;
    K.dt = dt;
    K.ks = 1.0 / ((double)sigma_start * sigma_start);
    K.kgp = 1.0 / ((double)sigma_gp * sigma_gp);
    K.kg = (sigma_goal > 0.f ? 1.0 / ((double)sigma_goal * sigma_goal) : 0.0);
    K.kc = 1.0 / ((double)sigma_coll * sigma_coll);
    K.delta = delta;
    K.step = step_size;
    K.trust = trust_region;
    const double* dm = trust_region ? (diag_mean ? diag_mean : w.diag_mean) : nullptr;
    // ---- which form.  The low-rank form (round 6, mpb_gpmp2_lr.hip: A0 shared by all particles and factored once per iteration,
    // a dense solve of the size of a particle's ACTIVE collision rows) wherever its LDS tile holds the rows (F (H - 1) <= 127);
    // the block elimination below otherwise.  MPB_GPMP2_FORM = lr / block forces one (tests, A/B timing); MPB_GPMP2_SM set
    // (either value) means the block form, as before round 6.   (read per call: the tests switch within one process)
    {
        const char* form_env = getenv("MPB_GPMP2_FORM");
        const bool want_block = (form_env && !strcmp(form_env, "block")) || (!form_env && getenv("MPB_GPMP2_SM") != nullptr);
        if (form_env && !strcmp(form_env, "lr") && !mpb_gpmp2_lr_ok(H, D, n_fields))
            return mpb_fail(MPB_E_UNSUPPORTED, "mpb_gpmp2_solve: MPB_GPMP2_FORM=lr, but n_fields * (H - 1) > 127");
        if (D > GP_BLOCK_MAX_DOF && (want_block || !mpb_gpmp2_lr_ok(H, D, n_fields)))
            return mpb_fail(MPB_E_UNSUPPORTED, "mpb_gpmp2_solve: more than 8 degrees of freedom need the low-rank form (n_fields * (H - 1) <= 127, H <= 128)");
        if (!want_block && mpb_gpmp2_lr_ok(H, D, n_fields)) {
            int rc = mpb_gpmp2_lr_launch(x, start, goal, w.jac, dm, w.fz, costs_out, B, H, D, n_fields, K, (hipStream_t)stream);
            if (rc) return rc;
            return mpb_check_launch("mpb_gpmp2_solve (low-rank form)");
        }
    }
    // two waves per particle (sweeps from both ends of the chain, see the kernel) halve the sequential chain: B = 256
    // -32 %; at B = 2048 the instruction throughput is the bound and both forms take the same time (measured).  The
    // one-wave form remains for chains too short to split
    static const int force_split = getenv("MPB_GPMP2_SPLIT") ? atoi(getenv("MPB_GPMP2_SPLIT")) : -1;   // tuning aid
    const int split = (H >= 4) && (force_split >= 0 ? force_split != 0 : 1);
    // Sherman-Morrison form of the collision factors (template flag SM, see the kernel): selected when the collision precision
    // exceeds the GP precision by more than 1e7 (x the number of fields) -- below that the assembled form is accurate to
    // <= 2e-6 of the step (tests) and C4 (ratio 1e6) keeps the kernel it was tuned with.  MPB_GPMP2_SM = 0 / 1 forces a form
    // (A/B timing and the tests that run both on the same system).
    const char* sm_env = getenv("MPB_GPMP2_SM");            // (read per call: the tests switch it within one process)
    const int force_sm = sm_env ? atoi(sm_env) : -1;
    const double ratio = ((double)sigma_gp / (double)sigma_coll) * ((double)sigma_gp / (double)sigma_coll) * n_fields;
    const bool sm = force_sm >= 0 ? force_sm != 0 : ratio > 1e7;
#define GP_LAUNCH_(DT, MULTI, SM)                                                                                              \
    hipLaunchKernelGGL((gpmp2_solve_kernel<DT, MULTI, SM>), dim3(B), dim3(split ? 128 : 64), 0, (hipStream_t)stream, x, start, \
                       goal, w.jac, dm, w.fz, costs_out, B, H, D, n_fields, split, K)
#define GP_LAUNCH(DT)                                      \
    if (n_fields == 1) {                                   \
        if (sm) GP_LAUNCH_(DT, false, true);               \
        else GP_LAUNCH_(DT, false, false);                 \
    } else {                                               \
        if (sm) GP_LAUNCH_(DT, true, true);                \
        else GP_LAUNCH_(DT, true, false);                  \
    }
    switch (D) {
        case 2: GP_LAUNCH(2); break;
        case 3: GP_LAUNCH(3); break;
        case 7: GP_LAUNCH(7); break;
        default: GP_LAUNCH(0); break;
    }
#undef GP_LAUNCH
#undef GP_LAUNCH_
    return mpb_check_launch("mpb_gpmp2_solve");
}

extern "C" int mpb_gpmp2_step(float* x, const float* start, const float* goal, const float* geom, int geom_flags, void* workspace,
                              float* costs_out, int B, int H, int D, float dt, float sigma_start, float sigma_gp,
                              float sigma_goal, float sigma_coll, float delta, int trust_region, float step_size,
                              int n_iters, int n_interp, int n_fields, void* stream) {
    if (!x || !start || !goal || !geom || !workspace) return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_step: null pointer");
    if (!gp_shape_ok(B, H, D) || n_iters < 0 || n_fields < 1 || n_fields > MPB_MAX_FIELDS)
        return mpb_fail(MPB_E_INVALID, "mpb_gpmp2_step: bad shape");
    if (B == 0) return MPB_OK;
    for (int it = 0; it < n_iters; ++it) {
        int rc = mpb_gpmp2_linearize(x, geom, geom_flags, workspace, B, H, D, n_interp, stream);
        if (rc) return rc;
        if (trust_region) {
            rc = mpb_gpmp2_diag(workspace, nullptr, B, H, D, n_fields, dt, sigma_start, sigma_gp, sigma_goal, sigma_coll, stream);
            if (rc) return rc;
        }
        rc = mpb_gpmp2_solve(x, start, goal, nullptr, workspace, costs_out, B, H, D, n_fields, dt, sigma_start, sigma_gp, sigma_goal,
                             sigma_coll, delta, trust_region, step_size, stream);
        if (rc) return rc;
    }
    return mpb_check_launch("mpb_gpmp2_step");
}
