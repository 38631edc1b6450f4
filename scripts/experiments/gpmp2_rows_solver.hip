// EXPERIMENT (round 2, not built, not shipped): GPMP2's block-tridiagonal solve with FOUR chains per wave.
// Result on MI355X: bit-for-bit green on every GPMP2 test (51 passed), but SLOWER than the MFMA-tile solver of
// csrc/mpb_gpmp2.hip -- C4 (B=2048, H=128, D=7) 0.889 vs 0.641 ms / iteration, B=256 0.759 vs 0.248 ms: with one wave per
// SIMD every pivot step is two dependent LDS round trips + a reciprocal chain (~650 cycles x 14 pivots per waypoint),
// and there is no second wave to hide them.  Kept as the record behind DESIGN.md's GPMP2 section.
//
// (original header) GPMP2's block-tridiagonal solve (see mpb_gpmp2.hip for the algebra) with FOUR chains per wave.
//
// The first solver (gpmp2_solve_kernel, mpb_gpmp2.hip) spends one wave on one chain: the 2D x 2D block lives as a
// 16 x 16 fp64 MFMA tile spread over the 64 lanes, and every step of its Gauss-Jordan inverse is a ~70-instruction
// dependent chain of read-lanes, shuffles and one matrix instruction for 196 useful multiply-adds -- 664 instructions
// per waypoint, two waves per SIMD (250 VGPRs), the fp64 VALU ~60 % busy: 0.62 ms per iteration at C4.
// Here a chain owns one 16-lane ROW of the wave and lane j of the row holds COLUMN j of the block in registers
// (2D <= 16 doubles).  A wave carries the two sweep directions of two particles, so the hand-over at the merge row stays
// inside the wave; B = 2048 particles are 1024 waves, one per SIMD.  The arithmetic per step is the plain one:
//   * Gauss-Jordan inverse in place, 2D pivot steps: the pivot column is published through LDS by its lane and read
//     back as a broadcast by the 16 lanes of the row; every lane updates its own column with 2D independent fma;
//   * z = W r, the next Schur complement -(U^T W U) (U = (2x2) (x) I couples a column only with its position/velocity
//     partner, fetched through LDS) and the carried right-hand side: 2D fma each;
//   * the substitution pass re-reads the upper triangle of W_t (same workspace layout as the first solver),
//     completes the symmetric row through LDS and runs one 2D-term dot product per lane.
// No matrix instruction: at 14 x 14 the MFMA tile was 23 % padding and its operands cost more lane traffic than the
// multiply-adds they fed.  Compile-time D in {2, 3, 7}; other sizes keep the first solver.
#include "mpb_common.h"

struct GpConst {       // (mirrors mpb_gpmp2.hip)
    double dt, ks, kgp, kg, kc, delta, step;
    int trust;
};

#define GP_N 16
#define GP_TRI (GP_N * (GP_N + 1) / 2)
#define GP_WS_PER_T (GP_TRI + GP_N)
#define ROWS_WAVES 4                       // waves per block; each wave = 2 particles x 2 directions
#define MPB_MAX_FIELDS_R 4

__device__ __forceinline__ double rows_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ int rows_tri(int i, int j) { return i * GP_N - ((i * (i - 1)) >> 1) + (j - i); }   // i <= j
__device__ __forceinline__ void rows_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <int DT, bool MULTI>
__global__ __launch_bounds__(64 * ROWS_WAVES) void gpmp2_solve_rows_kernel(
    float* __restrict__ x, const float* __restrict__ start, const float* __restrict__ goal, const float* __restrict__ jac,
    const double* __restrict__ diag_mean, double* __restrict__ work, float* __restrict__ costs_out, int B, int H, int Frt,
    GpConst K) {
    constexpr int D = DT, DIM = 2 * DT;
    const int F = MULTI ? Frt : 1;
    // per wave: a 16-double column slot per lane, the published pivot column / broadcast vectors per row
    __shared__ __attribute__((aligned(16))) double colbuf_[ROWS_WAVES][64 * GP_N];   // 8 KB per wave
    __shared__ __attribute__((aligned(16))) double pc_[ROWS_WAVES][4 * GP_N];
    __shared__ __attribute__((aligned(16))) double vb_[ROWS_WAVES][4 * GP_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 4, j = lane & 15;
    const int dir = row & 1;                                   // 0: rows 0 .. m then the merge row; 1: rows H-1 .. m+1
    const int b = (blockIdx.x * ROWS_WAVES + wave) * 2 + (row >> 1);
    const bool pvalid = b < B;
    const int bb = pvalid ? b : 0;                             // idle rows shadow particle 0 and store nothing
    double* colbuf = colbuf_[wave];
    double* mycol = colbuf + lane * GP_N;                      // this lane's column slot
    double* pc = pc_[wave] + row * GP_N;                       // row-shared
    double* vb = vb_[wave] + row * GP_N;
    const bool valid = j < DIM;                                // lanes 2D .. 15 of a row are padding
    const bool jpos = j < D;
    const int jj = jpos ? j : j - D;
    const int jpart = valid ? (jpos ? j + D : j - D) : j;      // position <-> velocity partner column
    const int dim = DIM;
    const int m = (H >= 4) ? (H - 1) >> 1 : H - 1;             // merge row (H < 4: plain top-down sweep by direction 0)
    const bool split = H >= 4;
    const int nst = dir ? (split ? H - 1 - m : 0) : m;         // plain elimination steps of this row
    const int nst_max = split ? ((H - 1 - m) > m ? (H - 1 - m) : m) : m;
    const double dt = K.dt;
    const double a = 12.0 / (dt * dt * dt) * K.kgp, bq = -6.0 / (dt * dt) * K.kgp, cq = 4.0 / dt * K.kgp;      // Qi
    const double p01 = 6.0 / (dt * dt) * K.kgp;                                                                // Phi^T Qi Phi = [[a, p01],[p01, cq]]
    const double u00 = -a, u11 = -(bq * dt + cq);
    const double u01 = dir ? -(a * dt + bq) : -bq, u10 = dir ? -bq : -(a * dt + bq);
    // same-dof entries of column j: rows jj (position) and jj + D (velocity)
    const double gA1 = jpos ? a : p01, gA2 = jpos ? a : bq;    // row jj     : Phi^T Qi Phi part (t < H-1), Qi part (t > 0)
    const double gB1 = jpos ? p01 : cq, gB2 = jpos ? bq : cq;  // row jj + D
    // sum_e U[e][b] W[.][j' + eD] = ue_own * (own column) + ue_par * (partner column)
    const double ue_own = jpos ? u00 : u11, ue_par = jpos ? u10 : u01;
    double* wW = work + (size_t)bb * H * GP_WS_PER_T;
    float* xb = x + (size_t)bb * H * dim;
    const float* jb = jac + (size_t)bb * H * (D + 1);
    const int t_first = dir ? H - 1 : 0, t_inc = dir ? -1 : 1;
    double cost = 0.0;
    double Sn[DIM];                                            // column j of -(U^T W U) of the previous step
#pragma unroll
    for (int i = 0; i < DIM; ++i) Sn[i] = 0.0;
    double rcarry = 0.0;
    for (int i = 0; i < GP_N; ++i) mycol[i] = 0.0;             // the constant-part slot: all zero but two entries per step
    rows_sync();

    // one elimination step of row t for the lanes of this chain.  merge: the row takes both carries.
    auto step = [&](int t, bool merge, const double (&Sx)[DIM], double rx) {
        const int tn = t + t_inc;                                              // the neighbour that the GP factor of this step couples
        const float xo = valid ? xb[t * dim + j] : 0.f, xp = valid ? xb[t * dim + jpart] : 0.f;
        double own_i = 0.0, gnext = 0.0;
        if (!merge && valid) {
            const float xno = xb[tn * dim + j], xnp = xb[tn * dim + jpart];
            const double lo_o = dir ? xno : xo, hi_o = dir ? xo : xno, lo_p = dir ? xnp : xp, hi_p = dir ? xp : xnp;
            const double ep = jpos ? hi_o - (lo_o + dt * lo_p) : hi_p - (lo_p + dt * lo_o);
            const double ev = jpos ? hi_p - lo_p : hi_o - lo_o;
            const double qp = a * ep + bq * ev, qv = bq * ep + cq * ev;          // Qi e
            const double qe_i = jpos ? qp : qv;
            const double pqe_i = jpos ? qp : dt * qp + qv;                       // Phi^T (Qi e)
            cost += jpos ? ep * qp : ev * qv;
            own_i = dir ? -qe_i : pqe_i;
            gnext = dir ? pqe_i : -qe_i;
        }
        // ---- column j of S_t = carried Schur term + constant part (through the lane's LDS slot: its two same-dof
        //      entries sit at lane-dependent rows) + collision rank-1
        const double first = (t == 0) ? 1.0 : 0.0, notfirst = 1.0 - first, notlast = (t < H - 1) ? 1.0 : 0.0;
        double dg = K.trust ? K.delta * diag_mean[(size_t)t * dim + (valid ? j : 0)] : K.delta;
        dg += first * K.ks + (1.0 - notlast) * K.kg;
        const double vA = notlast * gA1 + notfirst * gA2 + (jpos ? dg : 0.0);
        const double vB = notlast * gB1 + notfirst * gB2 + (jpos ? 0.0 : dg);
        if (valid) { mycol[jj] = vA; mycol[jj + D] = vB; }
        rows_sync();
        double S[DIM];
#pragma unroll
        for (int i = 0; i < DIM; i += 2) {
            const double2 c2 = *reinterpret_cast<const double2*>(mycol + i);
            S[i] = Sx[i] + c2.x;
            S[i + 1] = Sx[i + 1] + c2.y;
        }
        double r = rx + own_i;
        if (t == 0 && valid) {
            const double es = (double)start[(size_t)bb * dim + j] - (double)xo;
            r += K.ks * es;
            cost += K.ks * es * es;
        }
        if (t == H - 1 && valid) {
            const double eg = (double)goal[(size_t)bb * dim + j] - (double)xo;
            r += K.kg * eg;
            cost += K.kg * eg * eg;
        }
        if (t > 0) {
#pragma unroll
            for (int f = 0; f < MPB_MAX_FIELDS_R; ++f) {
                if (f < F) {
                    const float* hrow = jb + (size_t)f * B * H * (D + 1) + (size_t)t * (D + 1);
                    const double hj = jpos ? (double)hrow[j] : 0.0;
                    const double cf = (double)hrow[D];
                    const double khj = K.kc * hj;
#pragma unroll
                    for (int i = 0; i < D; ++i) S[i] = fma(khj, (double)hrow[i], S[i]);
                    r = fma(khj, cf, r);
                    if (j == 0) cost += K.kc * cf * cf;
                }
            }
        }
        if (!valid) r = 0.0;
        // ---- W = S^-1: Gauss-Jordan in place, column per lane
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            if (j == k) {
#pragma unroll
                for (int i = 0; i < DIM; i += 2) *reinterpret_cast<double2*>(pc + i) = make_double2(S[i], S[i + 1]);
            }
            rows_sync();
            double c[DIM];
#pragma unroll
            for (int i = 0; i < DIM; i += 2) {
                const double2 c2 = *reinterpret_cast<const double2*>(pc + i);
                c[i] = c2.x; c[i + 1] = c2.y;
            }
            const double p = rows_rcp(c[k]);
            const bool me = (j == k);
            const double tk = me ? p : S[k] * p;
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
                if (i == k) continue;
                S[i] = fma(-c[i], tk, me ? 0.0 : S[i]);
            }
            S[k] = tk;
            rows_sync();                                    // the reads of pc stay ahead of the next pivot's write
        }
        // ---- z = W r: z_j = sum_i W[i][j] r_i (W symmetric), r broadcast through LDS
        if (valid) vb[j] = r;
        // column j of W to the lane's slot for the partner (next Schur term); restored to zero below
#pragma unroll
        for (int i = 0; i < DIM; i += 2) *reinterpret_cast<double2*>(mycol + i) = make_double2(S[i], S[i + 1]);
        rows_sync();
        double z = 0.0;
#pragma unroll
        for (int i = 0; i < DIM; i += 2) {
            const double2 r2 = *reinterpret_cast<const double2*>(vb + i);
            z = fma(S[i], r2.x, z);
            z = fma(S[i + 1], r2.y, z);
        }
        // ---- W_t (upper triangle: rows i <= j of this lane's column) and z_t to the workspace
        if (pvalid && valid) {
            double* wt = wW + (size_t)t * GP_WS_PER_T;
#pragma unroll
            for (int i = 0; i < DIM; ++i)
                if (i <= j) wt[rows_tri(i, j)] = S[i];
            wt[GP_TRI + j] = z;
        }
        // ---- next Schur column: -(U^T W U)[i][j] = -(U[0][a] T[i'] + U[1][a] T[i' + D]),
        //      T[i] = sum_e U[e][b] W[i][j' + eD] = ue_own * own[i] + ue_par * partner[i]
        double T[DIM];
        {
            const double* pcol = colbuf + ((lane & ~15) + jpart) * GP_N;
#pragma unroll
            for (int i = 0; i < DIM; i += 2) {
                const double2 q2 = *reinterpret_cast<const double2*>(pcol + i);
                T[i] = fma(ue_par, q2.x, ue_own * S[i]);
                T[i + 1] = fma(ue_par, q2.y, ue_own * S[i + 1]);
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            Sn[i] = -(u00 * T[i] + u10 * T[i + D]);                 // row i' (position): U[0][0] T[i'] + U[1][0] T[i' + D]
            Sn[i + D] = -(u01 * T[i] + u11 * T[i + D]);             // row i' + D (velocity)
        }
        // ---- carry to the neighbour's right-hand side: gnext - U^T z
        if (valid) vb[j] = z;
        rows_sync();
        {
            const double zpart = vb[jpart];
            const double zp = jpos ? z : zpart, zv = jpos ? zpart : z;
            rcarry = gnext - (jpos ? u00 * zp + u10 * zv : u01 * zp + u11 * zv);
        }
        // the lane's slot back to all-zero for the next step's constant part
#pragma unroll
        for (int i = 0; i < DIM; i += 2) *reinterpret_cast<double2*>(mycol + i) = make_double2(0.0, 0.0);
        rows_sync();
        return z;
    };

    // ---- forward elimination: both directions in lock step; a row that has run out of plain steps idles
    for (int k = 0; k < nst_max; ++k) {
        if (k < nst) {
            double Sx[DIM];
#pragma unroll
            for (int i = 0; i < DIM; ++i) Sx[i] = (k > 0) ? Sn[i] : 0.0;
            step(t_first + t_inc * k, false, Sx, (k > 0) ? rcarry : 0.0);
        } else {
            // keep the wave-level syncs of step() matched: nothing to do (the syncs are wave barriers, not block barriers)
        }
    }
    // ---- merge row m: direction 1 hands its Schur column and carry to direction 0 (same particle: the row below)
    double zm = 0.0;
    {
        if (split && dir == 1) {
#pragma unroll
            for (int i = 0; i < DIM; i += 2) *reinterpret_cast<double2*>(mycol + i) = make_double2(Sn[i], Sn[i + 1]);
            if (valid) vb[j] = rcarry;
        }
        rows_sync();
        double Sx[DIM];
        double rx = (m > 0) ? rcarry : 0.0;
#pragma unroll
        for (int i = 0; i < DIM; ++i) Sx[i] = (m > 0) ? Sn[i] : 0.0;
        if (split && dir == 0) {
            const double* ocol = colbuf + (lane + 16) * GP_N;               // the same column of the direction-1 row
#pragma unroll
            for (int i = 0; i < DIM; i += 2) {
                const double2 o2 = *reinterpret_cast<const double2*>(ocol + i);
                Sx[i] += o2.x; Sx[i + 1] += o2.y;
            }
            rx += vb_[wave][(row + 1) * GP_N + j];
        }
        rows_sync();
        if (split && dir == 1) {
#pragma unroll
            for (int i = 0; i < DIM; i += 2) *reinterpret_cast<double2*>(mycol + i) = make_double2(0.0, 0.0);
        }
        rows_sync();
        if (dir == 0) zm = step(m, true, Sx, rx);
    }
    // dtheta_m = z_m: update x_m, publish to both directions
    if (dir == 0) {
        if (valid) vb[j] = zm;
        if (pvalid && valid) xb[m * dim + j] = (float)((double)xb[m * dim + j] + K.step * zm);
    }
    rows_sync();
    double dprev_o, dprev_p;                                   // dtheta of the row one step closer to the merge row: own / partner element
    {
        const double* src = vb_[wave] + (row & ~1) * GP_N;     // direction 0's vector of this particle
        dprev_o = src[j];
        dprev_p = src[jpart];
    }
    rows_sync();
    // ---- substitution away from the merge row: dtheta_t = z_t - W_t (U dtheta_prev)
    for (int k = nst_max - 1; k >= 0; --k) {
        const bool act = k < nst;
        const int t = t_first + t_inc * (act ? k : 0);
        const double* wt = wW + (size_t)t * GP_WS_PER_T;
        // v = U dtheta_prev, broadcast; the upper triangle of W_t mirrored into the lane's slot row
        const double v = jpos ? u00 * dprev_o + u01 * dprev_p : u10 * dprev_p + u11 * dprev_o;
        if (act && valid) {
            vb[j] = v;
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
                if (i <= j) {
                    const double w = wt[rows_tri(i, j)];
                    colbuf[((lane & ~15) + j) * GP_N + i] = w;            // W[i][j] into row j of the row-block ...
                    colbuf[((lane & ~15) + i) * GP_N + j] = w;            // ... and its mirror W[j][i] into row i
                }
            }
        }
        rows_sync();
        double d = 0.0;
        if (act && valid) {
            d = wt[GP_TRI + j];
#pragma unroll
            for (int i = 0; i < DIM; i += 2) {
                const double2 w2 = *reinterpret_cast<const double2*>(mycol + i);     // row j of W_t = column j
                const double2 v2 = *reinterpret_cast<const double2*>(vb + i);
                d = fma(-w2.x, v2.x, d);
                d = fma(-w2.y, v2.y, d);
            }
        }
        rows_sync();
        if (act && valid) {
            vb[j] = d;
            if (pvalid) xb[t * dim + j] = (float)((double)xb[t * dim + j] + K.step * d);
        }
        rows_sync();
        if (act) {
            dprev_o = vb[j];
            dprev_p = vb[jpart];
        }
        rows_sync();
    }
    // ---- cost of the particle: sum over the 32 lanes of its two rows
    {
        cost += __shfl_xor(cost, 1, 64);
        cost += __shfl_xor(cost, 2, 64);
        cost += __shfl_xor(cost, 4, 64);
        cost += __shfl_xor(cost, 8, 64);
        cost += __shfl_xor(cost, 16, 64);
        if (costs_out != nullptr && pvalid && (lane & 31) == 0) costs_out[b] = (float)cost;
    }
}

// launcher used by mpb_gpmp2_solve (mpb_gpmp2.hip); returns false when the shape is not served here
bool gpmp2_solve_rows_launch(float* x, const float* start, const float* goal, const float* jac, const double* dm, double* fz,
                             float* costs_out, int B, int H, int D, int n_fields, const void* Kp, hipStream_t st) {
    const GpConst K = *reinterpret_cast<const GpConst*>(Kp);
    const int per_block = 2 * ROWS_WAVES;
    const dim3 grid((B + per_block - 1) / per_block), block(64 * ROWS_WAVES);
#define ROWS_LAUNCH(DT)                                                                                                 \
    do {                                                                                                                \
        if (n_fields == 1)                                                                                              \
            hipLaunchKernelGGL((gpmp2_solve_rows_kernel<DT, false>), grid, block, 0, st, x, start, goal, jac, dm, fz,   \
                               costs_out, B, H, n_fields, K);                                                           \
        else                                                                                                            \
            hipLaunchKernelGGL((gpmp2_solve_rows_kernel<DT, true>), grid, block, 0, st, x, start, goal, jac, dm, fz,    \
                               costs_out, B, H, n_fields, K);                                                           \
    } while (0)
    switch (D) {
        case 2: ROWS_LAUNCH(2); return true;
        case 3: ROWS_LAUNCH(3); return true;
        case 7: ROWS_LAUNCH(7); return true;
        default: return false;
    }
#undef ROWS_LAUNCH
}
