// Shelved experiment (round 3; NOT built): the GPMP2 block-tridiagonal solve with NCH particles per wave, their
// Gauss-Jordan block steps interleaved in one instruction stream (a fragment of csrc/mpb_gpmp2.hip: it uses that file's
// helpers and was launched from mpb_gpmp2_solve with grid (B + NCH - 1) / NCH).  Finite results, not bit-identical to
// gpmp2_solve_kernel's (difference not investigated: the experiment was dropped for its timing), and
// SLOWER at C4 (B = 2048, H = 128, D = 7; scripts/experiments/ab_gpmp2_chains.py): 1 chain per wave, 2 waves per SIMD 0.467 ms /
// iteration; 2 chains, 2 waves 0.494; 4 chains, 1 wave (256 VGPRs + 103 AGPRs) 0.784.  One in-order instruction stream
// with shared s_waitcnt counters hides the latencies of four chains worse than two waves hide those of two.
// ------------------------------------------------------------------------------------------------
// The same solve with NCH independent particles per wave ("chains"), their steps interleaved in ONE instruction stream.
//
// gpmp2_solve_kernel is a dependent chain per wave -- per block step of the Gauss-Jordan inverse: v_readlane, determinant,
// v_rcp_f64 + two Newton steps, the B operand, one MFMA, the pivot rows -- and its ~250 VGPRs (85 of them per-lane constant
// tables of the tile assembly) allow two waves per SIMD: the fp64 pipe is busy about half the time.
// Here a wave carries NCH particles through the same recursion: the constant tables, the loop control, the damping row and
// every scalar are shared, only the tiles, carries and prefetched rows exist per chain, and the compiler interleaves the
// NCH independent chains (they sit in the same basic blocks, unrolled).  NCH = 4 at one wave per SIMD (512 registers) keeps
// four chains in flight per SIMD instead of two, and 4096 chains (C4: 2048 particles x 2 directions) are resident at once.
// Same arithmetic per chain, same bits as gpmp2_solve_kernel (tests/test_gpu_parity_gpmp2_mppi.py).
// Restricted to what C4 needs: compile-time D with 2D <= 14 (the right-hand side rides as column 14), one collision
// field, H >= 4 (two waves per particle, merge row).  Everything else takes gpmp2_solve_kernel.
// ------------------------------------------------------------------------------------------------
template <int DT, int NCH>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NCH >= 4 ? 1 : 2))) void gpmp2_solve_mc_kernel(
    float* __restrict__ x, const float* __restrict__ start, const float* __restrict__ goal, const float* __restrict__ jac,
    const double* __restrict__ diag_mean, double* __restrict__ work, float* __restrict__ costs_out, int B, int H, GpConst K) {
    static_assert(DT > 0 && 2 * DT <= 14, "the right-hand side needs column 14 of the tile");
    constexpr int D = DT, dim = 2 * DT;
    __shared__ double Sb_[2][NCH][GP_N * GP_LD];
    __shared__ double zv_[2][NCH][GP_N];
    __shared__ double dth_[2][NCH][GP_N];
    __shared__ double ex_S[NCH][4][64];
    __shared__ double ex_r[NCH][GP_N];
    __shared__ double ex_d[NCH][GP_N];
    __shared__ double ex_cost[NCH];
    const int lane = threadIdx.x & 63;
    const int dir = threadIdx.x >> 6;
    const int m = (H - 1) >> 1;
    const int nst = dir ? (H - 1 - m) : m;
    const int nrows = nst + 1;
    const double dt = K.dt;
    const double a = 12.0 / (dt * dt * dt) * K.kgp, bq = -6.0 / (dt * dt) * K.kgp, cq = 4.0 / dt * K.kgp;
    const double p00 = a, p01 = 6.0 / (dt * dt) * K.kgp, p11 = cq;
    const double u00 = -a, u11 = -(bq * dt + cq);
    const double u01 = dir ? -(a * dt + bq) : -bq, u10 = dir ? -bq : -(a * dt + bq);
    // chain c of this block: particle NCH * blockIdx.x + c (the last block repeats particle B - 1: identical values
    // written twice by the same wave)
    double* wW[NCH];
    float* xb[NCH];
    const float* jb[NCH];
    int bc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int b = min((int)blockIdx.x * NCH + c, B - 1);
        bc[c] = b;
        wW[c] = work + (size_t)b * H * GP_WS_PER_T;
        xb[c] = x + (size_t)b * H * dim;
        jb[c] = jac + (size_t)b * H * (D + 1);
    }
    const int li = lane & 15, lk = lane >> 4;

    double asm_g1[4], asm_g2[4], asm_dg[4], asm_pp[4], asm_id[4];
    bool asm_in[4];
    int asm_hi[4];
    const int asm_di = (li < dim) ? li : 0;
    const int asm_hj = (li < D) ? li : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q, j = li;
        const bool in = i < dim && j < dim;
        const bool ip = i < D, jp = j < D;
        const int ii = ip ? i : i - D, jj = jp ? j : j - D;
        const bool same = in && ii == jj;
        asm_in[q] = in;
        asm_g1[q] = same ? (ip ? (jp ? p00 : p01) : (jp ? p01 : p11)) : 0.0;
        asm_g2[q] = same ? (ip ? (jp ? a : bq) : (jp ? bq : cq)) : 0.0;
        asm_dg[q] = (in && i == j) ? 1.0 : 0.0;
        asm_pp[q] = (in && ip && jp) ? 1.0 : 0.0;
        asm_hi[q] = (i < D) ? i : 0;
        asm_id[q] = (i == j) ? 1.0 : 0.0;
    }
    double nt_c[4][4];
    int nt_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lk + 4 * q, j = li;
        const bool in = i < dim && j < dim;
        const bool ip = i < D, jp = j < D;
        const int ii = in ? (ip ? i : i - D) : 0, jj = in ? (jp ? j : j - D) : 0;
        const double uca0 = ip ? u00 : u01, uca1 = ip ? u10 : u11;
        const double ueb0 = jp ? u00 : u01, ueb1 = jp ? u10 : u11;
        const double mm = in ? -1.0 : 0.0;
        nt_c[q][0] = mm * uca0 * ueb0;
        nt_c[q][1] = mm * uca0 * ueb1;
        nt_c[q][2] = mm * uca1 * ueb0;
        nt_c[q][3] = mm * uca1 * ueb1;
        nt_off[q] = ii * GP_LD + jj;
    }
    int tri_st[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tri_st[q] = (lk + 4 * q <= li) ? gp_tri(lk + 4 * q, li) : 0;
    const int t_first = dir ? H - 1 : 0, t_inc = dir ? -1 : 1;

    f64x4 Snext[NCH];
    double rcarry[NCH], cost[NCH], z_last[NCH], x_last[NCH];
    float xr0[NCH], xr1[NCH], jr[NCH], jr1[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        Snext[c] = f64x4{0.0, 0.0, 0.0, 0.0};
        rcarry[c] = 0.0; cost[c] = 0.0; z_last[c] = 0.0; x_last[c] = 0.0;
        xr0[c] = (lane < dim) ? xb[c][t_first * dim + lane] : 0.f;
        xr1[c] = (lane < dim && nrows > 1) ? xb[c][(t_first + t_inc) * dim + lane] : 0.f;
        jr[c] = (lane <= D) ? jb[c][t_first * (D + 1) + lane] : 0.f;
        jr1[c] = (lane <= D && nrows > 1) ? jb[c][(t_first + t_inc) * (D + 1) + lane] : 0.f;
    }
    double dm0 = K.trust ? diag_mean[(size_t)t_first * dim + asm_di] : 0.0;
    double dm1 = (K.trust && nrows > 1) ? diag_mean[(size_t)(t_first + t_inc) * dim + asm_di] : 0.0;
    const int ksteps = dir ? nst : nst + 1;
    for (int k = 0; k < ksteps; ++k) {
        const int t = t_first + t_inc * k;
        const bool merge = (dir == 0) && (k == m);
        if (merge) __syncthreads();
        const int t2 = t + 2 * t_inc;
        const bool has2 = k + 2 < nrows;
        float xr2[NCH], jr2[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            xr2[c] = (lane < dim && has2) ? xb[c][t2 * dim + lane] : 0.f;
            jr2[c] = (lane <= D && has2) ? jb[c][t2 * (D + 1) + lane] : 0.f;
        }
        const double dm2 = (K.trust && has2) ? diag_mean[(size_t)t2 * dim + asm_di] : 0.0;
        const double first = (t == 0) ? 1.0 : 0.0, notfirst = 1.0 - first, notlast = (t < H - 1) ? 1.0 : 0.0;
        const double dg = (K.trust ? K.delta * dm0 : K.delta) + first * K.ks + (1.0 - notlast) * K.kg;
        f64x4 T[NCH];
        double x0s[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double x0 = (double)xr0[c], x1 = (double)xr1[c];
            x0s[c] = x0;
            const float hf = (t > 0) ? jr[c] : 0.f;
            // ---- GP factor between this row and the neighbour (gpmp2_solve_kernel)
            double own_i = 0.0, gnext = 0.0;
            if (!merge) {
                const int partner = (lane < D) ? lane + D : lane - D;
                const double x0p = (double)__shfl(xr0[c], partner, 64), x1p = (double)__shfl(xr1[c], partner, 64);
                if (lane < dim) {
                    const bool pos = lane < D;
                    const double lo_o = dir ? x1 : x0, hi_o = dir ? x0 : x1;
                    const double lo_p = dir ? x1p : x0p, hi_p = dir ? x0p : x1p;
                    const double ep = pos ? hi_o - (lo_o + dt * lo_p) : hi_p - (lo_p + dt * lo_o);
                    const double ev = pos ? hi_p - lo_p : hi_o - lo_o;
                    const double qp = a * ep + bq * ev, qv = bq * ep + cq * ev;
                    const double qe_i = pos ? qp : qv;
                    const double pqe_i = pos ? qp : dt * qp + qv;
                    cost[c] += pos ? ep * qp : ev * qv;
                    own_i = dir ? -qe_i : pqe_i;
                    gnext = dir ? pqe_i : -qe_i;
                }
            }
            // ---- tile and right-hand side
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = (k > 0) ? Snext[c][q] : 0.0;
                if (merge) v[q] += ex_S[c][q][lane];
                v[q] = fma(notlast, asm_g1[q], v[q]);
                v[q] = fma(notfirst, asm_g2[q], v[q]);
                v[q] = fma(asm_dg[q], dg, v[q]);
            }
            double r = (k > 0) ? rcarry[c] : 0.0;
            if (merge && lane < dim) r += ex_r[c][lane];
            if (t == 0 && lane < dim) {
                const double es = (double)start[(size_t)bc[c] * dim + lane] - x0;
                r += K.ks * es;
                cost[c] += K.ks * es * es;
            }
            if (t == H - 1 && lane < dim) {
                const double eg = (double)goal[(size_t)bc[c] * dim + lane] - x0;
                r += K.kg * eg;
                cost[c] += K.kg * eg * eg;
            }
            r += own_i;
            {
                const double hcol = (double)__shfl(hf, asm_hj, 64) * (K.kc * notfirst);
                const double cf = (double)__shfl(hf, D, 64);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fma(asm_pp[q] * hcol, (double)__shfl(hf, asm_hi[q], 64), v[q]);
                if (t > 0 && lane < D) r += K.kc * (double)hf * cf;
                if (t > 0 && lane == 0) cost[c] += K.kc * cf * cf;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) T[c][q] = asm_in[q] ? v[q] : asm_id[q];
            if (lane >= dim) r = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = lk + 4 * q;
                const double rq = __shfl(r, row < dim ? row : 0, 64);
                if (li == 14 && row < dim) T[c][q] = rq;
            }
            rcarry[c] = gnext;                 // (parked: the carry to the neighbour is finished behind the inverse)
        }
        // ---- W = S^-1: the block steps of the NCH chains side by side (gpmp2_solve_kernel for the step itself)
#pragma unroll
        for (int kb2 = 0; kb2 < DT; ++kb2) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int k0 = 2 * kb2, q = kb2 >> 1, half = kb2 & 1;
                const double tk = T[c][q];
                const double p00 = readlane_f64(tk, (2 * half) * 16 + k0), p01 = readlane_f64(tk, (2 * half) * 16 + k0 + 1);
                const double p11 = readlane_f64(tk, (2 * half + 1) * 16 + k0 + 1);
                const double id = fast_rcp(fma(p00, p11, -p01 * p01));
                const double i00 = p11 * id, i01 = -p01 * id, i11 = p00 * id;
                const bool jin = (li >= k0) && (li < k0 + 2);
                const double notj = jin ? 0.0 : 1.0;
                const double am0 = fma(notj, __shfl(tk, (2 * half) * 16 + li, 64), (li == k0) ? 1.0 : 0.0);
                const double am1 = fma(notj, __shfl(tk, (2 * half + 1) * 16 + li, 64), (li == k0 + 1) ? 1.0 : 0.0);
                const double sel0 = (lk == 2 * half) ? 1.0 : 0.0, sel1 = (lk == 2 * half + 1) ? 1.0 : 0.0;
                const double bop = fma(sel0 * i00 + sel1 * i01, am0, (sel0 * i01 + sel1 * i11) * am1);
                const double act = sel0 + sel1;
                const double aop = act * ((li < k0) ? tk : -tk);
                f64x4 cin;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) cin[qq] = notj * T[c][qq];
                T[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, cin, 0, 0, 0);
                T[c][q] = fma(act, bop, (1.0 - act) * T[c][q]);
            }
        }
        // ---- W_t to LDS and to the workspace, z = column 14, next tile, carry
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double* Wl = Sb_[dir][c];
#pragma unroll
            for (int q = 0; q < 4; ++q) Wl[(lk + 4 * q) * GP_LD + li] = T[c][q];
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double* W = Sb_[dir][c];
            double* wt = wW[c] + (size_t)t * GP_WS_PER_T;
            const int rowl = (lane < dim) ? lane : 0;
            const double zi = (lane < dim) ? W[rowl * GP_LD + 14] : 0.0;
            if (lane < dim) wt[GP_TRI + lane] = zi;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lk + 4 * q <= li) wt[tri_st[q]] = T[c][q];
            z_last[c] = zi;
            x_last[c] = x0s[c];
            if (!merge) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double* Wq = W + nt_off[q];
                    double v = nt_c[q][0] * Wq[0];
                    v = fma(nt_c[q][1], Wq[D], v);
                    v = fma(nt_c[q][2], Wq[D * GP_LD], v);
                    v = fma(nt_c[q][3], Wq[D * GP_LD + D], v);
                    Snext[c][q] = v;
                }
                const bool ip = lane < D;
                const double zpart = __shfl(zi, ip ? lane + D : lane - D, 64);
                const double zp = ip ? zi : zpart, zvv = ip ? zpart : zi;
                rcarry[c] = rcarry[c] - (ip ? u00 * zp + u10 * zvv : u01 * zp + u11 * zvv);
            }
        }
        if (!merge) wave_sync();
#pragma unroll
        for (int c = 0; c < NCH; ++c) { xr0[c] = xr1[c]; xr1[c] = xr2[c]; jr[c] = jr1[c]; jr1[c] = jr2[c]; }
        dm0 = dm1; dm1 = dm2;
    }
    // ---- hand-over at the merge row
    if (dir) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ex_S[c][q][lane] = Snext[c][q];
            if (lane < dim) ex_r[c][lane] = rcarry[c];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (lane < dim) {
                ex_d[c][lane] = z_last[c];
                xb[c][m * dim + lane] = (float)(x_last[c] + K.step * z_last[c]);
            }
    }
    __syncthreads();
    // ---- substitution away from the merge row (gpmp2_solve_kernel), the chains side by side
    const bool rowlane = lane < dim;
    const int rl = rowlane ? lane : 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (rowlane) {
            const bool ip = lane < D;
            const int ii = ip ? lane : lane - D;
            const double dp = ex_d[c][ii], dv = ex_d[c][ii + D];
            zv_[dir][c][lane] = ip ? u00 * dp + u01 * dv : u10 * dp + u11 * dv;
        }
    wave_sync();
    if (nst > 0) {
        constexpr int NW = 2 * DT;
        int tri_ld[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) tri_ld[j] = gp_tri(min(rl, j), max(rl, j));
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 sa[NCH][GP_PF], sb[NCH][GP_PF];
        float xst[NCH][GP_PF];
        constexpr int REC2 = GP_WS_PER_T / 2;
        auto fetch = [&](int c, int k, d2& a0, d2& b0, float& xdst) {
            const int t = t_first + t_inc * k;
            const d2* rec = reinterpret_cast<const d2*>(wW[c] + (size_t)t * GP_WS_PER_T);
            a0 = rec[lane];
            b0 = rec[64 + (lane < REC2 - 64 ? lane : 0)];
            xdst = xb[c][t * dim + rl];
        };
#pragma unroll
        for (int u = 0; u < GP_PF; ++u)
            if (nst - 1 - u >= 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) fetch(c, nst - 1 - u, sa[c][u], sb[c][u], xst[c][u]);
            }
        for (int kk = nst - 1; kk >= 0; kk -= GP_PF) {
#pragma unroll
            for (int u = 0; u < GP_PF; ++u) {
                const int k = kk - u;
                if (k < 0) break;
                const int t = t_first + t_inc * k;
                float xc[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    double* slot = Sb_[dir][c];
                    reinterpret_cast<d2*>(slot)[lane] = sa[c][u];
                    if (lane < REC2 - 64) reinterpret_cast<d2*>(slot)[64 + lane] = sb[c][u];
                    xc[c] = xst[c][u];
                    if (k - GP_PF >= 0) fetch(c, k - GP_PF, sa[c][u], sb[c][u], xst[c][u]);
                }
                wave_sync();
                double d[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const double* slot = Sb_[dir][c];
                    d[c] = slot[GP_TRI + rl];
                    if (rowlane) {
#pragma unroll
                        for (int j = 0; j < NW; ++j) d[c] -= slot[tri_ld[j]] * zv_[dir][c][j];
                    }
                }
                wave_sync();
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (rowlane) {
                        dth_[dir][c][lane] = d[c];
                        xb[c][t * dim + lane] = (float)((double)xc[c] + K.step * d[c]);
                    }
                wave_sync();
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (rowlane) {
                        const bool ip = lane < D;
                        const int ii = ip ? lane : lane - D;
                        const double dp = dth_[dir][c][ii], dv = dth_[dir][c][ii + D];
                        zv_[dir][c][lane] = ip ? u00 * dp + u01 * dv : u10 * dp + u11 * dv;
                    }
                wave_sync();
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        cost[c] = wave_sum_f64(cost[c]);
        if (dir && lane == 0) ex_cost[c] = cost[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (costs_out != nullptr && dir == 0 && lane == 0) costs_out[bc[c]] = (float)(cost[c] + ex_cost[c]);
}

