// mpb_prior.hip -- initial particles from the constant-velocity GP prior.
//
// Replaces OptimizationPlanner.get_random_trajs (base.py:155-202) over MultiMPPrior
// (costs/factors/mp_priors_multi.py:100-110, :213-256).  The reference builds the dense M x M precision
// K^-1 = A^T Q^-1 A (M = 2D*H) in fp64, lets MultivariateNormal turn it into a dense scale_tril
// (flip-Cholesky + triangular solve, torch/distributions/multivariate_normal.py:80-86) and multiplies it
// with eps.  K^-1 is block tridiagonal with blocks (2x2) (x) I_D, so that scale_tril is U^-T with
// K^-1 = U U^T and U block upper-bidiagonal with (2x2 upper-triangular) (x) I_D blocks: every
// (sample, degree of freedom) pair is an independent 2-state chain and  x = mean + U^-T eps  is a forward
// substitution along the horizon.  The host computes the H small 2x2 factors in fp64
// (planners/base.py: gp_prior_factor); this kernel does the substitution, one thread per chain, in fp64.
#include "mpb_common.h"

// 1/x in fp64: v_rcp_f64 + two Newton steps (the IEEE division hipcc emits is ~40 instructions, twice per step)
__device__ __forceinline__ double prior_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

__global__ __launch_bounds__(256) void gp_prior_sample_kernel(float* __restrict__ out, const double* __restrict__ means,
                                                              const double* __restrict__ eps,
                                                              const double* __restrict__ Udiag,
                                                              const double* __restrict__ Uoff, int G, int n, int H,
                                                              int D, uint32_t seed_lo, uint32_t seed_hi) {
    const int chain = blockIdx.x * blockDim.x + threadIdx.x;   // (i sample, g mode, d dof)
    if (chain >= n * G * D) return;
    const int d = chain % D, g = (chain / D) % G, i = chain / (D * G);
    const int dim = 2 * D;
    const double* mu = means + (size_t)g * H * dim;
    const double* ep = eps ? eps + ((size_t)i * G + g) * H * dim : nullptr;
    float* o = out + ((size_t)g * n + i) * H * dim;           // particle index = mode * n + sample (base.py:202)
    double yp = 0.0, yv = 0.0;
    float nrm[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < H; ++t) {
        double ep_p, ep_v;
        if (ep) {
            ep_p = ep[t * dim + d];
            ep_v = ep[t * dim + D + d];
        } else {
            // one Philox call per PAIR of time steps: (pos, vel) of t = 2 tp from (x, y), of t = 2 tp + 1 from (z, w)
            if ((t & 1) == 0) {
                const uint4 r = philox4x32_10(make_uint4((uint32_t)chain, (uint32_t)(t >> 1), 0x6770u, 0u), make_uint2(seed_lo, seed_hi));
                box_muller(r.x, r.y, nrm[0], nrm[1]);
                box_muller(r.z, r.w, nrm[2], nrm[3]);
            }
            ep_p = nrm[2 * (t & 1)]; ep_v = nrm[2 * (t & 1) + 1];
        }
        double rp = ep_p, rv = ep_v;
        if (t > 0) {   // rhs -= U_{t-1,t}^T y_{t-1}
            const double* O = Uoff + (size_t)(t - 1) * 4;     // row-major 2x2
            rp -= O[0] * yp + O[2] * yv;
            rv -= O[1] * yp + O[3] * yv;
        }
        const double* U = Udiag + (size_t)t * 3;              // u00, u01, u11  (U_tt = [[u00,u01],[0,u11]])
        yp = rp * prior_rcp(U[0]);
        yv = (rv - U[1] * yp) * prior_rcp(U[2]);
        o[t * dim + d] = (float)(mu[t * dim + d] + yp);
        o[t * dim + D + d] = (float)(mu[t * dim + D + d] + yv);
    }
}

// ------------------------------------------------------------------------------------------------
// Dense form on the matrix cores (H <= 128).  For one degree of freedom the map eps -> y is the 2H x 2H lower-
// triangular scale_tril T = U_dof^-T of the reference's MultivariateNormal (the same for every chain), so
// Y = T E with E the (2H x chains) block of standard normals is a GEMM: v_mfma_f64_16x16x4_f64, 16 chains per
// wave.  A workgroup owns SB whole particles (SB * D <= 64 chains) so that its output is one contiguous slab:
// T is streamed through LDS one 16-row tile at a time (shared by the four waves, odd row stride: conflict-free
// A-operand reads), the B operand -- this lane's normals -- stays in registers, and every row tile of the result
// is transposed through LDS and written with coalesced stores.  The sequential kernel above needs H dependent
// steps per chain; here the depth is the MFMA chain of one row tile.
// ------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KT>   // K tiles of 4: 2H <= 4 KT  (KT = 32: H <= 64, KT = 64: H <= 128)
__global__ __launch_bounds__(256) void gp_prior_dense_kernel(float* __restrict__ out, const double* __restrict__ means,
                                                             const float* __restrict__ means32,   // fp32 means instead (means == NULL)
                                                             const double* __restrict__ eps, const double* __restrict__ T,
                                                             int G, int n, int H, int D, int SB, uint32_t seed_lo,
                                                             uint32_t seed_hi) {
    constexpr int LD = 4 * KT + 1;                     // doubles per staged row (odd: distinct banks for 16 rows)
    // T row tile; before the first tile the same memory carries the normals exchange (4 waves x 4 KT x 16 floats)
    __shared__ double Tt[(16 * LD > 128 * KT) ? 16 * LD : 128 * KT];
    extern __shared__ float Yt[];                      // SB x 8 x 2D: one row tile of the block's output, particle-major
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int c = wave * 16 + li;                      // chain slot of this lane (column of the MFMA tile)
    const int sb = c / D, d = c - sb * D;
    const int pidx = blockIdx.x * SB + sb;             // particle index = mode * n + sample (base.py:202)
    const bool valid = sb < SB && pidx < G * n;
    const int pp = valid ? pidx : 0;
    const int g = pp / n, i = pp - g * n;
    const int chain = (i * G + g) * D + d;             // Philox counter: same stream as the sequential kernel
    const int dim = 2 * D, N2 = 2 * H;
    // ---- B operand: eps[k = 4 kt + lk] of this chain, k = 2 t + s (s = 0 position, 1 velocity)
    double e[KT];
    if (eps != nullptr) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = 4 * kt + lk, t = k >> 1, sv = k & 1;
            e[kt] = (valid && t < H) ? eps[(((size_t)i * G + g) * H + t) * dim + (sv ? D + d : d)] : 0.0;
        }
    } else {
        // one Philox call yields the four normals k = 4 tp .. 4 tp + 3 of a chain.  The four lanes that share a
        // chain (lk = 0..3) split the calls (tp = 4 j + lk) and swap the results through LDS, so every call is
        // made exactly once (Philox is 40 quarter-rate integer multiplies)
        float* Ex = reinterpret_cast<float*>(Tt) + (size_t)wave * (4 * KT) * 16;     // [k][chain slot li]
#pragma unroll
        for (int j = 0; j < KT / 4; ++j) {
            const int tp = 4 * j + lk;
            float nrm[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid && 2 * tp < H) {
                const uint4 r = philox4x32_10(make_uint4((uint32_t)chain, (uint32_t)tp, 0x6770u, 0u), make_uint2(seed_lo, seed_hi));
                box_muller(r.x, r.y, nrm[0], nrm[1]);
                box_muller(r.z, r.w, nrm[2], nrm[3]);
                if (2 * tp + 1 >= H) nrm[2] = nrm[3] = 0.f;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) Ex[(4 * tp + x) * 16 + li] = nrm[x];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) e[kt] = (double)Ex[(4 * kt + lk) * 16 + li];
    }
    const int nsb = min(SB, G * n - blockIdx.x * SB);  // particles of this block
    // write-out slots of this thread: flat index tid + 256 j over (particle of the block, 8 x 2D tile elements)
    constexpr int WSLOTS = 4;                          // SB * 8 * 2D <= 64 * 16 = 1024 elements per tile
    int w_s2[WSLOTS], w_rem[WSLOTS];
    size_t w_out[WSLOTS], w_mu[WSLOTS];
#pragma unroll
    for (int j = 0; j < WSLOTS; ++j) {
        const int idx = threadIdx.x + 256 * j;
        const int s2 = idx / (8 * dim);
        w_s2[j] = s2;
        w_rem[j] = idx - s2 * 8 * dim;
        const int p2 = min(blockIdx.x * SB + s2, G * n - 1);
        w_out[j] = (size_t)p2 * H * dim;
        w_mu[j] = (size_t)(p2 / n) * H * dim;
    }
    // row tile r of T is 16 x 16 (r + 1) doubles (lower triangular: the columns beyond the tile are zero) = r + 1 per
    // thread.  The NEXT tile is fetched into registers while the matrix instructions of the current one run: a tile's
    // trip from L2 (every workgroup streams the whole factor) would otherwise sit between two barriers, sixteen times
    double pf[KT / 4];
    auto fetch = [&](int r, int cnt) {
        const int ncol = 16 * (r + 1);
#pragma unroll
        for (int j = 0; j < KT / 4; ++j) {
            if (j < cnt) {
                const int idx = threadIdx.x + 256 * j;
                const int row = idx / ncol, col = idx - row * ncol;
                const int gr = 16 * r + row;
                pf[j] = (gr < N2 && col < N2) ? T[(size_t)gr * N2 + col] : 0.0;
            }
        }
    };
    fetch(0, 1);
#pragma unroll
    for (int r = 0; r < KT / 4; ++r) {                 // 16-row tiles of T = 8 time steps x (pos, vel)
        if (16 * r < N2) {
            const int ncol = 16 * (r + 1);
            __syncthreads();
#pragma unroll
            for (int j = 0; j <= r; ++j) {
                const int idx = threadIdx.x + 256 * j;
                const int row = idx / ncol, col = idx - row * ncol;
                Tt[row * LD + col] = pf[j];
            }
            if (r + 1 < KT / 4 && 16 * (r + 1) < N2) fetch(r + 1, r + 2);
            __syncthreads();
            // two independent accumulation chains (even / odd K tiles) keep the MFMA pipe busy
            f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kt = 0; kt < 4 * (r + 1); kt += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[li * LD + 4 * kt + lk], e[kt], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Tt[li * LD + 4 * (kt + 1) + lk], e[kt + 1], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += acc1[q];
            // D layout: rows lk + 4 q of the tile, column (chain) li  ->  Yt[sb][t_local][channel]
            if (sb < SB) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = lk + 4 * q, tl = row >> 1, sv = row & 1;
                    Yt[(sb * 8 + tl) * dim + (sv ? D + d : d)] = (float)acc[q];
                }
            }
            __syncthreads();
            // coalesced write-out: per particle the 8 time steps x 2D channels of this tile are contiguous; the
            // (particle, element) split of this thread's slots was fixed before the loop (no divisions here)
            const int t0 = 8 * r, nt = min(8, H - t0);
#pragma unroll
            for (int j = 0; j < WSLOTS; ++j) {
                if (w_s2[j] < nsb && w_rem[j] < nt * dim) {
                    const size_t off = (size_t)t0 * dim + w_rem[j];
                    const double mu = means32 ? (double)means32[w_mu[j] + off] : means[w_mu[j] + off];
                    out[w_out[j] + off] = (float)(mu + (double)Yt[w_s2[j] * 8 * dim + w_rem[j]]);
                }
            }
        }
    }
}

// dense sampler, means in fp64 (C-ABI) or fp32 (means32; mpb_stoch_gpmp_step starts from the fp32 particle means and
// needs no widened copy)
int mpb_gp_prior_dense_launch(float* out, const double* means, const float* means32, const double* eps,
                              const double* scale_tril, int G, int n, int H, int D, uint64_t seed, void* stream) {
    if (!out || (!means && !means32) || !scale_tril) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample_dense: null pointer");
    if (G < 1 || n < 0 || H < 2 || D < 1 || D > 64) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample_dense: bad shape");
    if (H > 128) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_gp_prior_sample_dense: H > 128 (use mpb_gp_prior_sample)");
    if (n == 0) return MPB_OK;
    const int SB = 64 / D;                                      // whole particles per workgroup (SB * D <= 64 chains)
    const dim3 grid((G * n + SB - 1) / SB), block(256);
    const size_t lds = (size_t)SB * 8 * 2 * D * sizeof(float);
    if (H <= 64)
        hipLaunchKernelGGL(gp_prior_dense_kernel<32>, grid, block, lds, (hipStream_t)stream, out, means, means32, eps,
                           scale_tril, G, n, H, D, SB, (uint32_t)seed, (uint32_t)(seed >> 32));
    else
        hipLaunchKernelGGL(gp_prior_dense_kernel<64>, grid, block, lds, (hipStream_t)stream, out, means, means32, eps,
                           scale_tril, G, n, H, D, SB, (uint32_t)seed, (uint32_t)(seed >> 32));
    return mpb_check_launch("mpb_gp_prior_sample_dense");
}

extern "C" int mpb_gp_prior_sample_dense(float* out, const double* means, const double* eps, const double* scale_tril,
                                         int G, int n, int H, int D, uint64_t seed, void* stream) {
    if (!means) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample_dense: null pointer");
    return mpb_gp_prior_dense_launch(out, means, nullptr, eps, scale_tril, G, n, H, D, seed, stream);
}

extern "C" int mpb_gp_prior_sample(float* out, const double* means, const double* eps, const double* Udiag,
                                   const double* Uoff, int G, int n, int H, int D, uint64_t seed, void* stream) {
    if (!out || !means || !Udiag || !Uoff) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample: null pointer");
    if (G < 1 || n < 0 || H < 2 || H > 4096 || D < 1 || D > 64) return mpb_fail(MPB_E_INVALID, "mpb_gp_prior_sample: bad shape");
    if (n == 0) return MPB_OK;
    const int chains = n * G * D;
    hipLaunchKernelGGL(gp_prior_sample_kernel, dim3((chains + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, means,
                       eps, Udiag, Uoff, G, n, H, D, (uint32_t)seed, (uint32_t)(seed >> 32));
    return mpb_check_launch("mpb_gp_prior_sample");
}

// ------------------------------------------------------------------------------------------------
// General multivariate normal from a DENSE scale_tril: x = mean + L eps (fp64), for MultiMPPrior with arbitrary
// (non-isotropic) start / GP / goal precisions (mp_priors_multi.py:213-256 takes any matrices): the precision has no
// (2x2) (x) I structure then and the host hands over the full M x M factor.  Set-up time only (M <= 4096): thread = row,
// the factor is read transposed (coalesced), eps is staged through LDS in chunks of 256.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mvn_dense_sample_kernel(float* __restrict__ out, const double* __restrict__ means,
                                                               const double* __restrict__ eps, const double* __restrict__ tril_t,
                                                               int G, int n, int M, uint32_t seed_lo, uint32_t seed_hi) {
    __shared__ double e_l[256];
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int sg = blockIdx.y;                     // sample * G + mode
    const int smp = sg / G, mode = sg - smp * G;
    double acc = 0.0;
    const int k_end = min(M, (int)(blockIdx.x + 1) * 256);    // rows of this block only reach columns < k_end (lower triangular)
    for (int k0 = 0; k0 < k_end; k0 += 256) {
        __syncthreads();
        const int k = k0 + threadIdx.x;
        double ev = 0.0;
        if (k < M) {
            if (eps != nullptr) {
                ev = eps[((size_t)smp * G + mode) * M + k];
            } else {       // device noise: one Philox4x32-10 call per four consecutive k
                const uint4 rr = philox4x32_10(make_uint4((uint32_t)smp, (uint32_t)mode, (uint32_t)(k >> 2), 0x4d564eu), make_uint2(seed_lo, seed_hi));
                float n0, n1, n2, n3;
                box_muller(rr.x, rr.y, n0, n1);
                box_muller(rr.z, rr.w, n2, n3);
                const int r = k & 3;
                ev = (double)(r == 0 ? n0 : r == 1 ? n1 : r == 2 ? n2 : n3);
            }
        }
        e_l[threadIdx.x] = ev;
        __syncthreads();
        if (m < M) {
            const int kn = min(256, m + 1 - k0);     // columns k0 .. min(k0 + 255, m)
            for (int t = 0; t < kn; ++t) acc = fma(tril_t[(size_t)(k0 + t) * M + m], e_l[t], acc);
        }
    }
    if (m < M) out[((size_t)mode * n + smp) * M + m] = (float)(means[(size_t)mode * M + m] + acc);
}

extern "C" int mpb_mvn_sample_dense(float* out, const double* means, const double* eps, const double* tril_t, int G, int n, int M,
                                    uint64_t seed, void* stream) {
    if (!out || !means || !tril_t) return mpb_fail(MPB_E_INVALID, "mpb_mvn_sample_dense: null pointer");
    if (G < 1 || n < 0 || M < 1 || M > 4096 || (long)n * G > 65535) return mpb_fail(MPB_E_INVALID, "mpb_mvn_sample_dense: bad shape");
    if (n == 0) return MPB_OK;
    hipLaunchKernelGGL(mvn_dense_sample_kernel, dim3((M + 255) / 256, n * G), dim3(256), 0, (hipStream_t)stream, out, means, eps, tril_t,
                       G, n, M, (uint32_t)seed, (uint32_t)(seed >> 32));
    return mpb_check_launch("mpb_mvn_sample_dense");
}
