// stomp_fused_halves.hip -- SHELVED EXPERIMENT (round 2; not built): the persistent STOMP loop of mpb_stomp_fused.hip with TWO
// independent half-blocks per workgroup.  Correct (passed tests/test_gpu_stomp_fused.py when wired into mpb_stomp_run) and
// SLOWER: 18.1-18.3 us / iteration at C3 against 16.3-16.4 for the 16-wave kernel on the same box, with or without an initial
// phase offset between the halves and with or without raised priority for the update phases.
// To try it again: add the file to build.py's SOURCES and call mpb_fused2_launch from mpb_stomp_run (single field, S <= 32).
//
// The 16-wave workgroup of mpb_stomp_fused.hip moves through its phases in lock-step (five block barriers per
// iteration): while the rollouts' cost phase saturates the VALU, the sample phase (LDS-bound) and the update phases
// (latency-bound: softmax partial, exchange with the partner workgroups, Sigma matvec) leave it idle -- 45 % of an
// iteration at C3.  A second workgroup per CU would fill those gaps, but 16 waves x 2 leave 64 VGPRs per wave and the
// LDS images of L, Sigma and the grid (50 KB) would have to be held twice.  Here ONE workgroup of 16 waves holds the
// constants once and runs two UNITS = (particle, chunk of 8 samples) of two DIFFERENT particles, one per half-block
// of 8 waves (two per SIMD).  The halves never meet at a hardware barrier after the prologue: each synchronises with
// its own LDS arrival counter (ds_add + poll, bounded like every wait of this kernel), so the halves drift out of phase
// (the SIMD arbiter serves the older waves first) and one half's VALU phase overlaps the other's LDS / latency phases.
// A particle's S <= 32 samples are nc = ceil(S / 8) chunks on nc workgroups (block indices 8 apart: same XCD);
// per wave the code is that of mpb_stomp_fused.hip, per thread the update phases own two trajectory elements.
#include <hip/hip_runtime.h>

#include "mpb_common.h"
#include "mpb_geom.h"
#include "mpb_stomp_noise.h"

#define F2_HW 8                              // waves per half-block
#define F2_HT (64 * F2_HW)                   // threads per half-block
#define F2_THREADS (2 * F2_HT)
#define F2_LD 68
#define F2_XCHG 912                          // = FUSED_XCHG (same workspace layout)
#define F2_MAX_CHUNKS 4                      // S <= 32
#define F2_TIMEOUT_TICKS 200000000ull        // 2 s of s_memrealtime (100 MHz)

typedef unsigned long long granule2_t;
static __device__ __forceinline__ void st_agent2(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void st_granule2(granule2_t* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((granule2_t)tag << 32) | (granule2_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ granule2_t ld_granule2(const granule2_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// barrier of the 8 waves of one half-block: arrival counter in LDS (monotonic; generation k is complete at 8 k).
// Bounded: a wave that waits longer than the time-out raises the half's abort word and leaves; the others follow.
static __device__ __forceinline__ void half_barrier(unsigned* ctr, unsigned& gen, int* abort_word) {
    gen += F2_HW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - gen) < 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - gen) >= 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > F2_TIMEOUT_TICKS) { *abort_word = 1; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int DCH, int MODEL>
__global__ __launch_bounds__(F2_THREADS, 4) void stomp_fused2_kernel(
    float* __restrict__ means, const float* __restrict__ eps, float* __restrict__ samples, float* __restrict__ costs,
    float* __restrict__ weights, const float* __restrict__ Lmat, const float* __restrict__ Sigma,
    const float* __restrict__ geom, float* __restrict__ ws, int P, int S, int nc, float k_sigma, float weight, float lr,
    float temperature, int n_iters, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter0, uint32_t particle_offset,
    uint32_t tag0) {
    constexpr int H = 64;
    constexpr int N = H * DCH;                    // elements of a trajectory
    constexpr int EPT = (N + F2_HT - 1) / F2_HT;  // trajectory elements per thread of a half-block (1 or 2)
    static_assert(N + 2 <= F2_XCHG && EPT <= 2, "two trajectory elements per thread at most");
    __shared__ __attribute__((aligned(16))) float Lp[H * H];                                  // 16 KB  (both halves)
    __shared__ __attribute__((aligned(16))) float tiles[2 * F2_HW * H * NT_STRIDE];           // 80 KB  (one tile per wave)
    __shared__ __attribute__((aligned(16))) unsigned gridw[MPB_GRID_MAX_CELLS];               // 16 KB  (both halves)
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];                                             //  1 KB
    __shared__ __attribute__((aligned(16))) float sig_l[H * F2_LD];                           // 17 KB  (both halves)
    __shared__ __attribute__((aligned(16))) float mean_l2[2][N];                              // 2 x 3.5 KB
    __shared__ __attribute__((aligned(16))) float delta2[2][DCH * F2_LD];                     // 2 x 3.7 KB (transposed)
    __shared__ float cst2[2][F2_HW];
    __shared__ int s_abort2[2];
    __shared__ unsigned bar2[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 3, hw = wave & (F2_HW - 1);
    // units of this block: the same chunk of the two particles of a pair.  XCD-aware (speed only): blocks b and b + 8 share
    // an L2 -- the nc chunks of a pair are given block indices 8 apart
    const int n_pairs = (P + 1) >> 1;
    int pair, chunk;
    if ((n_pairs & 7) == 0) {
        const int grp = blockIdx.x / (8 * nc), x = blockIdx.x & 7;
        chunk = (blockIdx.x >> 3) % nc;
        pair = 8 * grp + x;
    } else {
        pair = blockIdx.x / nc;
        chunk = blockIdx.x - pair * nc;
    }
    const int p = 2 * pair + half;
    const bool half_on = p < P;                         // (odd P: the last pair has one particle)
    const int pc = half_on ? p : P - 1;
    const int s = chunk * F2_HW + hw;                   // this wave's sample
    const bool live = s < S;
    const int j = lane & 15, g = lane >> 4;
    float* err_word = ws;
    granule2_t* xch = reinterpret_cast<granule2_t*>(ws + 16);
    float* mean_l = mean_l2[half];
    float* delta = delta2[half];
    float* cst = cst2[half];
    int* s_abort = &s_abort2[half];
    unsigned* bar = &bar2[half];
    const float* tiles_h = tiles + half * (F2_HW * H * NT_STRIDE);

    // ---- constants into LDS (once, all 16 waves)
    GeomView G0 = geom_view(geom);
    {
        const int g_rounds = (G0.n_cells + 4 * F2_THREADS - 1) / (4 * F2_THREADS);
        const uint4* g4 = reinterpret_cast<const uint4*>(G0.grid);
        const int n_pad16 = (G0.n_cells + MPB_GRID_PAD - 1) / MPB_GRID_PAD * (MPB_GRID_PAD / 4);   // uint4s in the padded section
        for (int u = 0; u < (MPB_GRID_MAX_CELLS / 4 + F2_THREADS - 1) / F2_THREADS; ++u) {
            const int i = tid + F2_THREADS * u;
            if (u < g_rounds && i < n_pad16) reinterpret_cast<uint4*>(gridw)[i] = g4[i];
        }
        for (int i = tid; i <= G0.n_sph && i <= MPB_GRID_MAX_SPH; i += F2_THREADS)
            otab[i] = (i < G0.n_sph) ? reinterpret_cast<const float4*>(G0.sph)[i] : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
        const f32x4 lv = reinterpret_cast<const f32x4*>(Lmat)[tid];
        const f32x4 sv = reinterpret_cast<const f32x4*>(Sigma)[tid];
        const int row = tid >> 4, col0 = (tid & 15) << 2;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) Lp[stomp_l_image_index(row, col0 + e4)] = lv[e4];
        *reinterpret_cast<f32x4*>(sig_l + row * F2_LD + col0) = sv;
        const int th0 = tid & (F2_HT - 1);
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int e = th0 + F2_HT * r;
            if (e < N) mean_l[e] = means[(size_t)pc * N + e];
        }
        if (th0 == 0) { *s_abort = 0; *bar = 0u; }
        if (blockIdx.x == 0 && tid == 0) st_agent2(ws + 1, __uint_as_float(tag0));     // header word 1 = this call's tag
    }
    __syncthreads();
    if (!half_on) return;              // no hardware barrier below this line

    const size_t eps_stride = (size_t)S * DCH * P * H;
    unsigned bgen = 0u;
#ifdef F2_START_DELAY      // (tuning) the second half starts F2_START_DELAY x 10 ns late: out of phase from the first iteration
    if (half == 1) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(F2_START_DELAY)) __builtin_amdgcn_s_sleep(8);
    }
#endif

    // ---- noise of iteration 0: straight into the wave's tile ([waypoint][channel], stride NT_STRIDE)
    float* nt = tiles + wave * (H * NT_STRIDE);
    {
        float e[16];
        f32x4 acc[4];
        stomp_b_operand<DCH>(e, eps ? eps + (size_t)(live ? s : 0) * DCH * P * H : nullptr, P, p, j, g,
                             particle_offset + (uint32_t)p, (uint32_t)s, iter0, seed_lo, seed_hi);
        stomp_noise_product(Lp, e, j, g, acc);
        stomp_noise_to_tile(nt, acc, lane);
    }

    for (int it = 0; it < n_iters; ++it) {
        // ============ A. samples of this iteration: x = mean + noise, stored, kept packed in the wave's tile
        float nz[16];
        stomp_noise_row<DCH>(nt, lane, nz);
        const int h = lane;
        const bool edge = (h == 0) || (h == H - 1);
        float x[DCH];
        if (DCH % 2 == 0) {
#pragma unroll
            for (int c = 0; c < DCH; c += 2) {
                const float2 mv = *reinterpret_cast<const float2*>(mean_l + h * DCH + c);
                x[c] = mv.x + (edge ? 0.f : nz[c]);
                x[c + 1] = mv.y + (edge ? 0.f : nz[c + 1]);
            }
#pragma unroll
            for (int c = 0; c < DCH; c += 2) *reinterpret_cast<float2*>(nt + h * DCH + c) = make_float2(x[c], x[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < DCH; ++c) x[c] = mean_l[h * DCH + c] + (edge ? 0.f : nz[c]);
#pragma unroll
            for (int c = 0; c < DCH; ++c) nt[h * DCH + c] = x[c];
        }
        __builtin_amdgcn_wave_barrier();
        if (live) {
            const f32x4* pk4 = reinterpret_cast<const f32x4*>(nt);
            f32x4* out4 = reinterpret_cast<f32x4*>(samples + ((size_t)p * S + s) * N);     // uniform
#pragma unroll
            for (int k = 0; k < (16 * DCH + 63) / 64; ++k) {
                const unsigned idx = (unsigned)lane + 64u * k;
                if (idx < 16u * DCH) out4[idx] = pk4[idx];
            }
        }
        // ============ B. collision cost of the rollout (one field: the launcher sends chained fields to the 16-wave kernel)
        {
            float q[MPB_MAX_DOF];
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < DCH) ? x[i < DCH ? i : 0] : 0.f;
            float c = 0.f;
            bool bad = false;
            if (live && h >= 1) {
                if (MODEL == PandaModel::ID) {
                    if (G0.model == PandaModel::ID) c = fmaf(G0.fscale, waypoint_cost_grid_model<PandaModel>(G0, gridw, otab, q), c);
                    else bad = true;
                } else {
                    c = fmaf(G0.fscale, waypoint_cost_grid(G0, gridw, otab, q), c);
                }
            }
            const double csum = wave_sum_f64((double)c);
            const float cw = weight * (k_sigma * (float)csum);
            if (lane == 0) {
                cst[hw] = cw;
                if (live) {
                    if (bad) reinterpret_cast<unsigned*>(costs)[(size_t)p * S + s] = 0x7FC00000u;
                    else costs[(size_t)p * S + s] = cw;
                }
            }
        }
#ifndef F2_NO_UPDATE_PRIO
        __builtin_amdgcn_s_setprio(3);      // the short, latency-bound update phases go ahead of the other half's rollouts
#endif
        half_barrier(bar, bgen, s_abort);                                                       // (1) costs of the chunk
        // ============ C. partial of this chunk: logits, local max, e_w, z, weighted (sample - mean)
        int th = tid & (F2_HT - 1);
        asm volatile("" : "+v"(th));
        const int lq = th & 63;
        const int sl = chunk * F2_HW + (lq & (F2_HW - 1));                  // every wave redundantly, lanes 0-7 carry the chunk
        const float xs = (lq < F2_HW && sl < S) ? -cst[lq & (F2_HW - 1)] / temperature : -3.0e38f;
        const float mb = wave_max_f32(xs);
        const float ex = (lq < F2_HW && sl < S) ? expf(xs - mb) : 0.f;
        const float zb = wave_sum_f32(ex);
        float dpart[EPT];
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int e = th + F2_HT * r;
            dpart[r] = 0.f;
            if (e < N) {
                const float mu = mean_l[e];
#pragma unroll
                for (int w = 0; w < F2_HW; ++w) {
                    const float ew = readlane_f32(ex, w);
                    dpart[r] = fmaf(ew, tiles_h[w * (H * NT_STRIDE) + e] - mu, dpart[r]);
                }
            }
        }
        float m_all = mb, z_all = zb, f_own = 1.f;
        const unsigned tag = tag0 + (unsigned)it;                  // unique per (call, iteration): stale granules never match
        if (nc > 1) {
            // ============ D. publish the partial as tagged granules
            granule2_t* mine = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc + chunk) * F2_XCHG;
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int e = th + F2_HT * r;
                if (e < N) st_granule2(mine + 2 + e, dpart[r], tag);
            }
            if (th == 0) { st_granule2(mine + 0, mb, tag); st_granule2(mine + 1, zb, tag); }
        }
        half_barrier(bar, bgen, s_abort);                                    // (2) the samples in the tiles are consumed
        if (it + 1 < n_iters) {
            // the noise of the NEXT iteration, between publishing and polling: the partners' latency
            float e[16];
            f32x4 acc[4];
            int jv = j, gv = g;
            asm volatile("" : "+v"(jv), "+v"(gv));
            stomp_b_operand<DCH, true>(e, eps ? eps + (size_t)(it + 1) * eps_stride + (size_t)(live ? s : 0) * DCH * P * H : nullptr, P, p, jv, gv,
                                       particle_offset + (uint32_t)p, (uint32_t)s, iter0 + (uint32_t)(it + 1), seed_lo, seed_hi);
            stomp_noise_product(Lp, e, j, g, acc);
            stomp_noise_to_tile(nt, acc, lane);
        }
#ifndef F2_NO_UPDATE_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        float dsum[EPT];
#pragma unroll
        for (int r = 0; r < EPT; ++r) dsum[r] = dpart[r];
        if (nc > 1) {
            // every thread waits for ITS granules of every chunk (its own included: the very bits the partners read);
            // combined in chunk order so that all partners compute bit-identical means
            float mk[F2_MAX_CHUNKS], zk[F2_MAX_CHUNKS], dk[F2_MAX_CHUNKS][EPT];
            const granule2_t* slot0 = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc) * F2_XCHG;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < F2_MAX_CHUNKS; ++k) {
                    mk[k] = -3.0e38f; zk[k] = 0.f;
#pragma unroll
                    for (int r = 0; r < EPT; ++r) dk[k][r] = 0.f;
                    if (k < nc) {
                        const granule2_t* theirs = slot0 + (size_t)k * F2_XCHG;
                        const granule2_t gm = ld_granule2(theirs + 0), gz = ld_granule2(theirs + 1);
                        ok = ok && (unsigned)(gm >> 32) == tag && (unsigned)(gz >> 32) == tag;
                        mk[k] = __uint_as_float((unsigned)gm);
                        zk[k] = __uint_as_float((unsigned)gz);
#pragma unroll
                        for (int r = 0; r < EPT; ++r) {
                            const int e = th + F2_HT * r;
                            const granule2_t gd = ld_granule2(theirs + 2 + (e < N ? e : 0));
                            ok = ok && (unsigned)(gd >> 32) == tag;
                            dk[k][r] = (e < N) ? __uint_as_float((unsigned)gd) : 0.f;
                        }
                    }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > F2_TIMEOUT_TICKS) { *s_abort = 1; break; }
            }
            m_all = mk[0];
#pragma unroll
            for (int k = 1; k < F2_MAX_CHUNKS; ++k) m_all = fmaxf(m_all, mk[k]);
            z_all = 0.f;
#pragma unroll
            for (int r = 0; r < EPT; ++r) dsum[r] = 0.f;
#pragma unroll
            for (int k = 0; k < F2_MAX_CHUNKS; ++k) {
                if (k < nc) {
                    const float f = expf(mk[k] - m_all);
                    z_all = fmaf(f, zk[k], z_all);
#pragma unroll
                    for (int r = 0; r < EPT; ++r) dsum[r] = fmaf(f, dk[k][r], dsum[r]);
                }
            }
            f_own = expf(mb - m_all);
        }
        // ============ E. weights out, delta (transposed) -> mean += lr * Sigma @ delta
        if (th < F2_HW && sl < S) weights[(size_t)p * S + sl] = ex * f_own / z_all;
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int e = th + F2_HT * r;
            if (e < N) {
                const int hh = e / DCH, cc = e - hh * DCH;
                delta[cc * F2_LD + hh] = dsum[r] / z_all;
            }
        }
        half_barrier(bar, bgen, s_abort);                                                       // (4) delta complete
        if (*s_abort) break;                                                                    // uniform over the half (set before barrier 4)
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int e = th + F2_HT * r;
            if (e < N) {
                const int hh = e / DCH, cc = e - hh * DCH;
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
                const float4* dcol = reinterpret_cast<const float4*>(delta + cc * F2_LD);
                const float4* srow = reinterpret_cast<const float4*>(sig_l + hh * F2_LD);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float4 dv = dcol[k], sv = srow[k];
                    a4[k & 3] = fmaf(sv.x, dv.x, a4[k & 3]);
                    a4[k & 3] = fmaf(sv.y, dv.y, a4[k & 3]);
                    a4[k & 3] = fmaf(sv.z, dv.z, a4[k & 3]);
                    a4[k & 3] = fmaf(sv.w, dv.w, a4[k & 3]);
                }
                mean_l[e] += lr * ((a4[0] + a4[1]) + (a4[2] + a4[3]));
            }
        }
        half_barrier(bar, bgen, s_abort);                                                       // (5) new mean visible, tiles free
        if (*s_abort) break;
    }
    if (*s_abort) {
        if ((tid & (F2_HT - 1)) == 0) st_agent2(err_word, __uint_as_float(tag0));
        return;
    }
    if (chunk == 0) {
        const int th = tid & (F2_HT - 1);
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int e = th + F2_HT * r;
            if (e < N) means[(size_t)p * N + e] = mean_l[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launcher (called by mpb_stomp_run, mpb_stomp_fused.hip)
// ------------------------------------------------------------------------------------------------
size_t mpb_fused2_ws_floats(int P, int S) {
    const int nc = (S + F2_HW - 1) / F2_HW;
    return 16 + 2 * 2 * (size_t)P * nc * F2_XCHG;
}

// single field with a usable grid, H = 64, S <= 32, d in the instantiated set; returns false when it does not apply
bool mpb_fused2_launch(float* means, const float* eps, float* samples, float* costs, float* weights, const float* L,
                       const float* Sigma, const float* geom, int geom_flags, float* workspace, int P, int S, int d,
                       float k_sigma, float weight, float lr, float temperature, int n_iters, uint32_t lo, uint32_t hi,
                       uint32_t iter0, uint32_t particle_offset, uint32_t tag0, hipStream_t st) {
    const int nc = (S + F2_HW - 1) / F2_HW;
    if (nc > F2_MAX_CHUNKS) return false;
    const int n_pairs = (P + 1) / 2;
    const dim3 grid(n_pairs * nc), block(F2_THREADS);
    const int model = geom_flags & 0xFF;
#define MPB_F2_CASE(DCH, MODEL)                                                                                       \
    hipLaunchKernelGGL((stomp_fused2_kernel<DCH, MODEL>), grid, block, 0, st, means, eps, samples, costs, weights, L, \
                       Sigma, geom, workspace, P, S, nc, k_sigma, weight, lr, temperature, n_iters, lo, hi, iter0,      \
                       particle_offset, tag0)
    if (model == PandaModel::ID && d == 7) MPB_F2_CASE(7, PandaModel::ID);
    else if (model == PandaModel::ID && d == 14) MPB_F2_CASE(14, PandaModel::ID);
    else if (d == 7) MPB_F2_CASE(7, 0);
    else if (d == 14) MPB_F2_CASE(14, 0);
    else return false;
#undef MPB_F2_CASE
    return true;
}
