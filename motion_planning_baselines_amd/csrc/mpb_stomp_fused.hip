// mpb_stomp_fused.hip -- the whole STOMP optimisation loop (stomp.py:150-160) as ONE persistent launch.
//
// The two-kernel path (mpb_kernels.hip) pays, per iteration, two dispatch ramps (a launch that fills the chip once
// needs ~4.3 us to start its 4096 waves, measured: scripts/launch_ramp.hip), two dependent-dispatch gaps (~1.7 us
// each) and re-stages its constants; at C3 that is ~40 % of a 25 us iteration.  Here a workgroup of 16 waves owns one
// UNIT = (particle p, chunk of 16 of its S samples) for all n_iters iterations:
//   * wave = rollout (sample), lane = waypoint, exactly the mapping of kernel A: noise product on the matrix cores,
//     FK + SDF in registers, the per-rollout cost a wave reduction;
//   * L (permuted MFMA image), Sigma, the broad-phase grid + obstacle table and the particle mean live in LDS for the
//     whole launch; the samples of an iteration stay in the waves' LDS tiles for the weighted-noise reduction
//     (stomp.py:199-211), which therefore reads LDS instead of re-reading the 14.7 MB of samples from L2;
//   * a particle with S > 16 is shared by nc = ceil(S / 16) workgroups: each reduces its own 16 samples to
//     (m_k = max logit, z_k = sum exp(logit - m_k), D_k = sum exp(logit - m_k) (sample - mean)), publishes that (3.6 KB)
//     and combines the nc partials IN CHUNK ORDER -- every partner computes bit-identical new means, so the copies
//     of the particle never drift apart.  Hand-off through global memory with agent-scope (sc1) stores / loads and a
//     flag per unit (MI355X_MICROARCH.md, "Valid forms"); the noise of the NEXT iteration (Philox + MFMA, which does
//     not depend on the means) is computed between publishing and polling, so the partner's latency is hidden;
//   * softmax algebra: w_s = exp(x_s - m) / z with m = max_k m_k, z = sum_k exp(m_k - m) z_k -- the same weights as
//     softmax(-c / T) up to rounding (the two-kernel path normalises before summing; both are within 1e-6 of fp64).
// Partners are paired by TICKET, not by block index: every workgroup of the exchange layout draws a ticket from a counter
// in the workspace header when it starts, and consecutive tickets of a pool own the chunks of one particle.  Tickets are handed out
// in the order the workgroups actually start, so the partners of a particle are always the workgroups that started next
// to each other -- nothing is assumed about dispatch order or co-residency (HIP promises neither), and a grid of any
// size makes progress as long as the device keeps starting its workgroups.  Every wait is still bounded
// (s_memrealtime): a workgroup whose partner does not show up in time raises the error word -- in the workspace header
// and, when the caller passed one, in a host-visible status block -- and every workgroup that starts afterwards leaves
// at once; the caller (planners/stomp.py) turns that into an exception.  The last workgroup out resets the counters.
// With at least as many particles as CUs the launcher picks the other layout of the same kernel (template NB = 2): ONE
// workgroup per particle runs the particle's S <= 32 samples as two batches of 16 and keeps the batches' partials in
// registers -- no exchange, five instead of six block barriers and one update instead of two per particle and iteration
// (C5's per-GPU load: 0.531 -> 0.497 ms / iteration); the partials are combined with the expressions of the exchange
// path, so both layouts produce the same bits (tests/test_gpu_stomp_fused.py).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <chrono>

#include "mpb_common.h"
#include "mpb_geom.h"
#include "mpb_stomp_noise.h"
#include "mpb_stomp_fused.h"

// workspace layout: FUSED_HDR_WORDS floats of header ([0] = error word; mpb_stomp_fused.h), then 2 parities x P x nc x FUSED_XCHG granules (8 B each):
// granule 0 = m, 1 = z, 2 + t = partial sum of trajectory element t
static inline size_t fused_ws_floats(int P, int nc) { return FUSED_HDR_WORDS + 2 * 2 * (size_t)P * nc * FUSED_XCHG; }

#ifdef MPB_STAMPS   // diagnostic build only: s_memtime per wave at the phase boundaries of iteration 2
__device__ unsigned long long g_fstamps[256 * FUSED_WAVES * 12];
#define FSTAMP(k)                                                                                   \
    do {                                                                                            \
        if (it == 2 && blockIdx.x < 256) {                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            unsigned long long t_;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if ((k) == 0) {   /* stamp 0 carries the SIMD the wave runs on in its top byte */        \
                unsigned hw_;                                                                       \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(hw_));             \
                t_ = (t_ & 0x00FFFFFFFFFFFFFFull) | ((unsigned long long)hw_ << 56);                \
            }                                                                                       \
            if ((threadIdx.x & 63) == 0) g_fstamps[(blockIdx.x * FUSED_WAVES + (threadIdx.x >> 6)) * 12 + (k)] = t_; \
        }                                                                                           \
    } while (0)
extern "C" int mpb_debug_read_fstamps(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 3;
}
// s_memrealtime (100 MHz) of wave 0 of every workgroup along the LAUNCH: entry, ticket drawn, constants staged, first noise
// drawn, end of iterations 0 / 1 / 2, exit
__device__ unsigned long long g_lstamps[1024 * 8];
#define LSTAMP(k)                                                                                   \
    do {                                                                                            \
        if (blockIdx.x < 1024 && threadIdx.x == 0) g_lstamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
extern "C" int mpb_debug_read_lstamps(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 3;
}
#else
#define FSTAMP(k)
#define LSTAMP(k)
#endif

// PRE: the exchange layout's speculative copy of the partners' partials (below: "exchange prefetch"), a kilobyte per wave
#ifdef FUSED_T_PRE_STATS   // tuning build: lanes x iterations whose prefetched granules were stale, by wave
__device__ unsigned g_pre_miss[FUSED_WAVES];
extern "C" int mpb_debug_read_pre_miss(unsigned* dst) {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pre_miss), sizeof(unsigned) * FUSED_WAVES) != hipSuccess) return 3;
    unsigned z[FUSED_WAVES] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pre_miss), z, sizeof(z)) == hipSuccess ? 0 : 3;
}
#endif
template <int DCH, bool PRE>
struct FusedSmem {
    float4 otab[MPB_GRID_MAX_SPH + 1];                       //  1 KB: obstacle table + the far dummy
    unsigned gridw[MPB_GRID_MAX_CELLS];                      // 16 KB: broad-phase grid (offset words)
    unsigned Limg[STOMP_LIMG_WORDS];                         // 18 KB: L as three bf16 components
    float sig_l[64 * FUSED_LD];                              // 17 KB: Sigma, padded rows
    float mean_l[64 * DCH];                                  // 3.5 KB
    float delta[DCH * FUSED_LD];                             // 3.7 KB (transposed)
    float cst[FUSED_WAVES];
    int s_abort;
    unsigned s_ticket;
    unsigned pad_[2];
    float tiles[FUSED_WAVES * 64 * NT_STRIDE];               // 80 KB: the waves' sample tiles
    __attribute__((aligned(16))) granule_t pre[PRE ? FUSED_WAVES * 128 : 2];     // 16 KB: per wave, its 64 elements' granules of chunks 0 and 1
    __attribute__((aligned(16))) granule_t premz[PRE ? FUSED_WAVES * 4 : 2];     // per wave: (m, z) of chunks 0 and 1
};

// NB = 1: the unit is (particle, chunk of 16 samples), partners exchange partials through the workspace (above).
// NB = 2: one workgroup per particle runs its S <= 32 samples as two batches of 16, one after the other, and keeps the
//         batches' partials in registers -- no exchange, five instead of six block barriers per particle and iteration, one
//         update instead of two.  For loads with at least as many particles as CUs (C5); the partials are combined with
//         the very expressions of the exchange path, so both layouts produce the same bits.
// CHAIN = false: ONE collision field by construction (geom_flags bit 12; only instantiated for the compile-time robot models): the
// loop over chained fields and its second GeomView are gone -- 7 / 16 SGPR spills less, c5 -1.9 %, C3 -0.4 % (round 5).
template <int DCH, int MODEL, int NB, bool INJ, bool CHAIN>
__global__ __launch_bounds__(FUSED_THREADS, 4) void stomp_fused_kernel(
    float* __restrict__ means, const float* __restrict__ eps, float* __restrict__ samples, float* __restrict__ costs,
    float* __restrict__ weights, const float* __restrict__ Lmat, const float* __restrict__ Sigma,
    const float* __restrict__ geom, float* __restrict__ ws, int P, int S, int nc, float k_sigma, float weight, float lr,
    float temperature, int n_iters, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter0, uint32_t particle_offset,
    uint32_t tag0, unsigned long long timeout_ticks, unsigned* __restrict__ status_host, float* __restrict__ means_copy) {
    constexpr int H = 64;
    constexpr int N = H * DCH;                    // elements of a trajectory
    static_assert(N <= FUSED_THREADS && N + 2 <= FUSED_XCHG, "one thread per trajectory element");
    // INJ = false: the instantiation of the device-noise calls (eps == NULL: what the planners run by default and what
    // bench.py times) does not carry the injected-noise path at all -- its pointer, strides and loads cost that path
    // thirteen SGPR spills and 2.2 % of the C3 iteration (round 5).  The launcher picks by eps; the draw, split and product
    // are the same functions either way, and tests/test_gpu_philox_vs_oracle.py holds the two instantiations to the same bits.
    if (!INJ) eps = nullptr;
    // ONE shared object with the layout fixed by hand: what the walk reads at random -- the obstacle table, the grid -- and the
    // constants of the other phases sit in the first 64 KB, where an LDS instruction's 16-bit offset field reaches them (left
    // to the linker the table landed at 141 536: a v_add per read to form the address, four per group of spheres and trip);
    // the 80 KB of sample tiles come last
    // exchange prefetch: the layout with partners, one rollout per draw (the paired draw of d <= 8 leaves half the waves out of
    // the noise phase; not built for it)
    // (nor for the table-driven robot and the injected-noise instantiations: they sit at the register budget, and the handful of
    // address registers of the prefetch tipped 10-19 VGPRs of theirs into scratch)
#ifdef FUSED_NO_PRE   // (A/B builds)
    constexpr bool PRE = false;
#else
    constexpr bool PRE = NB == 1 && 2 * DCH > 16 && MODEL != 0 && !INJ;
#endif
    __shared__ __attribute__((aligned(16))) FusedSmem<DCH, PRE> sm;
    float4 (&otab)[MPB_GRID_MAX_SPH + 1] = sm.otab;
    unsigned (&gridw)[MPB_GRID_MAX_CELLS] = sm.gridw;
    unsigned (&Limg)[STOMP_LIMG_WORDS] = sm.Limg;
    float (&sig_l)[64 * FUSED_LD] = sm.sig_l;
    float (&mean_l)[64 * DCH] = sm.mean_l;
    float (&delta)[DCH * FUSED_LD] = sm.delta;
    float (&cst)[FUSED_WAVES] = sm.cst;
    float (&tiles)[FUSED_WAVES * 64 * NT_STRIDE] = sm.tiles;
    int& s_abort = sm.s_abort;
    unsigned& s_ticket = sm.s_ticket;

    // the wave index as a SCALAR: everything derived from it (sample index, tile and output base addresses) then sits in
    // SGPRs, and per-lane addresses are a 32-bit offset from a uniform base instead of hoisted 64-bit VGPR pairs (which spill)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LSTAMP(0);
    unsigned* wsu = reinterpret_cast<unsigned*>(ws);
    const bool exchange = NB == 1 && nc > 1;
    // unit of this block: with partners to exchange with, by ticket (drawn here, read after the constants are staged);
    // otherwise (one workgroup per particle) by block index
    if (tid == 0) {
        s_abort = 0;
        fused_stamp_begin(wsu, status_host);
        unsigned u = blockIdx.x;
        if (exchange) {
            int why;
            u = fused_draw_unit(wsu, P, nc, tag0, why);
            s_abort = why;
        }
        s_ticket = u;
    }
    LSTAMP(1);
    const int j = lane & 15, g = lane >> 4;
    granule_t* xch = reinterpret_cast<granule_t*>(ws + FUSED_HDR_WORDS);

    // ---- constants into LDS (once)
    GeomView G0 = geom_view(geom);
    if (!CHAIN && G0.next != 0) G0.model = 0;    // geom_flags promised ONE field and the device header chains another: the model
                                                 // tag check of phase B then poisons every cost (never a silent mis-read)
    {
        const int g_rounds = (G0.n_cells + 4 * FUSED_THREADS - 1) / (4 * FUSED_THREADS);   // MPB_GRID_PAD = 1024 words: whole rounds of 256 lanes
        const uint4* g4 = reinterpret_cast<const uint4*>(G0.grid);
        // 1024 threads x 16 B = 4 rounds' worth of the 256-thread padding unit per pass; lanes beyond the padded section stay out
        const int n_pad16 = (G0.n_cells + MPB_GRID_PAD - 1) / MPB_GRID_PAD * (MPB_GRID_PAD / 4);   // uint4s in the padded section
        for (int u = 0; u < (MPB_GRID_MAX_CELLS / 4 + FUSED_THREADS - 1) / FUSED_THREADS; ++u) {
            const int i = tid + FUSED_THREADS * u;
            if (u < g_rounds && i < n_pad16) {
                // (as offset words: mpb_geom.h, grid_offset_word)
                const uint4 gw = g4[i];
                const unsigned ns = (unsigned)G0.n_sph;
                reinterpret_cast<uint4*>(gridw)[i] = make_uint4(grid_offset_word(gw.x, ns), grid_offset_word(gw.y, ns),
                                                                grid_offset_word(gw.z, ns), grid_offset_word(gw.w, ns));
            }
        }
        for (int i = tid; i <= G0.n_sph && i <= MPB_GRID_MAX_SPH; i += FUSED_THREADS)
            otab[i] = (i < G0.n_sph) ? reinterpret_cast<const float4*>(G0.sph)[i] : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
        // L as the three-component bf16 MFMA image (mpb_stomp_noise.h) and the padded Sigma image: one float4 per thread each
        const f32x4 lv = reinterpret_cast<const f32x4*>(Lmat)[tid];
        const f32x4 sv = reinterpret_cast<const f32x4*>(Sigma)[tid];
        const int row = tid >> 4, col0 = (tid & 15) << 2;
        // rows 0 and H - 1 of the image are ZERO: stomp.py:105-106 zeroes the noise of the first and the last waypoint, and a
        // zero row of L makes those rows of L eps exact zeros for free -- the sample phase adds the noise row without a select
        // (fourteen exec-mask regions per wave and iteration until round 4)
        const bool edge_row = row == 0 || row == H - 1;
        stomp_l_image_store(Limg, row, col0, edge_row ? f32x4{0.f, 0.f, 0.f, 0.f} : lv);
        *reinterpret_cast<f32x4*>(sig_l + row * FUSED_LD + col0) = sv;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane((int)s_ticket);
    const int p = exchange ? ticket / nc : ticket;
    const int chunk = exchange ? ticket - p * nc : 0;
    int s = chunk * FUSED_WAVES + wave;                 // this wave's sample (NB > 1: of the current batch)
    bool live = s < S;
    // the particle's mean: fetched now, dropped into LDS behind the first noise draw (which needs the unit, not the mean: its
    // ~1.7 us of Philox + matrix work hide the load's round trip at kernel entry)
    // (exchange layout only: in the two-batch instantiations the value carried across the draw pushed 14 VGPRs to scratch)
    float mean_reg = 0.f;
    if (NB == 1) mean_reg = (tid < N) ? means[(size_t)p * N + tid] : 0.f;
    else if (tid < N) mean_l[tid] = means[(size_t)p * N + tid];
    // header word 1 = this call's tag; word 0 (the error word) counts only when it equals it
    if (ticket == 0 && tid == 0) st_agent_u(wsu + FUSED_HDR_TAG, tag0);    // (unit 0 is drawn exactly once)
    LSTAMP(2);

    const size_t eps_stride = (size_t)S * DCH * P * H;
    const float inv_temperature = 1.0f / temperature;

    // ---- noise of iteration 0: straight into the wave's tile ([waypoint][channel], stride NT_STRIDE)
    // (PAIRED, 2 DCH <= 16: waves 0-7 draw and multiply for two rollouts each -- their own and wave + 8's --, waves 8-15 skip
    // the phase: mpb_stomp_noise.h, stomp_noise_bf16_pair)
#ifdef FUSED_NO_PAIR   // (A/B builds)
    constexpr bool PAIRED = false;
#else
    constexpr bool PAIRED = 2 * DCH <= 16;
#endif
    constexpr int PAIR_STRIDE = FUSED_WAVES / 2;
    float* nt = tiles + wave * (H * NT_STRIDE);
    if constexpr (PAIRED) {
        if (wave < PAIR_STRIDE) {
            f32x4 acc[4];
            stomp_noise_bf16_pair<DCH>(Limg, acc, eps, P, S, p, j, g, particle_offset + (uint32_t)p, (uint32_t)s, (uint32_t)(s + PAIR_STRIDE),
                                       iter0, seed_lo, seed_hi);
            stomp_noise_to_tile_pair<DCH>(nt, nt + PAIR_STRIDE * (H * NT_STRIDE), acc, lane);
        }
    } else {
        f32x4 acc[4];
        stomp_noise_bf16<DCH>(Limg, acc, eps ? eps + (size_t)(live ? s : 0) * DCH * P * H : nullptr, P, p, j, g,
                              particle_offset + (uint32_t)p, (uint32_t)s, iter0, seed_lo, seed_hi);
        stomp_noise_to_tile(nt, acc, lane);
    }
    if (NB == 1 && tid < N) mean_l[tid] = mean_reg;
    __syncthreads();

    LSTAMP(3);
    const int n_run = s_abort ? 0 : n_iters;          // (block-uniform: written before the barriers above)
#ifdef FUSED_T_PRE_STATS
    int pre_miss_ = 0;
#endif
    for (int it = 0; it < n_run; ++it) {
        // (NB > 1) partials of the batches, as the exchange path would publish them
        float pm0 = -3.0e38f, pz0 = 0.f, pe0 = 0.f, pm1 = -3.0e38f, pz1 = 0.f, pe1 = 0.f;
        f32x4 sd0 = {0.f, 0.f, 0.f, 0.f}, sd1 = {0.f, 0.f, 0.f, 0.f};      // Sigma times the partial of batch 0 / 1 (or of the only chunk), waves 0-3
        float mb = 0.f, zb = 0.f, ex = 0.f, dpart = 0.f;
        int tq = 0, hh = 0, cc = 0, sl = 0;
        // Sigma (64 x 64) times a (64 x d) tile held transposed in `delta`, on the matrix pipe: waves 0-3 (one per SIMD) own 16
        // rows each; the k index of a lane is 16 (lane >> 4) + 4 q + e, so that both operands are read as four 16-byte pieces
        // per lane -- every word of Sigma and delta leaves LDS once per workgroup (the per-element form read a whole row and a
        // whole column per thread: 458 KB of LDS traffic per iteration, 2.3 us of a 16.8 us iteration by duplication).
        // Returns acc[r] = row 16 wave + 4 lg + r, channel li (lanes li >= DCH: zeros).
        auto sigma_times_delta = [&](int li, int lg) {
            const float* arow = sig_l + (16 * wave + li) * FUSED_LD + 16 * lg;
            const float* brow = delta + (li < DCH ? li : 0) * FUSED_LD + 16 * lg;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 4 * q);
                f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 4 * q);
                if (li >= DCH) bv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
            }
            return acc;
        };
#pragma nounroll
        for (int bt = 0; bt < NB; ++bt) {
        if (NB > 1) { s = bt * FUSED_WAVES + wave; live = s < S; }
        // ============ A. samples of this iteration: x = mean + noise, stored, kept packed in the wave's tile
        FSTAMP(0);
        float nz[16];
        stomp_noise_row<DCH>(nt, lane, nz);
        const int h = lane;
        float x[DCH];
        // (the 16 waves of the block do this at the same moment: the LDS pipe, not the VALU, paces this phase -- 8-byte
        // accesses where the row length allows it: DCH even -> every row starts 8-byte aligned)
        if (DCH % 2 == 0) {
#pragma unroll
            for (int c = 0; c < DCH; c += 2) {
#ifdef FUSED_T_A_NOMEAN   // (wrong-result timing switch, tuning builds only)
                const float2 mv = make_float2(0.f, 0.f);
#else
                const float2 mv = *reinterpret_cast<const float2*>(mean_l + h * DCH + c);
#endif
                x[c] = mv.x + nz[c];                      // (rows 0 / H - 1 of the noise are exact zeros: the L image's rows are)
                x[c + 1] = mv.y + nz[c + 1];
            }
#pragma unroll
            for (int c = 0; c < DCH; c += 2) *reinterpret_cast<float2*>(nt + h * DCH + c) = make_float2(x[c], x[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < DCH; ++c) x[c] = mean_l[h * DCH + c] + nz[c];
#pragma unroll
            for (int c = 0; c < DCH; ++c) nt[h * DCH + c] = x[c];
        }
        __builtin_amdgcn_wave_barrier();
#ifdef FUSED_T_A_NOSTORE   // (wrong-result timing switch, tuning builds only)
        if (false) {
#else
        if (live) {
#endif
            const f32x4* pk4 = reinterpret_cast<const f32x4*>(nt);
            f32x4* out4 = reinterpret_cast<f32x4*>(samples + ((size_t)p * S + s) * N);     // uniform
#pragma unroll
            for (int k = 0; k < (16 * DCH + 63) / 64; ++k) {
                const unsigned idx = (unsigned)lane + 64u * k;
                if (idx < 16u * DCH) out4[idx] = pk4[idx];
            }
        }
        // ============ B. collision cost of the rollout
        FSTAMP(1);
        {
            float q[MPB_MAX_DOF];
            // (a row of d channels holds D = d or D = d / 2 joint positions: with more channels than MPB_MAX_DOF it must be d / 2 -- the
            // joints beyond that are zeros the compiler can fold, which keeps the d = 14 kernels at the registers they had with 8)
            constexpr int DQ_ = (DCH > MPB_MAX_DOF) ? DCH / 2 : DCH;
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < DQ_) ? x[i < DQ_ ? i : 0] : 0.f;
            float c = 0.f;
            bool bad = false;
            GeomView G = G0;
            for (const float* gp = geom;;) {
                if (gp != geom) {     // a chained field: its grid replaces the first one's (restored before the next iteration)
                    __syncthreads();
                    grid_stage_offsets(G, gridw, otab, tid, FUSED_THREADS);
                    __syncthreads();
                }
#ifdef FUSED_T_SKIP_COST   // (wrong-result timing switch, tuning builds only)
                if (false) {
#else
                if (live && h >= 1) {
#endif
                    if (MODEL == PandaModel::ID) {
                        if (G.model == PandaModel::ID) c = fmaf(G.fscale, waypoint_cost_grid_model<PandaModel, true>(G, gridw, otab, q), c);
                    } else {
                        c = fmaf(G.fscale, waypoint_cost_grid<true>(G, gridw, otab, q), c);
                    }
                }
                if (MODEL != 0 && G.model != MODEL) bad = true;    // (wave-uniform: lane 0 -- waypoint 0, outside the walk -- writes the cost)
                if (!CHAIN) break;            // (one field by construction: the launcher read geom_flags bit 12)
                if (G.next == 0) break;
                gp += G.next;
                G = geom_view(gp);
            }
            if (CHAIN && G0.next != 0) {       // more than one field: put the first field's grid back for the next iteration
                __syncthreads();
                // (from an opaque copy of the thread index: the per-thread source addresses of this rare path are loop invariants
                // the compiler would otherwise carry -- spill -- through every iteration)
                int tid_o = tid;
                asm volatile("" : "+v"(tid_o));
                grid_stage_offsets(G0, gridw, otab, tid_o, FUSED_THREADS);
            }
            const double csum = wave_sum_f64((double)c);
            const float cw = weight * (k_sigma * (float)csum);
            if (lane == 0) {
                cst[wave] = cw;
                if (live) {
                    if (bad) reinterpret_cast<unsigned*>(costs)[(size_t)p * S + s] = 0x7FC00000u;
                    else costs[(size_t)p * S + s] = cw;
                }
            }
        }
        FSTAMP(2);
        __syncthreads();                                                                        // (1) costs of the chunk
        FSTAMP(3);
        // ============ C. partial of this chunk: logits, local max, e_w, z, weighted (sample - mean)
        // (thread-index arithmetic of the update phases is redone here from an opaque copy: hoisted out of the loop it
        // would sit in registers through the cost phase, which has none to spare)
        tq = tid;
        asm volatile("" : "+v"(tq));
        const int lq = tq & 63;
        hh = (tq < N) ? tq / DCH : 0; cc = (tq < N) ? tq - hh * DCH : 0;   // the trajectory element this thread owns
        sl = (NB > 1 ? bt : chunk) * FUSED_WAVES + (lq & 15);              // every wave redundantly, lanes 0-15 carry the chunk
        // (all FOUR rows of a wave carry the chunk's sixteen costs: the softmax statistics are row reductions -- no v_readlane, no
        // combine across rows; same bits as the full-wave forms over one row and three rows of neutral elements -- and the weight of
        // sample w reaches a lane's fma through DPP row_newbcast instead of a v_readlane and a scalar register)
        const float xs = (sl < S) ? -cst[lq & 15] * inv_temperature : -3.0e38f;
        mb = row_max_f32(xs);
        ex = (sl < S) ? fast_expf(xs - mb) : 0.f;
        zb = row_sum_f32(ex);
        dpart = 0.f;
        if (tq < N) {       // (N = 64 DCH: whole waves -- DPP reads the neighbours' registers, every lane of a wave that is here is active)
            const float mu = mean_l[tq];
            // dpart = fma(ex[row lane w], df[w], dpart), w ascending: v_fmac_f32 with its first source through DPP row_newbcast, the
            // sixteen of them and their wait states in one asm statement (fmac_row_bcast_seq, mpb_common.h)
            float df[FUSED_WAVES];
#pragma unroll
            for (int w = 0; w < FUSED_WAVES; ++w) df[w] = tiles[w * (H * NT_STRIDE) + tq] - mu;
            fmac_row_bcast_seq(dpart, ex, df);
        }
        if (NB > 1) {            // (wave-uniform) keep this batch's partial
            if (bt == 0) { pm0 = mb; pz0 = zb; pe0 = ex; } else { pm1 = mb; pz1 = zb; pe1 = ex; }
        }
        FSTAMP(4);
        const unsigned tag = tag0 + (unsigned)it;                  // unique per (call, iteration): stale granules never match
        // ============ D. Sigma times THIS chunk's (batch's) partial, before any combining (round 4): Sigma (sum_k f_k D_k) =
        //              sum_k f_k (Sigma D_k), so the matrix product can run per partial ahead of the exchange instead of on
        //              the combined sum after it.  What follows the poll is then elementwise -- combine, divide, add to the
        //              mean, one barrier -- where it used to be delta -> LDS, barrier, matrix product, barrier: the partner's
        //              round trip no longer has an LDS round trip and a barrier queued behind it.  m and z go out first; behind
        //              the barrier that also says "the samples in the tiles are consumed" waves 0-3 multiply -- and publish
        //              (exchange layout) or keep (two batches / a single chunk) the product -- while the other waves start the
        //              next iteration's noise.  All layouts combine the same products with the same expressions: same bits.
        if (NB == 1 && nc > 1 && tq == 0) {
            granule_t* mine = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc + chunk) * FUSED_XCHG;
            st_granule(mine + 0, mb, tag);
            st_granule(mine + 1, zb, tag);
        }
        if (tq < N) delta[cc * FUSED_LD + hh] = dpart;
        __syncthreads();                                                     // (2) the samples in the tiles are consumed, delta complete
        if (wave < 4) {
            // (round 6: at the highest issue priority.  These four waves left the previous noise phase at priority 0 and the other
            // twelve enter the next one at 3: the product and the partner's granules waited until the draws were through --
            // the stats build of the exchange prefetch showed the publish landing in the second half of the phase)
#ifndef FUSED_NO_PUBPRIO
            if (NB == 1) __builtin_amdgcn_s_setprio(3);
#endif
            const int li = tq & 15, lg = (tq >> 4) & 3;
            const f32x4 sd = sigma_times_delta(li, lg);
            if (NB == 1 && nc > 1) {
                granule_t* mine = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc + chunk) * FUSED_XCHG;
                if (li < DCH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_granule(mine + 2 + (16 * wave + 4 * lg + r) * DCH + li, sd[r], tag);
                }
            } else if (NB > 1 && bt > 0) {
                sd1 = sd;
            } else {
                sd0 = sd;
            }
        }
        FSTAMP(5);
        const bool more_batches = NB > 1 && bt + 1 < NB;
        const int it_n = more_batches ? it : it + 1;                       // (iteration, batch) whose noise is drawn now
        const int s_n = NB > 1 ? (more_batches ? bt + 1 : 0) * FUSED_WAVES + wave : s;
#ifdef FUSED_T_SKIP_NOISE   // (wrong-result timing switch, tuning builds only: is the draw on the iteration's critical path?)
        if (false) {
#else
        if (it_n < n_run) {
#endif
            // the noise of the NEXT iteration (it does not depend on the means) is drawn here, between publishing and polling:
            // the partner's latency.  (Drawing it before barrier 1 at the lowest issue priority, to fill the wait for the
            // block's slowest rollout, was measured 15 % slower: the rollouts leave few issue slots free, and the matrix
            // work that follows then runs with no Philox of another wave to overlap with.)
            f32x4 acc[4];
            // (opaque copies: the first Philox round multiplies two counter words that do not change from one iteration to
            // the next, and the compiler would hoist those products out of the loop and keep -- spill -- them)
            int jv = j, gv = g;
            asm volatile("" : "+v"(jv), "+v"(gv));
            // (likewise the key: the seven round keys seed + r * W are loop invariants the compiler hoists -- twelve SGPRs that
            // do not survive the cost phase, i.e. 52 v_readlane of SGPR-spill reloads per iteration on the pipe that binds the
            // kernel, where recomputing them is twelve s_add on the scalar unit)
            uint32_t slo = seed_lo, shi = seed_hi;
            asm volatile("" : "+s"(slo), "+s"(shi));
            // (a k-block pipelined form -- Philox of block q+1 issued between the MFMAs of block q, straight-line code -- was
            // measured 3 % slower: the waves of a SIMD already overlap one wave's matrix work with another's Philox)
            // issue priority by progress through the draws (mpb_stomp_noise.h).  Measured against it on the same box
            // (scripts/ab_k20.sh): four fixed levels for the four waves of a SIMD, so that one wave's matrix product would run
            // under the others' draws (STOMP_PRIO_STAGGER) +3 %, no priority at all +3 %
#ifndef FUSED_NOISE_PRIO
#define FUSED_NOISE_PRIO STOMP_PRIO_PROGRESS
#endif
            // (the tile addresses of the write below from an opaque copy of the lane: hoisted out of the loop they are four VGPRs
            // the cost phase does not have)
            int lane_w = lane;
            asm volatile("" : "+v"(lane_w));
            if constexpr (PAIRED) {
                if (wave < PAIR_STRIDE) {
                    stomp_noise_bf16_pair<DCH, FUSED_NOISE_PRIO>(Limg, acc, eps ? eps + (size_t)it_n * eps_stride : nullptr, P, S, p, jv, gv,
                                                                 particle_offset + (uint32_t)p, (uint32_t)s_n, (uint32_t)(s_n + PAIR_STRIDE),
                                                                 iter0 + (uint32_t)it_n, slo, shi, 3 - (wave >> 1));
                    // (the partner's tile: its samples were consumed before barrier 2 like this wave's own)
                    stomp_noise_to_tile_pair<DCH>(nt, nt + PAIR_STRIDE * (H * NT_STRIDE), acc, lane_w);
                }
            } else {
                // EXCHANGE PREFETCH (round 6).  The poll after this phase was a full round trip of device-coherent loads on the
                // iteration's critical path (~1.6 k of 28.5 k cycles: the partner published long before, the loads only left
                // when the phase was over; issued into registers ahead of the phase they cost more than they hid, round 4).
                // LDS-DMA needs no registers: half-way through the draw every wave copies the granules its own threads will
                // ask for -- 64 elements x 2 chunks, 16 B per lane, and (m, z) of both chunks by two lanes -- into its own
                // kilobyte of LDS (sc1: coherent at the device like the atomic loads of the poll).  The copy is SPECULATIVE: a
                // granule whose tag is not this iteration's is simply not there yet, and its thread falls back to the poll.
                auto prefetch = [&]() {
                    if (PRE && nc == 2) {
                        int lp = lane;
                        asm volatile("" : "+v"(lp));          // (addresses formed here, not carried through the cost phase)
                        const granule_t* slot0 = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc) * FUSED_XCHG;
                        const int wv = wave < DCH ? wave : DCH - 1;                  // (waves beyond the trajectory: any valid piece)
                        const granule_t* gsrc = slot0 + (size_t)(lp >> 5) * FUSED_XCHG + 2 + 64 * wv + 2 * (lp & 31);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                                         (__attribute__((address_space(3))) void*)(sm.pre + wave * 128), 16, 0, 16);
                        if (lp < 2)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slot0 + (size_t)lp * FUSED_XCHG),
                                                             (__attribute__((address_space(3))) void*)(sm.premz + wave * 4), 16, 0, 16);
                    }
                };
                if constexpr (PRE)
                    stomp_noise_bf16<DCH, FUSED_NOISE_PRIO>(Limg, acc, eps ? eps + (size_t)it_n * eps_stride + (size_t)(s_n < S ? s_n : 0) * DCH * P * H : nullptr,
                                                            P, p, jv, gv, particle_offset + (uint32_t)p, (uint32_t)s_n, iter0 + (uint32_t)it_n,
                                                            slo, shi, 3 - (wave >> 2), prefetch);
                else
                    stomp_noise_bf16<DCH, FUSED_NOISE_PRIO>(Limg, acc, eps ? eps + (size_t)it_n * eps_stride + (size_t)(s_n < S ? s_n : 0) * DCH * P * H : nullptr,
                                                            P, p, jv, gv, particle_offset + (uint32_t)p, (uint32_t)s_n, iter0 + (uint32_t)it_n,
                                                            slo, shi, 3 - (wave >> 2));
                stomp_noise_to_tile(nt, acc, lane_w);         // (the samples packed in the tile were consumed before barrier 2)
            }
            if (FUSED_NOISE_PRIO == STOMP_PRIO_STAGGER) __builtin_amdgcn_s_setprio(0);
        }
        // (paired draw, a batch to go: the next batch reads tiles another wave has just written)
        if (PAIRED && more_batches) __syncthreads();
        }   // batches
        float m_all = mb, z_all = zb, f_own = 1.f, f_own0 = 1.f, f_sd0 = 1.f, f_sd1 = 0.f;
        const unsigned tag = tag0 + (unsigned)it;
        float dsum = 0.f;
        FSTAMP(6);
        if (NB > 1) {
            // the two batches combined in batch order with the expressions of the exchange path below
            m_all = fmaxf(pm0, pm1);
            const float f0 = fast_expf(pm0 - m_all), f1 = fast_expf(pm1 - m_all);
            z_all = fmaf(f1, pz1, fmaf(f0, pz0, 0.f));
            f_own0 = f0;
            f_own = f1;
            f_sd0 = f0;
            f_sd1 = f1;
        }
        if (NB == 1 && nc > 1) {
            // every thread waits for ITS granules of every chunk (its own included: the very bits the partners read) --
            // no block barrier, no flag; combined in chunk order so that all partners compute bit-identical means
            float mk[FUSED_MAX_CHUNKS], zk[FUSED_MAX_CHUNKS], dk[FUSED_MAX_CHUNKS];
            const granule_t* slot0 = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc) * FUSED_XCHG;
            // the prefetched copy first (see "exchange prefetch" in the noise phase): the wave's own LDS-DMAs, so no barrier
            // (only in an iteration whose noise phase ran: otherwise the LDS copy is not this launch's)
            bool have = false;
            if (PRE && nc == 2 && it + 1 < n_run) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const granule_t* mine = sm.pre + wave * 128;
                const granule_t* mz = sm.premz + wave * 4;
                const granule_t g0 = mine[tq & 63], g1 = mine[64 + (tq & 63)];
                const granule_t gm0 = mz[0], gz0 = mz[1], gm1 = mz[2], gz1 = mz[3];
                have = (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag && (unsigned)(gm0 >> 32) == tag && (unsigned)(gz0 >> 32) == tag &&
                       (unsigned)(gm1 >> 32) == tag && (unsigned)(gz1 >> 32) == tag;
#pragma unroll
                for (int k = 2; k < FUSED_MAX_CHUNKS; ++k) { mk[k] = -3.0e38f; zk[k] = 0.f; dk[k] = 0.f; }
                mk[0] = __uint_as_float((unsigned)gm0); zk[0] = __uint_as_float((unsigned)gz0);
                mk[1] = __uint_as_float((unsigned)gm1); zk[1] = __uint_as_float((unsigned)gz1);
                dk[0] = (tq < N) ? __uint_as_float((unsigned)g0) : 0.f;
                dk[1] = (tq < N) ? __uint_as_float((unsigned)g1) : 0.f;
            }
#ifdef FUSED_T_PRE_STATS
            if (!have) ++pre_miss_;
#endif
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            // (measured and dropped, round 4: the first attempt's loads issued inside the noise phase, ahead of their use: the
            // twelve registers they hold there cost more than the round trip they hide -- 13.2 -> 13.4 us per iteration; issued
            // only just ahead of the noise tile's sixteen LDS stores they still tip 17 VGPRs of the kernel into scratch)
            for (;;) {
                if (have) break;
                bool ok = true;
                // (one chunk per call with a compile-time index: under the prefetch's branch the unroller gave up on the loop over
                // k and the partials were selected by compares -- two dependent round trips where this is one)
                auto poll_chunk = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    mk[k] = -3.0e38f; zk[k] = 0.f; dk[k] = 0.f;
                    if (k < nc) {
                        const granule_t* theirs = slot0 + (size_t)k * FUSED_XCHG;
                        const granule_t gm = ld_granule(theirs + 0), gz = ld_granule(theirs + 1);
                        const granule_t gd = ld_granule(theirs + 2 + (tq < N ? tq : 0));
                        ok = ok && (unsigned)(gm >> 32) == tag && (unsigned)(gz >> 32) == tag && (unsigned)(gd >> 32) == tag;
                        mk[k] = __uint_as_float((unsigned)gm);
                        zk[k] = __uint_as_float((unsigned)gz);
                        dk[k] = (tq < N) ? __uint_as_float((unsigned)gd) : 0.f;
                    }
                };
                static_assert(FUSED_MAX_CHUNKS == 4, "poll_chunk is called once per chunk");
                poll_chunk(std::integral_constant<int, 0>{});
                poll_chunk(std::integral_constant<int, 1>{});
                poll_chunk(std::integral_constant<int, 2>{});
                poll_chunk(std::integral_constant<int, 3>{});
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { s_abort = 1; break; }
            }
            FSTAMP(7);
            m_all = mk[0];
#pragma unroll
            for (int k = 1; k < FUSED_MAX_CHUNKS; ++k) m_all = fmaxf(m_all, mk[k]);
            z_all = 0.f;
            dsum = 0.f;
#pragma unroll
            for (int k = 0; k < FUSED_MAX_CHUNKS; ++k) {
                if (k < nc) {
                    const float f = fast_expf(mk[k] - m_all);
                    z_all = fmaf(f, zk[k], z_all);
                    dsum = fmaf(f, dk[k], dsum);
                }
            }
            f_own = fast_expf(mb - m_all);
        }
        FSTAMP(8);
        // ============ E. weights out; mean += lr * (sum_k f_k Sigma D_k) / z   (z >= 1: it holds the term exp(0) of the maximum;
        //              one v_rcp_f32 instead of an IEEE division per use, round 5)
        const float rz = fast_rcpf(z_all);
        if (NB > 1) {
            const int sl0 = tq & 15;
            if (tq < FUSED_WAVES && sl0 < S) weights[(size_t)p * S + sl0] = pe0 * f_own0 * rz;
            if (tq < FUSED_WAVES && sl < S) weights[(size_t)p * S + sl] = pe1 * f_own * rz;
        } else if (tq < FUSED_WAVES && sl < S) weights[(size_t)p * S + sl] = ex * f_own * rz;
        if (NB == 1 && nc > 1) {
            if (tq < N) mean_l[tq] += lr * (dsum * rz);
        } else if (wave < 4) {
            // (the products are still in the registers of waves 0-3: row 16 wave + 4 lg + r, channel li)
            const int li = tq & 15, lg = (tq >> 4) & 3;
            if (li < DCH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float ds = fmaf(f_sd0, sd0[r], 0.f);
                    if (NB > 1) ds = fmaf(f_sd1, sd1[r], ds);
                    mean_l[(16 * wave + 4 * lg + r) * DCH + li] += lr * (ds * rz);
                }
            }
        }
        FSTAMP(9);
        FSTAMP(10);
        __syncthreads();                                                                        // (5) new mean visible, tiles free
        FSTAMP(11);
        if (s_abort) break;                                                                     // block-uniform (set before barrier 5)
        if (it < 3) LSTAMP(4 + it);
    }
#ifdef FUSED_T_PRE_STATS
    if (pre_miss_ > 1) atomicAdd(&g_pre_miss[wave], (unsigned)(pre_miss_ - 1));     // (the last iteration has no prefetch)
#endif
    const int aborted = s_abort;                        // (block-uniform: last written before a barrier every thread passed)
    if (!aborted && chunk == 0 && tid < N) {
        const float m = mean_l[tid];
        means[(size_t)p * N + tid] = m;
        if (means_copy) means_copy[(size_t)p * N + tid] = m;      // the caller's own copy (OptimizationPlanner._get_traj clones)
    }
    // a lost call leaves the means as they were: the caller's copy says the same (never uninitialised memory)
    if (aborted == 1 && chunk == 0 && tid < N && means_copy) means_copy[(size_t)p * N + tid] = means[(size_t)p * N + tid];
    // (header not zeroed: no unit was drawn -- the first P workgroups copy one particle each)
    if (aborted == 2 && (int)blockIdx.x < P && tid < N && means_copy) means_copy[(size_t)blockIdx.x * N + tid] = means[(size_t)blockIdx.x * N + tid];
    // ---- leaving: the error word (device header + the caller's host-visible status block), then the head count; the
    //      last workgroup out re-arms the header for the next call and reports the call as completed
    //      (with check='sync' the host returns as soon as the status block holds the tag, so every wave's result stores
    //      must be ordered before thread 0's release: each wave waits until its own stores have been acknowledged by the L2
    //      (s_waitcnt vmcnt(0): gfx9 counts stores there), then the barrier; thread 0's agent-scope release in fused_leave writes
    //      the L2 back.  NOT __threadfence(): an agent-scope fence in every wave is an L2 write-back per wave -- measured +63 us
    //      per launch at C3 (launch_fixed_ms 0.019 -> 0.082), k1 25 k -> 10 k it/s)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        LSTAMP(7);
        fused_leave(wsu, status_host, tag0, aborted);
    }
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
static int device_cu_count() {       // (every GPU of a node is the same part: asked once)
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return n_cu;
}

// the generalised kernel (mpb_stomp_fused_hx.hip): any H <= 128, d <= 16, S <= 128
bool mpb_fused_hx_plan(int geom_flags, int n_cu, int P, int S, int H, int d, int* nc_out, int* nb_out, size_t* ws_bytes);
int mpb_fused_hx_launch(float* means, const float* eps, float* samples, float* costs, float* weights, const float* L,
                        const float* Sigma, const float* geom, int geom_flags, float* workspace, int P, int S, int H, int d, int nc,
                        int nb, float k_sigma, float weight, float lr, float temperature, int n_iters, uint32_t lo, uint32_t hi,
                        uint32_t iter0, uint32_t particle_offset, uint32_t tag0, unsigned long long timeout, unsigned* status_dev,
                        float* means_copy, hipStream_t st, const FusedProfile* prof);

// which form of the loop serves a call, and the workspace it needs
struct FusedPlan {
    int path;            // MPB_STOMP_PATH_*: 0 two-kernel loop, 1 persistent with exchange, 2 persistent one workgroup per particle
    int nc;              // workgroups per particle (exchange layout)
    bool two_batches;    // (H = 64 kernel) one workgroup per particle runs two batches of 16 samples
    bool hx;             // served by the generalised kernel (any H <= 128, d <= 16, S <= 128)
    int nb;              // (generalised kernel) passes per workgroup and iteration
    size_t ws_bytes;     // workspace the persistent kernel needs (header only when nothing is exchanged)
};
static FusedPlan fused_plan(int geom_flags, int P, int S, int H, int d) {
    FusedPlan f = {MPB_STOMP_PATH_TWO_KERNEL, 1, false, false, 1, 0};
    if (P < 1 || S < 1) return f;
    const int n_cu = device_cu_count();
    // MPB_STOMP_HX = 1 sends every shape to the generalised kernel (a test aid: it is compared with the H = 64 kernel)
    const char* hx_env = getenv("MPB_STOMP_HX");
    const int force_hx = hx_env ? atoi(hx_env) : 0;
    const bool v1 = !force_hx && H == 64 && S <= FUSED_WAVES * FUSED_MAX_CHUNKS && (geom_flags & 0x100) &&
                    (d == 2 || d == 3 || d == 4 || d == 6 || d == 7 || d == 14);
    if (!v1) {
        if (!mpb_fused_hx_plan(geom_flags, n_cu, P, S, H, d, &f.nc, &f.nb, &f.ws_bytes)) return f;
        f.hx = true;
        f.path = f.nc > 1 ? MPB_STOMP_PATH_PERSISTENT_EXCHANGE : MPB_STOMP_PATH_PERSISTENT;
        return f;
    }
    f.nc = (S + FUSED_WAVES - 1) / FUSED_WAVES;
    // layout: one workgroup per (particle, chunk of 16 samples) with the exchange -- or, when there are at least as many
    // particles as CUs and S <= 32, one workgroup per particle running two batches of 16 (no exchange; same bits).
    // MPB_STOMP_BATCHES = 1 / 2 forces one or the other (2 only where it applies).
    static const int force_nb = [] { const char* e = getenv("MPB_STOMP_BATCHES"); return e ? atoi(e) : 0; }();
    // rounds of workgroups either layout needs on this chip: the two-batch workgroup takes ~1.88 x as long per iteration
    const long r1 = (2L * P + n_cu - 1) / n_cu, r2 = ((long)P + n_cu - 1) / n_cu;
    f.two_batches = f.nc == 2 && force_nb != 1 && (force_nb == 2 || 188 * r2 < 100 * r1);
    const bool exchange = f.nc > 1 && !f.two_batches;
    f.path = exchange ? MPB_STOMP_PATH_PERSISTENT_EXCHANGE : MPB_STOMP_PATH_PERSISTENT;
    f.ws_bytes = (exchange ? fused_ws_floats(P, f.nc) : FUSED_HDR_WORDS) * sizeof(float);
    return f;
}

extern "C" size_t mpb_stomp_workspace_bytes(int P, int S, int H, int d) {
    if (P < 1 || S < 1) return 0;
    // what the layout the launcher will pick needs (grid-backed fields assumed; a call the persistent kernel cannot
    // serve needs none): the exchange area only when partner workgroups exchange partials, else just the header
    const FusedPlan f = fused_plan(0x100, P, S, H, d);
    // (a scene packed with LIST grids -- geometry version 7, flag bit 13 -- goes to the generalised kernel even at H = 64, whose
    // exchange slots are larger: the workspace serves whichever of the two the geometry will select)
    const FusedPlan fl = fused_plan(0x2000, P, S, H, d);
    size_t b = f.path == MPB_STOMP_PATH_TWO_KERNEL ? FUSED_HDR_WORDS * sizeof(float) : f.ws_bytes;
    if (fl.path != MPB_STOMP_PATH_TWO_KERNEL && fl.ws_bytes > b) b = fl.ws_bytes;
    return b;
}

extern "C" int mpb_stomp_workspace_init(float* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < FUSED_HDR_WORDS * sizeof(float)) return mpb_fail(MPB_E_INVALID, "mpb_stomp_workspace_init: workspace too small");
    if (hipMemsetAsync(workspace, 0, FUSED_HDR_WORDS * sizeof(float), (hipStream_t)stream) != hipSuccess) return mpb_fail(MPB_E_HIP, "mpb_stomp_workspace_init: memset failed");
    return MPB_OK;
}

extern "C" int mpb_stomp_run_path(int geom_flags, size_t workspace_bytes, int P, int S, int H, int d) {
    const FusedPlan f = fused_plan(geom_flags, P, S, H, d);
    return (f.path != MPB_STOMP_PATH_TWO_KERNEL && workspace_bytes >= f.ws_bytes) ? f.path : MPB_STOMP_PATH_TWO_KERNEL;
}

extern "C" int mpb_stomp_step(float* means, const float* eps, float* samples, float* costs, float* weights,
                              const float* L, const float* Sigma, const float* geom, int geom_flags, int P, int S, int H, int d, int D,
                              float k_sigma, float weight, float lr, float temperature, int n_iters, uint64_t seed,
                              uint32_t iter0, uint32_t particle_offset, void* stream);

static thread_local const FusedProfile* t_prof = nullptr;      // set by mpb_stomp_run_timed around its launch

extern "C" int mpb_stomp_run_checked(float* means, const float* eps, float* samples, float* costs, float* weights,
                                     const float* L, const float* Sigma, const float* geom, int geom_flags, float* workspace,
                                     size_t workspace_bytes, int P, int S, int H, int d, int D, float k_sigma, float weight, float lr,
                                     float temperature, int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset,
                                     uint32_t* status, uint32_t* tag_out, float* means_copy, void* stream) {
    if (tag_out) *tag_out = 0u;
    if (P == 0) return MPB_OK;
    const FusedPlan f = fused_plan(geom_flags, P, S, H, d);
    if (n_iters == 0 || !workspace || f.path == MPB_STOMP_PATH_TWO_KERNEL || workspace_bytes < f.ws_bytes) {
        const int rc = n_iters == 0 ? MPB_OK : mpb_stomp_step(means, eps, samples, costs, weights, L, Sigma, geom, geom_flags, P, S, H, d, D,
                                                             k_sigma, weight, lr, temperature, n_iters, seed, iter0, particle_offset, stream);
        if (rc == MPB_OK && means_copy && means &&
            hipMemcpyAsync(means_copy, means, sizeof(float) * (size_t)P * H * d, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return mpb_fail(MPB_E_HIP, "mpb_stomp_run: copy of the means failed");
        return rc;
    }
    if (!means || !samples || !costs || !weights || !L || !Sigma || !geom) return mpb_fail(MPB_E_INVALID, "mpb_stomp_run: null pointer");
    if (P < 0 || S < 1 || n_iters < 0 || !(d == D || d == 2 * D)) return mpb_fail(MPB_E_INVALID, "mpb_stomp_run: bad shape");
    if (!(temperature > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_stomp_run: temperature must be > 0");
    if (mpb_misaligned16(means, eps, samples, L, Sigma, geom) || mpb_misaligned16(workspace, means_copy))
        return mpb_fail(MPB_E_INVALID, "mpb_stomp_run: means / eps / samples / L / Sigma / geom / workspace / means_copy must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    // the status block is host memory the device can write (pinned + mapped): its device address
    unsigned* status_dev = nullptr;
    if (status) {
        static thread_local uint32_t* seen_host = nullptr;      // (a planner passes the same block every call: asked once)
        static thread_local unsigned* seen_dev = nullptr;
        if (status != seen_host) {
            unsigned* dp = nullptr;
            if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dp), status, 0) != hipSuccess) {
                (void)hipGetLastError();
                return mpb_fail(MPB_E_INVALID, "mpb_stomp_run: status is not pinned, device-mapped host memory");
            }
            seen_host = status;
            seen_dev = dp;
        }
        status_dev = seen_dev;
    }
    // the granules' tags and the error word carry a per-call epoch (process-wide counter scrambled over 32 bits), so
    // whatever an earlier call left in the exchange area does not match.  Header: word 0 = tag of the call in which a
    // workgroup gave up, word 1 = tag of the last call; "lost" <=> word 0 == word 1 != 0.  Not capturable in a HIP
    // graph: a replay would reuse the tag.
    static std::atomic<uint32_t> epoch{(uint32_t)std::chrono::steady_clock::now().time_since_epoch().count()};
    uint32_t tag0 = (epoch.fetch_add(1u) + 1u) * 0x9E3779B9u;
    if (tag0 == 0u) tag0 = 0x9E3779B9u;      // 0 means "none" in the header and the status block
    if (tag_out) *tag_out = tag0;
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    // bound of every wait for a partner; MPB_STOMP_TIMEOUT_US overrides it (a test aid)
    unsigned long long timeout = FUSED_TIMEOUT_TICKS + FUSED_TIMEOUT_PER_ITER * (unsigned long long)n_iters;
    if (const char* e = getenv("MPB_STOMP_TIMEOUT_US")) { const long long us = atoll(e); if (us > 0) timeout = 100ull * (unsigned long long)us; }
    if (f.hx)
        return mpb_fused_hx_launch(means, eps, samples, costs, weights, L, Sigma, geom, geom_flags, workspace, P, S, H, d, f.nc, f.nb,
                                   k_sigma, weight, lr, temperature, n_iters, lo, hi, iter0, particle_offset, tag0, timeout, status_dev,
                                   means_copy, st, t_prof);
    const dim3 grid(f.two_batches ? P : P * f.nc), block(FUSED_THREADS);
    const int nc_k = f.two_batches ? 1 : f.nc;
    const int model = geom_flags & 0xFF;
#define MPB_F_LAUNCH_(DCH, MODEL, NB, INJ, CHAIN)                                                                                      \
    MPB_FUSED_LAUNCH(t_prof, (stomp_fused_kernel<DCH, MODEL, NB, INJ, CHAIN>), grid, block, st, means, eps, samples, costs, weights, L, \
                     Sigma, geom, workspace, P, S, nc_k, k_sigma, weight, lr, temperature, n_iters, lo, hi, iter0,                      \
                     particle_offset, tag0, timeout, status_dev, means_copy)
    const bool one_field = (geom_flags & 0x1000) != 0;
#define MPB_F_LAUNCH(DCH, MODEL, NB, INJ)                                              \
    do {                                                                               \
        if ((MODEL) != 0 && one_field) MPB_F_LAUNCH_(DCH, MODEL, NB, INJ, (MODEL) == 0); \
        else MPB_F_LAUNCH_(DCH, MODEL, NB, INJ, true);                                 \
    } while (0)
#define MPB_F_CASE(DCH, MODEL)                                             \
    do {                                                                   \
        if (f.two_batches && eps) MPB_F_LAUNCH(DCH, MODEL, 2, true);       \
        else if (f.two_batches) MPB_F_LAUNCH(DCH, MODEL, 2, false);        \
        else if (eps) MPB_F_LAUNCH(DCH, MODEL, 1, true);                   \
        else MPB_F_LAUNCH(DCH, MODEL, 1, false);                           \
    } while (0)
    if (model == PandaModel::ID && d == 7) MPB_F_CASE(7, PandaModel::ID);
    else if (model == PandaModel::ID && d == 14) MPB_F_CASE(14, PandaModel::ID);
    else if (d == 2) MPB_F_CASE(2, 0);
    else if (d == 3) MPB_F_CASE(3, 0);
    else if (d == 4) MPB_F_CASE(4, 0);
    else if (d == 6) MPB_F_CASE(6, 0);
    else if (d == 7) MPB_F_CASE(7, 0);
    else MPB_F_CASE(14, 0);
#undef MPB_F_CASE
#undef MPB_F_LAUNCH
#undef MPB_F_LAUNCH_
    return mpb_check_launch("mpb_stomp_run");
}

extern "C" int mpb_stomp_run(float* means, const float* eps, float* samples, float* costs, float* weights,
                             const float* L, const float* Sigma, const float* geom, int geom_flags, float* workspace,
                             size_t workspace_bytes, int P, int S, int H, int d, int D, float k_sigma, float weight, float lr,
                             float temperature, int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset,
                             void* stream) {
    return mpb_stomp_run_checked(means, eps, samples, costs, weights, L, Sigma, geom, geom_flags, workspace, workspace_bytes, P, S,
                                 H, d, D, k_sigma, weight, lr, temperature, n_iters, seed, iter0, particle_offset, nullptr,
                                 nullptr, nullptr, stream);
}

/* A call of mpb_stomp_run_checked with everything but (n_iters, iter0, means_copy, stream) fixed, kept on the library's side:
   a planner whose buffers do not change between optimize() calls hands over four values per call instead of twenty-eight
   (the foreign-function marshalling of the long form is ~3 us of the ~11 us host side of a call).  Device noise only (eps = NULL). */
struct mpb_stomp_plan_s {
    float *means, *samples, *costs, *weights;
    const float *L, *Sigma, *geom;
    int geom_flags;
    float* workspace;
    size_t workspace_bytes;
    int P, S, H, d, D;
    float k_sigma, weight, lr, temperature;
    uint64_t seed;
    uint32_t particle_offset;
    uint32_t* status;
};

extern "C" int mpb_stomp_plan_create(mpb_stomp_plan** plan, float* means, float* samples, float* costs, float* weights, const float* L,
                                     const float* Sigma, const float* geom, int geom_flags, float* workspace, size_t workspace_bytes,
                                     int P, int S, int H, int d, int D, float k_sigma, float weight, float lr, float temperature,
                                     uint64_t seed, uint32_t particle_offset, uint32_t* status) {
    if (!plan) return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_create: null pointer");
    *plan = nullptr;
    if (!means || !samples || !costs || !weights || !L || !Sigma || !geom) return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_create: null pointer");
    if (P < 0 || S < 1 || !(d == D || d == 2 * D)) return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_create: bad shape");
    if (!(temperature > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_create: temperature must be > 0");
    if (mpb_misaligned16(means, samples, L, Sigma, geom, workspace))
        return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_create: means / samples / L / Sigma / geom / workspace must be 16-byte aligned");
    mpb_stomp_plan_s* q = static_cast<mpb_stomp_plan_s*>(malloc(sizeof(mpb_stomp_plan_s)));
    if (!q) return mpb_fail(MPB_E_HIP, "mpb_stomp_plan_create: out of host memory");
    *q = mpb_stomp_plan_s{means, samples, costs, weights, L, Sigma, geom, geom_flags, workspace, workspace_bytes, P, S, H, d, D,
                          k_sigma, weight, lr, temperature, seed, particle_offset, status};
    *plan = q;
    return MPB_OK;
}

extern "C" int mpb_stomp_plan_launch(mpb_stomp_plan* plan, int n_iters, uint32_t iter0, float* means_copy, void* stream, uint32_t* tag_out) {
    if (!plan) return mpb_fail(MPB_E_INVALID, "mpb_stomp_plan_launch: null plan");
    const mpb_stomp_plan_s& q = *plan;
    return mpb_stomp_run_checked(q.means, nullptr, q.samples, q.costs, q.weights, q.L, q.Sigma, q.geom, q.geom_flags, q.workspace,
                                 q.workspace_bytes, q.P, q.S, q.H, q.d, q.D, q.k_sigma, q.weight, q.lr, q.temperature, n_iters, q.seed,
                                 iter0, q.particle_offset, q.status, tag_out, means_copy, stream);
}

extern "C" int mpb_stomp_plan_destroy(mpb_stomp_plan* plan) {
    free(plan);
    return MPB_OK;
}

/* measurement aid for bench.py: mpb_stomp_run_checked with the kernel's begin / end timestamps recorded on the dispatch
   itself; synchronises the stream and returns the kernel's duration (0 when the call ran the two-kernel loop) */
extern "C" int mpb_stomp_run_timed(float* means, const float* eps, float* samples, float* costs, float* weights,
                                   const float* L, const float* Sigma, const float* geom, int geom_flags, float* workspace,
                                   size_t workspace_bytes, int P, int S, int H, int d, int D, float k_sigma, float weight, float lr,
                                   float temperature, int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset,
                                   uint32_t* status, uint32_t* tag_out, float* means_copy, void* stream, float* kernel_ms) {
    if (!kernel_ms) return mpb_fail(MPB_E_INVALID, "mpb_stomp_run_timed: null pointer");
    *kernel_ms = 0.f;
    FusedProfile pr = {nullptr, nullptr};
    if (hipEventCreate(&pr.start) != hipSuccess || hipEventCreate(&pr.stop) != hipSuccess) {
        if (pr.start) (void)hipEventDestroy(pr.start);
        return mpb_fail(MPB_E_HIP, "mpb_stomp_run_timed: hipEventCreate failed");
    }
    uint32_t tag = 0;
    t_prof = &pr;
    int rc = mpb_stomp_run_checked(means, eps, samples, costs, weights, L, Sigma, geom, geom_flags, workspace, workspace_bytes, P, S, H,
                                   d, D, k_sigma, weight, lr, temperature, n_iters, seed, iter0, particle_offset, status, &tag,
                                   means_copy, stream);
    t_prof = nullptr;
    if (tag_out) *tag_out = tag;
    if (rc == MPB_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = mpb_fail(MPB_E_HIP, "mpb_stomp_run_timed: synchronize failed");
    if (rc == MPB_OK && tag != 0u && hipEventElapsedTime(kernel_ms, pr.start, pr.stop) != hipSuccess) {
        (void)hipGetLastError();
        rc = mpb_fail(MPB_E_HIP, "mpb_stomp_run_timed: hipEventElapsedTime failed");
    }
    (void)hipEventDestroy(pr.start);
    (void)hipEventDestroy(pr.stop);
    return rc;
}

/* state of the last persistent launch on this workspace (host-side read of the header: synchronises the stream):
   0 = fine (or no persistent launch yet), 1 = a workgroup gave up waiting for its partner, 2 = header not initialised */
extern "C" int mpb_stomp_run_status(const float* workspace, void* stream, int* timed_out) {
    if (!workspace || !timed_out) return mpb_fail(MPB_E_INVALID, "mpb_stomp_run_status: null pointer");
    uint32_t w[4] = {0u};
    if (hipMemcpyAsync(w, workspace, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return mpb_fail(MPB_E_HIP, "mpb_stomp_run_status: copy failed");
    *timed_out = (w[FUSED_HDR_ERR] == w[FUSED_HDR_TAG] && w[FUSED_HDR_TAG] != 0u) ? (w[FUSED_HDR_WHY] == 2u ? 2 : 1) : 0;
    return MPB_OK;
}
