// mpb_stomp_fused_hx.hip -- the persistent one-launch STOMP loop (stomp.py:150-160) for the shapes the H = 64 kernel of
// mpb_stomp_fused.hip does not serve: any horizon up to 128 support points, any channel count d <= 16, any number of
// samples per particle up to 128.  Same idea -- a workgroup of 16 waves owns (particle, chunk of its samples) for all
// iterations, constants / means / the iteration's samples stay in LDS, the workgroups of a particle exchange per-chunk
// softmax partials as tagged granules, paired by start-order tickets (mpb_stomp_fused.h) -- generalised along three axes:
//   * horizon: a rollout is cut in HC = ceil(H / 64) chunks of 64 waypoints and a WAVE owns one chunk of one rollout
//     (lane = waypoint of the chunk), so every per-wave phase is that of the H = 64 kernel; the noise of chunk hc is
//     sum_{kc <= hc} L[hc][kc] eps[kc], block by block on the matrix cores (the blocks of the lower-triangular scale_tril
//     sit in LDS as permuted images), and a wave of chunk hc draws eps[0..hc] itself (counter-based: no exchange);
//     chunks are dealt to the waves so that every SIMD holds as many first as second chunks;
//   * samples: a workgroup runs its RB = 16 / HC rollouts per pass, `nb` passes ("batches") per iteration, and folds each
//     batch into a RUNNING softmax partial (m, z, sum e (x - mu)) -- the streaming form of the same algebra, any nb;
//   * channels: d is a run-time value (DCH = 0) or a template constant (the Panda's 7 / 14).
// The noise product is issued TRANSPOSED (A = eps, B = L^T image: D[channel][waypoint]) and brought to lane = waypoint
// with v_permlane32_swap / v_permlane16_swap (a 4 x 4 block transpose across the four 16-lane rows) -- no LDS round trip;
// the update  mean += lr * Sigma @ delta  runs on the matrix cores too (Sigma rows streamed from L2: a 128 x 128 Sigma
// does not fit in LDS next to the tiles).  Results equal the two-kernel path (mpb_stomp_step) to rounding: samples and costs
// of a 64-waypoint horizon bit for bit, weights / means within 1e-6 (tests/test_gpu_stomp_fused_hx.py).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <type_traits>

#include "mpb_common.h"
#include "mpb_geom.h"
#include "mpb_stomp_noise.h"
#include "mpb_stomp_fused.h"

#define HX_XCHG 2064                       // granules per published partial: m, z, then H*d <= 2048 values, padded
#define HX_MAX_NB 8

static inline size_t hx_ws_floats(int P, int nc) { return FUSED_HDR_WORDS + 2 * 2 * (size_t)P * nc * HX_XCHG; }

// acc[n] (lane (j, g): D[channel 4g + rr][waypoint 16 n + j]) -> nz[c] of THIS lane's waypoint (lane = 16 g + j): a 4 x 4
// block transpose over the four 16-lane rows, per accumulator register rr
__device__ __forceinline__ void hx_transpose_to_rows(const f32x4 (&acc)[4], float (&nz)[16]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        unsigned x0 = __float_as_uint(acc[0][rr]), x1 = __float_as_uint(acc[1][rr]), x2 = __float_as_uint(acc[2][rr]),
                 x3 = __float_as_uint(acc[3][rr]);
        // stage 1: rows {2,3} of x0 / x1 <-> rows {0,1} of x2 / x3
        auto a = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
        // stage 2: odd rows of the first <-> even rows of the second
        auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
        auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
        nz[0 + rr] = __uint_as_float(lo[0]);
        nz[4 + rr] = __uint_as_float(lo[1]);
        nz[8 + rr] = __uint_as_float(hi[0]);
        nz[12 + rr] = __uint_as_float(hi[1]);
    }
}

// INJ = false: the instantiation of the device-noise calls does not carry the injected-noise path (as in mpb_stomp_fused.hip)
// LIST = true (round 6, H <= 64 only: the second horizon chunk leaves no LDS for it): the fields carry LIST grids (geometry version 7:
// scenes beyond the compact grid's 63 spheres, boxes culled like spheres; mpb_geom.h, spheres_hinge_list) -- cell words, up to 16 KB of
// candidate bytes, the sphere table (255) and the box table (127) in LDS, +24 KB.
template <int DCH, int MODEL, int HC, bool INJ, bool LIST = false>
__global__ __launch_bounds__(FUSED_THREADS, 4) void stomp_fused_hx_kernel(
    float* __restrict__ means, const float* __restrict__ eps, float* __restrict__ samples, float* __restrict__ costs,
    float* __restrict__ weights, const float* __restrict__ Lmat, const float* __restrict__ Sigma,
    const float* __restrict__ geom, float* __restrict__ ws, int P, int S, int H, int d_rt, int nc, int nb, float k_sigma,
    float weight, float lr, float temperature, int n_iters, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter0,
    uint32_t particle_offset, uint32_t tag0, unsigned long long timeout_ticks, unsigned* __restrict__ status_host,
    float* __restrict__ means_copy) {
    if (!INJ) eps = nullptr;
    constexpr int DX = DCH ? DCH : 16;                 // channels held in registers / tile rows
    constexpr int RB = FUSED_WAVES / HC;               // rollouts per pass
    constexpr int HP = 64 * HC;                        // padded horizon
    constexpr int NLB = HC * (HC + 1) / 2;             // 64 x 64 blocks of the lower triangle of L
    constexpr int EPT = (HP * DX + FUSED_THREADS - 1) / FUSED_THREADS;   // trajectory elements per thread
    constexpr int DLD = HP + 4;                        // row (floats) of the transposed delta tile
    constexpr int TILE = 64 * DX;                      // floats of a wave's tile: its chunk of one rollout, rows packed
    static_assert(HC == 1 || HC == 2, "horizons up to 128 support points");
    static_assert(!LIST || HC == 1, "the list grid's tables only fit next to ONE horizon chunk");
    static_assert(HP * DX + 2 <= HX_XCHG, "exchange slot too small");
    // the 64 x 64 blocks of the lower triangle of L as three-component bf16 MFMA images (mpb_stomp_noise.h): block 0 = (0,0)
    // and block 2 = (1,1) are diagonal blocks (6 tiles, 18 KB), block 1 = (1,0) is full (8 tiles, 24 KB)
    constexpr int LIMG_WORDS = (HC == 1) ? STOMP_LIMG_WORDS : 2 * STOMP_LIMG_WORDS + STOMP_LIMG_WORDS_FULL;
    // (one shared object, layout fixed by hand: the obstacle table, the grid and the small arrays first -- inside the reach of an
    // LDS instruction's 16-bit offset field --, the L images and the sample tiles behind them; mpb_stomp_fused.hip)
    struct Smem {
        float4 otab[(LIST ? MPB_LIST_MAX_SPH : MPB_GRID_MAX_SPH) + 1];
        unsigned gridw[MPB_GRID_MAX_CELLS];
        float4 btab[LIST ? 2 * (MPB_LIST_MAX_BOX + 1) : 1];
        unsigned char cand[LIST ? MPB_LIST_MAX_CAND + 16 : 16];
        unsigned Limg[LIMG_WORDS];
        float mean_l[HP * DX];
        float delta[16 * DLD];
        float sig_l[HC == 1 ? 64 * DLD : 4];           // H <= 64: Sigma stays in LDS (padded rows)
        float cst[FUSED_WAVES];
        float ewl[HX_MAX_NB * FUSED_WAVES];            // exp(logit - batch max) of the unit's samples
        float mbl[HX_MAX_NB];                          // the batches' maxima
        int s_abort;
        unsigned s_ticket;
        unsigned pad_[2];
        float tiles[FUSED_WAVES * TILE];
    };
    static_assert((HX_MAX_NB * FUSED_WAVES + HX_MAX_NB) % 4 == 0, "Limg must stay 16-byte aligned");
    __shared__ __attribute__((aligned(16))) Smem sm;
    float4 (&otab)[(LIST ? MPB_LIST_MAX_SPH : MPB_GRID_MAX_SPH) + 1] = sm.otab;
    unsigned (&gridw)[MPB_GRID_MAX_CELLS] = sm.gridw;
    const ListView LV = {sm.gridw, sm.cand, sm.otab, sm.btab};
    // (a field's grid into LDS: offset words of the compact grid, or the list grid's four tables)
    auto stage_field = [&](const GeomView& Gf, int t_) {
        if constexpr (LIST) {
            if (list_usable(Gf)) list_stage(Gf, sm.gridw, sm.cand, sm.otab, sm.btab, t_, FUSED_THREADS);
        } else {
            grid_stage_offsets(Gf, gridw, otab, t_, FUSED_THREADS);
        }
    };
    float (&mean_l)[HP * DX] = sm.mean_l;
    float (&delta)[16 * DLD] = sm.delta;
    float (&sig_l)[HC == 1 ? 64 * DLD : 4] = sm.sig_l;
    float (&cst)[FUSED_WAVES] = sm.cst;
    float (&ewl)[HX_MAX_NB * FUSED_WAVES] = sm.ewl;
    float (&mbl)[HX_MAX_NB] = sm.mbl;
    int& s_abort = sm.s_abort;
    unsigned& s_ticket = sm.s_ticket;
    unsigned (&Limg)[LIMG_WORDS] = sm.Limg;
    float (&tiles)[FUSED_WAVES * TILE] = sm.tiles;

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;                                   // (re-derived per pass from an opaque copy: see HX_FRESH_LANE)
    const int d = DCH ? DCH : d_rt;
    const int N = H * d;
    unsigned* wsu = reinterpret_cast<unsigned*>(ws);
    granule_t* xch = reinterpret_cast<granule_t*>(ws + FUSED_HDR_WORDS);
    const bool exchange = nc > 1;
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
    // this wave's rollout slot and horizon chunk: with two chunks the waves of a SIMD (w, w + 4, w + 8, w + 12) hold two of each
    const int hcw = (HC == 1) ? 0 : ((wave >> 2) & 1);
    const int rw = (HC == 1) ? wave : ((wave & 3) | ((wave >> 3) << 2));
    auto wave_of = [](int r, int hc) { return (HC == 1) ? r : ((r & 3) | (hc << 2) | ((r >> 2) << 3)); };
    int j = lane & 15, g = lane >> 4;
    // per-lane index arithmetic (tile rows, operand addresses, masks) is loop invariant, and the compiler hoists ALL of it out
    // of the iteration loop: ~30 VGPRs that do not survive the cost phase and came back as 65 scratch reloads per iteration
    // (124 B / lane).  Re-deriving lane / j / g from an opaque copy at the top of every pass keeps them local to it.
#define HX_FRESH_LANE()                          \
    do {                                         \
        asm volatile("" : "+v"(lane));           \
        j = lane & 15;                           \
        g = lane >> 4;                           \
    } while (0)

    // ---- constants into LDS (once): the broad-phase grid + obstacle table, the blocks of L as permuted MFMA images
    GeomView G0 = geom_view(geom);
    stage_field(G0, tid);                                         // (compact grid: as offset words, mpb_geom.h grid_offset_word)
#pragma unroll
    for (int b = 0; b < NLB; ++b) {
        const int hc = (b == 0) ? 0 : 1, kc = (b == 2) ? 1 : 0;
        // one thread per four consecutive columns of a row (zero past H)
        const int row = tid >> 4, col0 = (tid & 15) << 2;
        const int gr = 64 * hc + row, gc = 64 * kc + col0;
        f32x4 lv;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) lv[e4] = (gr > 0 && gr < H - 1 && gc + e4 < H) ? Lmat[(size_t)gr * H + gc + e4] : 0.f;   // (rows 0 / H - 1 of the image are zero: stomp.py:105-106, as in mpb_stomp_fused.hip)
        if (b == 1) stomp_l_image_store<true>(Limg + STOMP_LIMG_WORDS, row, col0, lv);
        else stomp_l_image_store<false>(Limg + (b == 0 ? 0 : STOMP_LIMG_WORDS + STOMP_LIMG_WORDS_FULL), row, col0, lv);
    }
    for (int i = tid; i < 16 * DLD; i += FUSED_THREADS) delta[i] = 0.f;        // (padding rows / columns stay zero)
    if (HC == 1) {
        for (int v = tid; v < 64 * 64; v += FUSED_THREADS) {
            const int row = v >> 6, col = v & 63;
            sig_l[row * DLD + col] = (row < H && col < H) ? Sigma[(size_t)row * H + col] : 0.f;
        }
    }
    __syncthreads();
    const int unit = __builtin_amdgcn_readfirstlane((int)s_ticket);
    const int p = exchange ? unit / nc : unit;
    const int chunk = exchange ? unit - p * nc : 0;
    for (int e = tid; e < N; e += FUSED_THREADS) mean_l[e] = means[(size_t)p * N + e];
    if (unit == 0 && tid == 0) st_agent_u(wsu + FUSED_HDR_TAG, tag0);
    __syncthreads();
    const int n_run = s_abort ? 0 : n_iters;
    const float inv_temperature = 1.0f / temperature;

    float* nt = tiles + wave * TILE;
    const size_t eps_stride = (size_t)S * d * P * H;

    // the noise rows of (iteration it_n, sample s_n) for this wave's chunk, parked in the wave's tile (rows packed, stride d).
    // Block-uniform call sites only (two chunks: one workgroup barrier inside).
    auto draw_noise = [&](int it_n, int s_n) {
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        int jv = j, gv = g;
        asm volatile("" : "+v"(jv), "+v"(gv));
        uint32_t slo = seed_lo, shi = seed_hi;     // (opaque: keeps the round keys from being hoisted out of the loop and spilled)
        asm volatile("" : "+s"(slo), "+s"(shi));
        // the eps columns of chunk kc as the values of this lane: v[32 KB' + e] = eps[c = j][k = 64 kc + 32 KB' + 8 g + e]
        // (device noise: Philox calls 2 KB', 2 KB' + 1 of the chunk -- mpb_stomp_noise.h, stomp_eps_column)
        auto operand = [&](int kc, float (&e)[16]) {
            if (eps != nullptr) {
                const float* ep = eps + (size_t)it_n * eps_stride + (((size_t)(s_n < S ? s_n : 0) * d + (jv < d ? jv : 0)) * P + p) * H;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = 64 * kc + 32 * (u >> 3) + 8 * gv + (u & 7);
                    e[u] = (jv < d && k < H) ? ep[k] : 0.f;
                }
            } else {
                float lo8[8], hi8[8];
                uint32_t carry[2] = {0u, 0u};
#pragma unroll
                for (int q = 0; q < 8; ++q) lo8[q] = hi8[q] = 0.f;
                if (jv < d) {
                    stomp_normals_lo<STOMP_PRIO_PROGRESS>(particle_offset + (uint32_t)p, (uint32_t)s_n, (uint32_t)jv, (uint32_t)gv,
                                                          (uint32_t)(kc << 4), iter0 + (uint32_t)it_n, slo, shi, lo8, carry);
                    stomp_normals_hi<STOMP_PRIO_PROGRESS>(particle_offset + (uint32_t)p, (uint32_t)s_n, (uint32_t)jv, (uint32_t)gv,
                                                          (uint32_t)(kc << 4), iter0 + (uint32_t)it_n, slo, shi, carry, hi8);
                }
                stomp_setprio(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) { e[q] = lo8[q]; e[8 + q] = hi8[q]; }
            }
        };
        // acc += (block b of L) * e, issued transposed; both column blocks of the 64-column chunk
        auto product = [&](auto bc, const float (&e)[16]) {
            constexpr int b = decltype(bc)::value;
            constexpr bool full = b == 1;
            const unsigned* img = Limg + (b == 0 ? 0 : (b == 1 ? STOMP_LIMG_WORDS : STOMP_LIMG_WORDS + STOMP_LIMG_WORDS_FULL));
            StompEps8 sp;
            float eh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) eh[u] = e[u];
            stomp_split_product<0, full, true>(img, eh, sp, eps != nullptr, j, g, acc);
#pragma unroll
            for (int u = 0; u < 8; ++u) eh[u] = e[8 + u];
            stomp_split_product<1, full, true>(img, eh, sp, eps != nullptr, j, g, acc);
        };
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        using B2 = std::integral_constant<int, 2>;
        float e[16];
        operand(hcw, e);                                   // every wave draws the columns of ITS chunk
        if (HC == 1) {
            product(B0{}, e);
        } else {
            // the second-chunk wave of a rollout also needs the first chunk's columns: its partner has just drawn them and
            // hands them over through the (free) tile of the second-chunk wave instead of both drawing them (device
            // noise only; injected noise is simply loaded twice)
            const bool share = eps == nullptr;
            float* hand = tiles + wave_of(rw, 1) * TILE + (g * d + (j < d ? j : 0)) * 16;    // 4 d lanes x 16 floats = the tile
            if (share && hcw == 0 && j < d) {
#pragma unroll
                for (int v = 0; v < 4; ++v) reinterpret_cast<f32x4*>(hand)[v] = f32x4{e[4 * v], e[4 * v + 1], e[4 * v + 2], e[4 * v + 3]};
            }
            if (share) __syncthreads();
            if (hcw == 0) {
                product(B0{}, e);
            } else {
                float e0[16];
                if (share) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const f32x4 t = (j < d) ? reinterpret_cast<const f32x4*>(hand)[v] : f32x4{0.f, 0.f, 0.f, 0.f};
                        e0[4 * v] = t[0]; e0[4 * v + 1] = t[1]; e0[4 * v + 2] = t[2]; e0[4 * v + 3] = t[3];
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    operand(0, e0);
                }
                product(B1{}, e0);                         // same accumulation order as the two-kernel path: kc = 0, then 1
                product(B2{}, e);
            }
        }
        float nz[16];
        hx_transpose_to_rows(acc, nz);
        if (DCH != 0 && (DCH % 2) == 0) {
#pragma unroll
            for (int c = 0; c < DX; c += 2) *reinterpret_cast<float2*>(nt + lane * DX + c) = make_float2(nz[c], nz[c + 1]);
        } else {
#pragma unroll
            for (int c = 0; c < DX; ++c)
                if (c < d) nt[lane * d + c] = nz[c];
        }
        __builtin_amdgcn_wave_barrier();
    };

    if (n_run > 0) draw_noise(0, (chunk * nb) * RB + rw);

    for (int it = 0; it < n_run; ++it) {
        float m_run = -3.0e38f, z_run = 0.f, d_run[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) d_run[u] = 0.f;
#pragma nounroll
        for (int bt = 0; bt < nb; ++bt) {
            HX_FRESH_LANE();
            const int s = (chunk * nb + bt) * RB + rw;                       // this wave's sample in this batch
            const bool live = s < S;
            // ============ A. samples: x = mean + noise (zero at both ends, stomp.py:105-106), stored, kept packed in the tile
            const int h = 64 * hcw + lane;
            const bool on = h < H;
            float x[DX];
            if (DCH != 0 && (DCH % 2) == 0) {
#pragma unroll
                for (int c = 0; c < DX; c += 2) {
                    const float2 nv = *reinterpret_cast<const float2*>(nt + lane * DX + c);
                    const float2 mv = *reinterpret_cast<const float2*>(mean_l + (on ? h : 0) * DX + c);
                    x[c] = mv.x + nv.x;                  // (rows 0 / H - 1 of the noise are exact zeros: the L image's rows are)
                    x[c + 1] = mv.y + nv.y;
                }
#pragma unroll
                for (int c = 0; c < DX; c += 2) *reinterpret_cast<float2*>(nt + lane * DX + c) = make_float2(x[c], x[c + 1]);
            } else {
#pragma unroll
                for (int c = 0; c < DX; ++c) x[c] = (c < d) ? mean_l[(on ? h : 0) * d + c] + nt[lane * d + c] : 0.f;
#pragma unroll
                for (int c = 0; c < DX; ++c)
                    if (c < d) nt[lane * d + c] = x[c];
            }
            __builtin_amdgcn_wave_barrier();
            if (live) {
                float* sbase = samples + (((size_t)p * S + s) * H + 64 * hcw) * d;       // uniform
                const int nfl = min(H - 64 * hcw, 64) * d;                              // floats of this chunk (<= 0: none)
                if (((H * d) & 3) == 0 && ((64 * d) & 3) == 0) {
                    const f32x4* pk4 = reinterpret_cast<const f32x4*>(nt);
                    f32x4* out4 = reinterpret_cast<f32x4*>(sbase);
                    // (a fixed trip count, masked: the open-ended loop kept a 64-bit induction pointer alive across the phase)
#pragma unroll
                    for (int k = 0; k < (16 * DX + 63) / 64; ++k) {
                        const int idx = lane + 64 * k;
                        if (4 * idx < nfl) out4[idx] = pk4[idx];
                    }
                } else {
                    for (int idx = lane; idx < nfl; idx += 64) sbase[idx] = nt[idx];
                }
            }
            // ============ B. collision cost of this chunk of the rollout
            {
                float q[MPB_MAX_DOF];
                // (a row of d channels holds D = d or D = d / 2 joint positions: with more channels than MPB_MAX_DOF it must be d / 2 -- the
            // joints beyond that are zeros the compiler can fold, which keeps the d = 14 kernels at the registers they had with 8)
            constexpr int DQ_ = DCH ? ((DCH > MPB_MAX_DOF) ? DCH / 2 : DCH) : MPB_MAX_DOF;     // (run-time d: any joint count)
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < DQ_) ? x[i < DQ_ ? i : 0] : 0.f;
                float c = 0.f;
                bool bad = false;
                GeomView G = G0;
                for (const float* gp = geom;;) {
                    if (gp != geom) {     // a chained field: its grid replaces the first one's (restored before the next pass)
                        __syncthreads();
                        stage_field(G, tid);
                        __syncthreads();
                    }
                    // (the launcher picked LIST from geom_flags; a device header of the other format poisons the cost instead of
                    // being mis-read)
                    const bool fmt_ok = LIST ? list_usable(G) : (G.version == MPB_GEOM_VERSION);
                    if (live && on && h >= 1 && fmt_ok) {
                        if (MODEL == PandaModel::ID) {
                            if (G.model == PandaModel::ID) c = fmaf(G.fscale, waypoint_cost_grid_model<PandaModel, true, LIST>(G, gridw, otab, q, &LV), c);
                        } else {
                            c = fmaf(G.fscale, waypoint_cost_grid<true, LIST>(G, gridw, otab, q, &LV), c);
                        }
                    }
                    if (!fmt_ok || (MODEL != 0 && G.model != MODEL)) bad = true;    // (wave-uniform: lane 0 -- possibly waypoint 0, outside the walk -- writes the cost)
                    if (G.next == 0) break;
                    gp += G.next;
                    G = geom_view(gp);
                }
                if (G0.next != 0) {
                    __syncthreads();
                    int tid_o = tid;                         // (opaque: this rare path's source addresses must not be hoisted)
                    asm volatile("" : "+v"(tid_o));
                    stage_field(G0, tid_o);
                }
                const double csum = wave_sum_f64((double)c);
                if (lane == 0) cst[wave] = bad ? __uint_as_float(0x7FC00000u) : (float)csum;
            }
            __syncthreads();                                                                    // (1) costs of the pass
            // ============ C. fold this batch into the running partial: logits, batch max, e_w, z, weighted (sample - mean)
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const int lq = tq & 63;
            const int rq = lq % RB;                                       // lanes 0..RB-1 carry the rollouts (every wave redundantly)
            const float c0 = cst[wave_of(rq, 0)], c1 = (HC == 2) ? cst[wave_of(rq, 1)] : 0.f;
            const bool poisoned = __float_as_uint(c0) == 0x7FC00000u;     // (geometry / kernel mismatch: see mpb_geom_flags)
            const float cw = weight * (k_sigma * (poisoned ? 0.f : c0 + c1));
            const int sl = (chunk * nb + bt) * RB + rq;
            const bool carries = (lq & 15) < RB && sl < S;                 // (all FOUR rows of a wave carry the pass's costs: below)
            if (tq < RB && sl < S) reinterpret_cast<unsigned*>(costs)[(size_t)p * S + sl] = poisoned ? 0x7FC00000u : __float_as_uint(cw);
            const float xs = carries ? -cw * inv_temperature : -3.0e38f;
            // (row reductions over replicated rows and DPP row_newbcast in the weighted sum: csrc/mpb_stomp_fused.hip, phase C -- same bits
            // as the full-wave reductions over one carrying row, no v_readlane)
            const float mb = row_max_f32(xs);
            const float ex = carries ? fast_expf(xs - mb) : 0.f;
            const float zb = row_sum_f32(ex);
            if (tq < RB) ewl[bt * RB + tq] = ex;
            if (tq == 0) mbl[bt] = mb;
            const float m_new = fmaxf(m_run, mb);
            const float f_old = fast_expf(m_run - m_new), f_b = fast_expf(mb - m_new);
            z_run = fmaf(zb, f_b, z_run * f_old);
            m_run = m_new;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e_raw = tq + FUSED_THREADS * u;
                // WHOLE waves run the weighted sum (N = H d need not be a multiple of 64): DPP reads the source lanes' registers
                // only while those lanes are enabled -- a lane row cut by `e < N` would drop the terms of its disabled source lanes.
                // The lanes past N work on element N - 1 and their result is never read.
                if ((e_raw & ~63) < N) {
                    const int e = (e_raw < N) ? e_raw : N - 1;
                    const int hc_e = (HC == 1) ? 0 : (e >= 64 * d ? 1 : 0);
                    const int off = e - 64 * d * hc_e;                  // rows of a tile are packed with stride d
                    const float mu = mean_l[e];
                    float dp = 0.f, df[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) df[r] = tiles[wave_of(r, hc_e) * TILE + off] - mu;
                    fmac_row_bcast_seq(dp, ex, df);             // (one asm statement with its own DPP wait states: mpb_common.h)
                    d_run[u] = fmaf(dp, f_b, d_run[u] * f_old);
                }
            }
            __syncthreads();                                                     // (2) the samples in the tiles are consumed
            // ============ the noise of the next pass (next batch, or the first batch of the next iteration): independent of
            //              the means, so it is drawn here, ahead of the exchange
            const bool more = bt + 1 < nb;
            const int it_n = more ? it : it + 1;
            if (it_n < n_run) draw_noise(it_n, (chunk * nb + (more ? bt + 1 : 0)) * RB + rw);
        }
        __builtin_amdgcn_s_setprio(0);
        HX_FRESH_LANE();
        int tq = tid;
        asm volatile("" : "+v"(tq));
        // ============ D. exchange: publish this unit's partial, wait for the partners', combine in chunk order
        float m_all = m_run, z_all = z_run, dsum[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) dsum[u] = d_run[u];
        const unsigned tag = tag0 + (unsigned)it;
        if (exchange) {
            granule_t* mine = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc + chunk) * HX_XCHG;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tq + FUSED_THREADS * u;
                if (e < N) st_granule(mine + 2 + e, d_run[u], tag);
            }
            if (tq == 0) { st_granule(mine + 0, m_run, tag); st_granule(mine + 1, z_run, tag); }
            float mk[FUSED_MAX_CHUNKS], zk[FUSED_MAX_CHUNKS], dk[FUSED_MAX_CHUNKS][EPT];
            const granule_t* slot0 = xch + ((size_t)(it & 1) * P * nc + (size_t)p * nc) * HX_XCHG;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < FUSED_MAX_CHUNKS; ++k) {
                    mk[k] = -3.0e38f; zk[k] = 0.f;
#pragma unroll
                    for (int u = 0; u < EPT; ++u) dk[k][u] = 0.f;
                    if (k < nc) {
                        const granule_t* theirs = slot0 + (size_t)k * HX_XCHG;
                        const granule_t gm = ld_granule(theirs + 0), gz = ld_granule(theirs + 1);
                        ok = ok && (unsigned)(gm >> 32) == tag && (unsigned)(gz >> 32) == tag;
                        mk[k] = __uint_as_float((unsigned)gm);
                        zk[k] = __uint_as_float((unsigned)gz);
#pragma unroll
                        for (int u = 0; u < EPT; ++u) {
                            const int e = tq + FUSED_THREADS * u;
                            if (e < N) {
                                const granule_t gd = ld_granule(theirs + 2 + e);
                                ok = ok && (unsigned)(gd >> 32) == tag;
                                dk[k][u] = __uint_as_float((unsigned)gd);
                            }
                        }
                    }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { s_abort = 1; break; }
            }
            m_all = mk[0];
#pragma unroll
            for (int k = 1; k < FUSED_MAX_CHUNKS; ++k) m_all = fmaxf(m_all, mk[k]);
            z_all = 0.f;
#pragma unroll
            for (int u = 0; u < EPT; ++u) dsum[u] = 0.f;
#pragma unroll
            for (int k = 0; k < FUSED_MAX_CHUNKS; ++k) {
                if (k < nc) {
                    const float f = fast_expf(mk[k] - m_all);
                    z_all = fmaf(f, zk[k], z_all);
#pragma unroll
                    for (int u = 0; u < EPT; ++u) dsum[u] = fmaf(f, dk[k][u], dsum[u]);
                }
            }
        }
        // ============ E. weights out, delta (transposed) -> mean += lr * Sigma @ delta on the matrix cores
        const float rz = fast_rcpf(z_all);          // (z >= 1; mpb_stomp_fused.hip, phase E)
        for (int i = tq; i < nb * RB; i += FUSED_THREADS) {
            const int sl = chunk * nb * RB + i;
            if (sl < S) weights[(size_t)p * S + sl] = ewl[i] * fast_expf(mbl[i / RB] - m_all) * rz;
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = tq + FUSED_THREADS * u;
            if (e < N) {
                const int hh = e / d, cc = e - hh * d;
                delta[cc * DLD + hh] = dsum[u] * rz;
            }
        }
        __syncthreads();                                                                        // (3) delta complete
        if (s_abort) break;                                                                     // block-uniform (set before barrier 3)
        // the rows of Sigma this wave's 16-row tile needs, all fetched up front (from L2: every workgroup reads the same
        // H x H matrix every iteration).  (Fetched ahead of the exchange instead, so that the latency would run under the
        // poll, the 32 registers spill: 232 B / lane of scratch.)
        f32x4 sig_a[HP / 16];
        if (wave < HP / 16) {
            const int row = 16 * wave + j;
            const bool vec = (H & 3) == 0;
#pragma unroll
            for (int ks4 = 0; ks4 < HP / 16; ++ks4) {
                const int k0 = 16 * ks4 + 4 * g;
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                if (HC == 1) {
                    a = *reinterpret_cast<const f32x4*>(sig_l + row * DLD + k0);
                } else if (row < H) {
                    if (vec) {
                        if (k0 < H) a = *reinterpret_cast<const f32x4*>(Sigma + (size_t)row * H + k0);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (k0 + kk < H) a[kk] = Sigma[(size_t)row * H + k0 + kk];
                    }
                }
                sig_a[ks4] = a;
            }
        }
        if (wave < HP / 16) {
            // row tile `wave` of Sigma @ delta: A = Sigma[16 w + i][k], B = delta[k][c = j];
            // both operands deliver the k-set {16 ks4 + 4 g + kk} per step (any order of k is the same sum up to rounding)
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks4 = 0; ks4 < HP / 16; ++ks4) {
                const f32x4 a = sig_a[ks4];
                const f32x4 b = *reinterpret_cast<const f32x4*>(delta + j * DLD + 16 * ks4 + 4 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int hh = 16 * wave + 4 * g + rr;
                if (hh < H && j < d) mean_l[hh * d + j] += lr * acc[rr];
            }
        }
        __syncthreads();                                                                        // (4) new mean visible, tiles free
    }
    const int aborted = s_abort;
    if (!aborted && chunk == 0) {
        for (int e = tid; e < N; e += FUSED_THREADS) {
            const float m = mean_l[e];
            means[(size_t)p * N + e] = m;
            if (means_copy) means_copy[(size_t)p * N + e] = m;
        }
    }
    if (aborted == 1 && chunk == 0 && means_copy)
        for (int e = tid; e < N; e += FUSED_THREADS) means_copy[(size_t)p * N + e] = means[(size_t)p * N + e];
    if (aborted == 2 && (int)blockIdx.x < P && means_copy)       // (header not zeroed: no unit was drawn)
        for (int e = tid; e < N; e += FUSED_THREADS) means_copy[(size_t)blockIdx.x * N + e] = means[(size_t)blockIdx.x * N + e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's result stores acknowledged before thread 0's release of the status tag (mpb_stomp_fused.hip)
    __syncthreads();
    if (tid == 0) fused_leave(wsu, status_host, tag0, aborted);
}

// ------------------------------------------------------------------------------------------------
// launcher (called by mpb_stomp_run_checked, mpb_stomp_fused.hip)
// ------------------------------------------------------------------------------------------------
// can this kernel serve the shape, and with which split of the S samples over workgroups (nc) and passes (nb)?
bool mpb_fused_hx_plan(int geom_flags, int n_cu, int P, int S, int H, int d, int* nc_out, int* nb_out, size_t* ws_bytes) {
    // every field grid-backed: compact grids (flag bit 8) at any horizon, list grids (bit 13, round 6) up to 64 support points
    if (P < 1 || S < 1 || S > 128 || H < 3 || H > 128 || d < 1 || d > 16) return false;
    if (!(geom_flags & 0x100) && !((geom_flags & 0x2000) && H <= 64)) return false;
    const int HC = H > 64 ? 2 : 1, RB = FUSED_WAVES / HC;
    const int passes = (S + RB - 1) / RB;                 // passes of RB rollouts a particle needs per iteration
    // few particles: as many workgroups per particle as fit the chip in one round (and the exchange allows); many
    // particles: one workgroup per particle running every pass itself (no exchange)
    int nc = n_cu / P;
    if (nc > FUSED_MAX_CHUNKS) nc = FUSED_MAX_CHUNKS;
    if (nc > passes) nc = passes;
    if (nc < 1) nc = 1;
    int nb = (passes + nc - 1) / nc;
    if (nb > HX_MAX_NB) {                                  // more passes than a workgroup may run: more workgroups
        nc = (passes + HX_MAX_NB - 1) / HX_MAX_NB;
        if (nc > FUSED_MAX_CHUNKS) return false;
        nb = (passes + nc - 1) / nc;
    }
    *nc_out = nc;
    *nb_out = nb;
    *ws_bytes = (nc > 1 ? hx_ws_floats(P, nc) : FUSED_HDR_WORDS) * sizeof(float);
    return true;
}

int mpb_fused_hx_launch(float* means, const float* eps, float* samples, float* costs, float* weights, const float* L,
                        const float* Sigma, const float* geom, int geom_flags, float* workspace, int P, int S, int H, int d, int nc,
                        int nb, float k_sigma, float weight, float lr, float temperature, int n_iters, uint32_t lo, uint32_t hi,
                        uint32_t iter0, uint32_t particle_offset, uint32_t tag0, unsigned long long timeout, unsigned* status_dev,
                        float* means_copy, hipStream_t st, const FusedProfile* prof) {
    const dim3 grid(P * nc), block(FUSED_THREADS);
    const int model = geom_flags & 0xFF;
#define MPB_HX_LAUNCH_(DCH, MODEL, HC, INJ, LIST)                                                                                        \
    MPB_FUSED_LAUNCH(prof, (stomp_fused_hx_kernel<DCH, MODEL, HC, INJ, LIST>), grid, block, st, means, eps, samples, costs, weights, L,  \
                     Sigma, geom, workspace, P, S, H, d, nc, nb, k_sigma, weight, lr, temperature, n_iters, lo, hi, iter0,         \
                     particle_offset, tag0, timeout, status_dev, means_copy)
#define MPB_HX_LAUNCH(DCH, MODEL, HC)                              \
    do {                                                           \
        if (eps) MPB_HX_LAUNCH_(DCH, MODEL, HC, true, false);      \
        else MPB_HX_LAUNCH_(DCH, MODEL, HC, false, false);         \
    } while (0)
#define MPB_HX_LAUNCH_LIST(DCH, MODEL)                             \
    do {                                                           \
        if (eps) MPB_HX_LAUNCH_(DCH, MODEL, 1, true, true);        \
        else MPB_HX_LAUNCH_(DCH, MODEL, 1, false, true);           \
    } while (0)
    if (!(geom_flags & 0x100)) {              // list grids (mpb_fused_hx_plan admitted them: H <= 64)
        if (model == PandaModel::ID && d == 7) MPB_HX_LAUNCH_LIST(7, PandaModel::ID);
        else if (model == PandaModel::ID && d == 14) MPB_HX_LAUNCH_LIST(14, PandaModel::ID);
        else MPB_HX_LAUNCH_LIST(0, 0);
    } else if (H > 64) {
        if (model == PandaModel::ID && d == 7) MPB_HX_LAUNCH(7, PandaModel::ID, 2);
        else if (model == PandaModel::ID && d == 14) MPB_HX_LAUNCH(14, PandaModel::ID, 2);
        else MPB_HX_LAUNCH(0, 0, 2);
    } else {
        if (model == PandaModel::ID && d == 7) MPB_HX_LAUNCH(7, PandaModel::ID, 1);
        else if (model == PandaModel::ID && d == 14) MPB_HX_LAUNCH(14, PandaModel::ID, 1);
        else MPB_HX_LAUNCH(0, 0, 1);
    }
#undef MPB_HX_LAUNCH
#undef MPB_HX_LAUNCH_LIST
#undef MPB_HX_LAUNCH_
    return mpb_check_launch("mpb_stomp_run");
}
