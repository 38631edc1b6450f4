// mpb_stomp_noise.h -- the time-correlated STOMP noise  N = L * eps  of one rollout per wave on the matrix cores
// (shared by the two-kernel path, mpb_kernels.hip, and the persistent fused kernel, mpb_stomp_fused.hip).
//
// L (64 x 64 lower-triangular scale_tril of the precision matrix R) is the A operand, eps (64 x d standard normals, generated
// in registers -- Philox + Box-Muller -- or loaded) the B operand of v_mfma_f32_16x16x32_bf16 tiles on an exact three-way bf16
// split of both (below); lower-triangular: row tile m only needs the 32-column blocks up to its diagonal.
// (stomp_l_image_index is the permuted fp32 image of rounds 1-3, still used by the MPPI kernel's fp32 product.)
#pragma once
#include "mpb_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NT_STRIDE 20  // floats per waypoint row of a wave's noise tile: 80 B keeps ds_read_b128 conflict-free

// Standard normals of a lane for one 64-column chunk: SIXTEEN per (particle, sample, channel j, k-group g), from THREE
// Philox4x32-7 calls (round 4; rounds 1-3 spent a call per four normals): the twelve words are cut into sixteen 23-bit
// fields, each dropped straight into the mantissa of a float in [1, 2) -- word triple (a, b, c) -> a[31:9], a[8:0] b[31:18],
// b[17:0] c[31:27], c[22:0]; one v_alignbit or v_and_or per field, no integer-to-float conversion, no scaling --, and
// Box-Muller takes them in pairs (m_2i -> radius: u = 2 - m in (0, 1], exact; m_2i+1 -> angle: v_sin / v_cos count in
// revolutions and have period 1, so m itself is the argument).  Normal u is eps[j][column] with the column the caller's
// (H = 64 paths: stomp_eps_column(g, u >> 2, u & 3) = 32 (u >> 3) + 8 g + (u & 7); the chunked paths add 64 kc).  The counter
// holds the GLOBAL particle id (the noise does not depend on the sharding nor on which kernel draws it), word 2 = (j << 16) |
// (g << 8) | call with call = (kc << 4) | {0, 1, 2}.  The draw comes in two halves so that the matrix product of the first
// column block runs between them: normals 0-7 need words 0-5 (calls 0, 1; words 6, 7 are carried), normals 8-15 words 6-11.
// What enters the product is the normal cut to SIXTEEN significant bits (its two leading bf16 components, below): a drawn
// normal is DEFINED as that value (stomp_eps_quantise; mpb_debug_stomp_normals returns it).
__device__ __forceinline__ void box_muller23(uint32_t m1, uint32_t m2, float& n0, float& n1) {
    // (no contraction: the normal is the ROUNDED product r * cos -- the split below subtracts from it, and fma(r, cos, -h) is not
    // (r * cos) - h; the compiler did fuse the two in one of the kernels that draw this stream and not in the others)
#pragma clang fp contract(off)
    const float u1 = 2.0f - __uint_as_float(m1);                                  // (0, 1], exact (Sterbenz)
    const float ang = __uint_as_float(m2);                                        // revolutions, [1, 2)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // raw v_sqrt_f32 / v_log_f32 (log2)
    n0 = r * __builtin_amdgcn_cosf(ang);
    n1 = r * __builtin_amdgcn_sinf(ang);
}
__device__ __forceinline__ void stomp_fields4(uint32_t a, uint32_t b, uint32_t c, float (&n)[4]) {
    constexpr uint32_t ONE = 0x3F800000u, MANT = 0x007FFFFFu;
    const uint32_t f0 = __builtin_amdgcn_alignbit(0x7Fu, a, 9), f1 = (__builtin_amdgcn_alignbit(a, b, 18) & MANT) | ONE,
                   f2 = (__builtin_amdgcn_alignbit(b, c, 27) & MANT) | ONE, f3 = (c & MANT) | ONE;
    box_muller23(f0, f1, n[0], n[1]);
    box_muller23(f2, f3, n[2], n[3]);
}
#define STOMP_PRIO_NONE 0
#define STOMP_PRIO_PROGRESS 1
#define STOMP_PRIO_STAGGER 2
__device__ __forceinline__ void stomp_setprio(int level) {       // (the operand of s_setprio is an immediate)
    switch (level) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
// normals 0-7 (column block 0 of the chunk); `carry` = words 6, 7 for stomp_normals_hi.  PRIO == STOMP_PRIO_PROGRESS: the issue
// priority falls as the wave advances through its three calls (3, 2, then 1 in stomp_normals_hi; see stomp_eps8)
template <int PRIO>
__device__ __forceinline__ void stomp_normals_lo(uint32_t p_global, uint32_t s, uint32_t j, uint32_t g, uint32_t call0, uint32_t iter,
                                                 uint32_t seed_lo, uint32_t seed_hi, float (&n)[8], uint32_t (&carry)[2]) {
    const uint32_t z = (j << 16) | (g << 8) | call0;
    if (PRIO == STOMP_PRIO_PROGRESS) stomp_setprio(3);
    const uint4 r0 = philox4x32<7>(make_uint4(p_global, s, z, iter), make_uint2(seed_lo, seed_hi));
    float a[4], b[4];
    stomp_fields4(r0.x, r0.y, r0.z, a);
    // (one call at a time: interleaved by the scheduler the two calls and their Box-Muller chains want ~15 more VGPRs than
    // the phase has -- they went to scratch)
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO == STOMP_PRIO_PROGRESS) stomp_setprio(2);
    const uint4 r1 = philox4x32<7>(make_uint4(p_global, s, z + 1u, iter), make_uint2(seed_lo, seed_hi));
    stomp_fields4(r0.w, r1.x, r1.y, b);
#pragma unroll
    for (int q = 0; q < 4; ++q) { n[q] = a[q]; n[4 + q] = b[q]; }
    carry[0] = r1.z;
    carry[1] = r1.w;
}
template <int PRIO>
__device__ __forceinline__ void stomp_normals_hi(uint32_t p_global, uint32_t s, uint32_t j, uint32_t g, uint32_t call0, uint32_t iter,
                                                 uint32_t seed_lo, uint32_t seed_hi, const uint32_t (&carry)[2], float (&n)[8]) {
    if (PRIO == STOMP_PRIO_PROGRESS) stomp_setprio(1);
    const uint4 r2 = philox4x32<7>(make_uint4(p_global, s, ((j << 16) | (g << 8) | call0) + 2u, iter), make_uint2(seed_lo, seed_hi));
    float a[4], b[4];
    stomp_fields4(carry[0], carry[1], r2.x, a);
    stomp_fields4(r2.y, r2.z, r2.w, b);
#pragma unroll
    for (int q = 0; q < 4; ++q) { n[q] = a[q]; n[4 + q] = b[q]; }
}

// index of L[row][col] in the permuted LDS image: Lp[(((m*4 + ks4)*4 + g)*16 + i)*4 + kk] = L[16m+i][4*(4*ks4+kk) + g]
__device__ __forceinline__ int stomp_l_image_index(int row, int col) {
    const int m = row >> 4, i = row & 15, ks = col >> 2, gq = col & 3;
    return ((((m * 4 + (ks >> 2)) * 4 + gq) * 16 + i) << 2) + (ks & 3);
}

// ------------------------------------------------------------------------------------------------
// The product on the bf16 matrix pipe, EXACT in fp32 inputs (round 4; rounds 1-3: 40 v_mfma_f32_16x16x4_f32 per rollout).
//
// v_mfma_f32_16x16x4_f32 runs on the fp32 multipliers the vector instructions use: 32 cycles for 1 024 multiply-adds, no
// overlap with another wave's VALU work (profiles/r03_microbench_overlap.txt) -- the 40 MFMAs of the product were 15 % of
// the C3 iteration.  v_mfma_f32_16x16x32_bf16 does 8 192 multiply-adds in ~16-20 cycles and holds the vector issue for
// about half of them (profiles/r04_microbench_overlap_bf16.txt).  An fp32 value is the exact sum of three bf16 values
// (truncation split: 8 + 8 + 8 significant bits, same exponent range), so
//     L * eps = sum over the six component pairs (h,h) (h,m) (m,h) (h,l) (l,h) (m,m) of  L_a * eps_b   + O(2^-24 |L||eps|)
// with every bf16 x bf16 product exact in the fp32 accumulator: the dropped pairs (m,l) (l,m) (l,l) are below the rounding of
// an fp32 dot product.  36 bf16 MFMAs (6 lower-triangular (row tile, 32-column block) pairs x 6 component pairs) replace
// 40 fp32 ones at a quarter of the pipe time; splitting eps costs ~11 VALU instructions per pair of values.
//
// Operand layout of a 16 x 16 x 32 tile: lane (i = l & 15, g = l >> 4) holds A[i][8 g + e], e = 0..7, and B[8 g + e][j = l & 15];
// D as for the fp32 form (lane (j, g) holds D[4 g + rr][j]).  Column k = 32 kb + 8 g + e of a 64-column chunk.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// image of one 64 x 64 block of L: a DIAGONAL (lower-triangular) block stores 6 (row tile m, column block kb) tiles --
// (0,0) (1,0) (2,0) (3,0) (2,1) (3,1); the two above the diagonal are zero and never multiplied -- a FULL (off-diagonal,
// horizons beyond 64) block all 8
#define STOMP_LIMG_TILES 6
#define STOMP_LIMG_WORDS (3 * STOMP_LIMG_TILES * 4 * 16 * 4)  // 32-bit words of a diagonal block's three-component image: 18 KB
#define STOMP_LIMG_WORDS_FULL (3 * 8 * 4 * 16 * 4)            // of a full block's: 24 KB
template <bool FULL>
__device__ __forceinline__ constexpr int stomp_limg_tile(int m, int kb) { return FULL ? m + 4 * kb : (kb == 0 ? m : 2 + m); }

// x = h + m + l exactly, each a bf16 value held in the top half of an fp32 word (truncation split)
__device__ __forceinline__ void stomp_split3(float x, unsigned& h, unsigned& m, unsigned& l) {
#pragma clang fp contract(off)
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);                                  // (only the top halves of h, m, l are used)
}
// top halves of two words -> one word of two bf16 (lo: element 2 i, hi: element 2 i + 1)
__device__ __forceinline__ unsigned stomp_pack_top(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// four consecutive columns col0 .. col0 + 3 (col0 a multiple of 4) of row `row` of a 64 x 64 block into the block's
// three-component bf16 image (32-bit words; entry ((c * TILES + t) * 4 + g) * 16 + i = eight bf16 of row 16 m + i, columns
// 32 kb + 8 g .. + 7).  Diagonal block: the tiles above the diagonal (m < 2, kb = 1) are not stored.
template <bool FULL = false>
__device__ __forceinline__ void stomp_l_image_store(unsigned* __restrict__ img, int row, int col0, f32x4 lv) {
    const int m = row >> 4, i = row & 15, kb = col0 >> 5, g = (col0 & 31) >> 3, e0 = col0 & 7;
    if (!FULL && kb == 1 && m < 2) return;
    unsigned h[4], md[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) stomp_split3(lv[e], h[e], md[e], l[e]);
    const int t = stomp_limg_tile<FULL>(m, kb);
    const int w = ((t * 4 + g) * 16 + i) * 4 + (e0 >> 1);
    constexpr int CW = (FULL ? 8 : STOMP_LIMG_TILES) * 4 * 16 * 4;   // words per component
    *reinterpret_cast<uint2*>(img + w) = make_uint2(stomp_pack_top(h[0], h[1]), stomp_pack_top(h[2], h[3]));
    *reinterpret_cast<uint2*>(img + CW + w) = make_uint2(stomp_pack_top(md[0], md[1]), stomp_pack_top(md[2], md[3]));
    *reinterpret_cast<uint2*>(img + 2 * CW + w) = make_uint2(stomp_pack_top(l[0], l[1]), stomp_pack_top(l[2], l[3]));
}

// the eight values v[e] = eps[c][32 kb + 8 g + e] of a lane as the three bf16 operand components.  LOW = false (drawn
// normals): only the two leading components -- the value that enters the product is v cut to 16 significant bits, which IS the
// drawn normal by definition (stomp_eps_quantise); 6 instead of 13 vector instructions per pair of values and one matrix
// instruction less per tile.  Injected normals (LOW = true) are arbitrary fp32 values and keep all three.
struct StompEps8 { u32x4 h, m, l; };
template <bool LOW = true>
__device__ __forceinline__ void stomp_split8(const float (&v)[8], StompEps8& b) {
#pragma clang fp contract(off)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (LOW) {
            unsigned h0, m0, l0, h1, m1, l1;
            stomp_split3(v[2 * q], h0, m0, l0);
            stomp_split3(v[2 * q + 1], h1, m1, l1);
            b.h[q] = stomp_pack_top(h0, h1);
            b.m[q] = stomp_pack_top(m0, m1);
            b.l[q] = stomp_pack_top(l0, l1);
        } else {
            const unsigned h0 = __float_as_uint(v[2 * q]), h1 = __float_as_uint(v[2 * q + 1]);
            const float r0 = v[2 * q] - __uint_as_float(h0 & 0xFFFF0000u), r1 = v[2 * q + 1] - __uint_as_float(h1 & 0xFFFF0000u);
            b.h[q] = stomp_pack_top(h0, h1);
            b.m[q] = stomp_pack_top(__float_as_uint(r0), __float_as_uint(r1));
            b.l[q] = 0u;
        }
    }
}
// a drawn normal as it enters the product: its two leading bf16 components
__device__ __forceinline__ float stomp_eps_quantise(float x) {
#pragma clang fp contract(off)
    const float h = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    return h + __uint_as_float(__float_as_uint(x - h) & 0xFFFF0000u);
}

// column of a 64-column chunk that normal u = 4 q4 + r of lane group g lands on (stomp_normals_lo / _hi): 32 (u >> 3) + 8 g + (u & 7)
__device__ __forceinline__ constexpr int stomp_eps_column(int g, int q4, int r) { return 32 * (q4 >> 1) + 8 * g + 4 * (q4 & 1) + r; }

// acc[m] += rows 16 m .. 16 m + 15 of (column block KB of the 64 x 64 block behind `img`) * eps, for the row tiles the
// column block reaches (diagonal block: KB = 0 all four, KB = 1 m = 2, 3; full block: all four).  Small component pairs
// first; LOW = false: eps has no third component (drawn normals, stomp_split8<false>).  TRANSPOSED: the product is issued as eps * L^T -- lane (j, g) then holds D[channel 4 g + rr][waypoint 16 m + j]
// (the generalised persistent kernel transposes by lane permutes instead of an LDS round trip); operand registers are
// the same either way.
template <int KB, bool FULL = false, bool TRANSPOSED = false, bool LOW = true>
__device__ __forceinline__ void stomp_noise_product_kb(const unsigned* __restrict__ img, const StompEps8& b, int j, int g, f32x4 (&acc)[4]) {
    const u32x4* img4 = reinterpret_cast<const u32x4*>(img);
    constexpr int CQ = (FULL ? 8 : STOMP_LIMG_TILES) * 4 * 16;       // 16-byte entries per component
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.h), bm = __builtin_bit_cast(bf16x8, b.m), bl = __builtin_bit_cast(bf16x8, b.l);
#pragma unroll
    for (int m = ((KB == 0 || FULL) ? 0 : 2); m < 4; ++m) {
        const int q = (stomp_limg_tile<FULL>(m, KB) * 4 + g) * 16 + j;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, img4[q]), am = __builtin_bit_cast(bf16x8, img4[CQ + q]),
                     al = __builtin_bit_cast(bf16x8, img4[2 * CQ + q]);
        if (!TRANSPOSED) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[m], 0, 0, 0);
            if (LOW) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[m], 0, 0, 0);
        } else {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
            if (LOW) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
        }
    }
}

// the eight values of column block KB for lane (j, g): drawn (stomp_normals_lo / _hi: `carry` links the two halves) or loaded
// (eps_s: pre-drawn normals laid out (d, P, H) for this sample, pointer already at [s]); zero for the padding channels j >= DCH
template <int DCH, int KB, int PRIO>
__device__ __forceinline__ void stomp_eps8(float (&v)[8], uint32_t (&carry)[2], const float* __restrict__ eps_s, int P, int p, int j,
                                           int g, uint32_t p_global, uint32_t s, uint32_t iter, uint32_t seed_lo, uint32_t seed_hi) {
    constexpr int H = 64;
    if (eps_s != nullptr) {
        const f32x4* ep = reinterpret_cast<const f32x4*>(eps_s + ((size_t)(j < DCH ? j : 0) * P + p) * H + 32 * KB + 8 * g);
        const f32x4 a = (j < DCH) ? ep[0] : f32x4{0.f, 0.f, 0.f, 0.f}, b = (j < DCH) ? ep[1] : f32x4{0.f, 0.f, 0.f, 0.f};
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        if (j < DCH) {
            if (KB == 0) stomp_normals_lo<PRIO>(p_global, s, (uint32_t)j, (uint32_t)g, 0u, iter, seed_lo, seed_hi, v, carry);
            else stomp_normals_hi<PRIO>(p_global, s, (uint32_t)j, (uint32_t)g, 0u, iter, seed_lo, seed_hi, carry, v);
        }
    }
}

// split + multiply one column block: injected values (wave-uniform `injected`) with all three components, drawn ones with two
template <int KB, bool FULL = false, bool TRANSPOSED = false>
__device__ __forceinline__ void stomp_split_product(const unsigned* __restrict__ img, const float (&v)[8], StompEps8& b, bool injected,
                                                    int j, int g, f32x4 (&acc)[4]) {
    if (injected) {
        stomp_split8<true>(v, b);
        stomp_noise_product_kb<KB, FULL, TRANSPOSED, true>(img, b, j, g, acc);
    } else {
        stomp_split8<false>(v, b);
        stomp_noise_product_kb<KB, FULL, TRANSPOSED, false>(img, b, j, g, acc);
    }
}

// acc[m] = rows 16 m .. 16 m + 15 of L * eps for the wave's rollout: draw / load, split and multiply, column block by column block
struct StompNoMid { __device__ __forceinline__ void operator()() const {} };
// `mid` runs between the two column blocks (the persistent kernel issues its exchange prefetch there: half the phase behind it
// for the partner to have published, half ahead of it to cover the round trip)
template <int DCH, int PRIO = STOMP_PRIO_NONE, typename MID = StompNoMid>
__device__ __forceinline__ void stomp_noise_bf16(const unsigned* __restrict__ img, f32x4 (&acc)[4], const float* __restrict__ eps_s,
                                                 int P, int p, int j, int g, uint32_t p_global, uint32_t s, uint32_t iter,
                                                 uint32_t seed_lo, uint32_t seed_hi, int level = 0, MID mid = MID()) {
    if (PRIO == STOMP_PRIO_STAGGER && eps_s == nullptr) stomp_setprio(level);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    uint32_t carry[2] = {0u, 0u};
    StompEps8 b;
    stomp_eps8<DCH, 0, PRIO>(v, carry, eps_s, P, p, j, g, p_global, s, iter, seed_lo, seed_hi);
    stomp_split_product<0>(img, v, b, eps_s != nullptr, j, g, acc);
    mid();
    stomp_eps8<DCH, 1, PRIO>(v, carry, eps_s, P, p, j, g, p_global, s, iter, seed_lo, seed_hi);
    stomp_split_product<1>(img, v, b, eps_s != nullptr, j, g, acc);
    if (PRIO == STOMP_PRIO_PROGRESS && eps_s == nullptr) stomp_setprio(0);
}

// TWO rollouts per product (2 DCH <= 16; round 4): the matrix instruction has 16 operand columns and a rollout fills DCH of them --
// at DCH = 7 the draws above run on 28 of the wave's 64 lanes.  Here column j < DCH is channel j of sample s0, column
// DCH <= j < 2 DCH channel j - DCH of sample s1: one wave draws and multiplies for two rollouts on 56 lanes, its partner wave
// skips the phase -- half the wave-instructions of the Philox / Box-Muller / split / matrix work per workgroup.  The counter
// words of a draw are the rollout's own (sample, channel, k-group), and a column of the product does not see the other
// columns: the samples are bit for bit those of stomp_noise_bf16.  eps_it: the pre-drawn normals of this iteration ((S, d, P,
// H), or nullptr), S their sample count (a sample index beyond it reads sample 0: never stored).
template <int DCH>
struct StompPairLane { int jj; uint32_t ss; bool act; };
template <int DCH>
__device__ __forceinline__ StompPairLane<DCH> stomp_pair_lane(int j, uint32_t s0, uint32_t s1) {
    static_assert(2 * DCH <= 16, "two rollouts must fit the 16 operand columns");
    const bool second = j >= DCH;
    return StompPairLane<DCH>{second ? j - DCH : j, second ? s1 : s0, j < 2 * DCH};
}
template <int DCH, int KB, int PRIO>
__device__ __forceinline__ void stomp_eps8_pair(float (&v)[8], uint32_t (&carry)[2], const float* __restrict__ eps_it, int P, int S, int p,
                                                const StompPairLane<DCH>& pl, int g, uint32_t p_global, uint32_t iter, uint32_t seed_lo,
                                                uint32_t seed_hi) {
    constexpr int H = 64;
    if (eps_it != nullptr) {
        const size_t sa = pl.ss < (uint32_t)S ? pl.ss : 0u;
        const f32x4* ep = reinterpret_cast<const f32x4*>(eps_it + ((sa * DCH + (size_t)pl.jj) * P + p) * H + 32 * KB + 8 * g);
        const f32x4 a = pl.act ? ep[0] : f32x4{0.f, 0.f, 0.f, 0.f}, b = pl.act ? ep[1] : f32x4{0.f, 0.f, 0.f, 0.f};
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        if (pl.act) {
            if (KB == 0) stomp_normals_lo<PRIO>(p_global, pl.ss, (uint32_t)pl.jj, (uint32_t)g, 0u, iter, seed_lo, seed_hi, v, carry);
            else stomp_normals_hi<PRIO>(p_global, pl.ss, (uint32_t)pl.jj, (uint32_t)g, 0u, iter, seed_lo, seed_hi, carry, v);
        }
    }
}
template <int DCH, int PRIO = STOMP_PRIO_NONE>
__device__ __forceinline__ void stomp_noise_bf16_pair(const unsigned* __restrict__ img, f32x4 (&acc)[4], const float* __restrict__ eps_it,
                                                      int P, int S, int p, int j, int g, uint32_t p_global, uint32_t s0, uint32_t s1,
                                                      uint32_t iter, uint32_t seed_lo, uint32_t seed_hi, int level = 0) {
    if (PRIO == STOMP_PRIO_STAGGER && eps_it == nullptr) stomp_setprio(level);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const StompPairLane<DCH> pl = stomp_pair_lane<DCH>(j, s0, s1);
    float v[8];
    uint32_t carry[2] = {0u, 0u};
    StompEps8 b;
    stomp_eps8_pair<DCH, 0, PRIO>(v, carry, eps_it, P, S, p, pl, g, p_global, iter, seed_lo, seed_hi);
    stomp_split_product<0>(img, v, b, eps_it != nullptr, j, g, acc);
    stomp_eps8_pair<DCH, 1, PRIO>(v, carry, eps_it, P, S, p, pl, g, p_global, iter, seed_lo, seed_hi);
    stomp_split_product<1>(img, v, b, eps_it != nullptr, j, g, acc);
    if (PRIO == STOMP_PRIO_PROGRESS && eps_it == nullptr) stomp_setprio(0);
}
// the D tiles of a paired product -> the two rollouts' LDS tiles (nt0: columns j < DCH, nt1: DCH <= j < 2 DCH; the columns
// beyond carry zeros into nt1's padding channels)
template <int DCH>
__device__ __forceinline__ void stomp_noise_to_tile_pair(float* __restrict__ nt0, float* __restrict__ nt1, const f32x4 (&acc)[4], int lane) {
    const int j = lane & 15, g = lane >> 4;
    float* nt = (j >= DCH) ? nt1 + (j - DCH) : nt0 + j;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) nt[(16 * m + 4 * g + rr) * NT_STRIDE] = acc[m][rr];
}

// D tiles (lane = channel) -> the wave's LDS tile [waypoint][channel] (stomp_noise_to_tile), read back as this lane's waypoint row
// (stomp_noise_row; lane = waypoint).  Written and read by the SAME wave (LDS operations of a wave complete in order: no barrier).
__device__ __forceinline__ void stomp_noise_to_tile(float* __restrict__ nt, const f32x4 (&acc)[4], int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) nt[(16 * m + 4 * g + rr) * NT_STRIDE + j] = acc[m][rr];
    __builtin_amdgcn_wave_barrier();
}
template <int DCH>
__device__ __forceinline__ void stomp_noise_row(const float* __restrict__ nt, int lane, float (&nz)[16]) {
    const f32x4* row = reinterpret_cast<const f32x4*>(nt + lane * NT_STRIDE);
#pragma unroll
    for (int v = 0; v < (DCH + 3) / 4; ++v) {
        const f32x4 t = row[v];
        nz[4 * v + 0] = t[0]; nz[4 * v + 1] = t[1]; nz[4 * v + 2] = t[2]; nz[4 * v + 3] = t[3];
    }
    __builtin_amdgcn_wave_barrier();
}
