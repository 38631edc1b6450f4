// mpb_common.h -- error plumbing and small device helpers shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mpb.h"

// Wrong-result tuning switches (GP_T_*: by-elimination timing experiments on the GPMP2 solve) compile only with
// -DMPB_TUNING_BUILD.  build.build() -- the product build -- refuses that flag, build.build_variant() adds it by
// itself, and such a library says so in mpb_version() (bit 30), which _lib.lib() refuses unless it was asked for the
// variant explicitly (MPB_LIB_PATH).
#if !defined(MPB_TUNING_BUILD) && (defined(GP_T_GJ_STEPS) || defined(GP_T_NO_Z) || defined(GP_T_SKIP_STORE) || defined(GP_T_SKIP_SUBST) || \
                                   defined(FUSED_T_SKIP_NOISE) || defined(FUSED_T_SKIP_COST) || defined(FUSED_T_A_NOMEAN) || defined(FUSED_T_A_NOSTORE) || defined(LR_T_NOLOAD) || defined(LR_T_NOSTORE) || defined(LR_T_CAP_NOASM) || defined(LR_T_CAP_NOCHOL) || defined(LR_T_CAP_NOBACK) || defined(LR_T_CLK) || defined(FUSED_T_PRE_STATS) || defined(LR_T_GRAD_NOSTORE))
#error "a wrong-result tuning switch (GP_T_*) was defined without -DMPB_TUNING_BUILD"
#endif

// thread-local last-error string (defined in mpb_kernels.hip)
char* mpb_err_buf();
static inline int mpb_fail(int code, const char* msg) {
    snprintf(mpb_err_buf(), 512, "%s", msg);
    return code;
}
// the STOMP kernels read eps / L / Sigma / the means and write the samples as 16-byte vectors: a pointer the C-ABI is handed must
// be 16-byte aligned (a view at a 4-byte offset into an allocation would be misaligned dwordx4 accesses)
static inline bool mpb_misaligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                                    const void* e = nullptr, const void* f = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d) | ((uintptr_t)e) | ((uintptr_t)f)) & 15u) != 0;
}
static inline int mpb_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(mpb_err_buf(), 512, "%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return MPB_E_HIP;
    }
    return MPB_OK;
}

// Wave reductions on the DPP path (data-parallel primitives: the VALU reads a neighbouring lane directly) instead of
// ds_bpermute shuffles through the LDS crossbar: four dependent full-rate instructions bring every 16-lane row to its
// row total, the four row totals are read with v_readlane.  Callers are wave-uniform (all 64 lanes active).
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
#ifdef MPB_SHFL_REDUCE   // the former butterfly, kept for A/B measurements
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
#else
// debug builds (-DMPB_DEBUG): the DPP reductions read neighbouring lanes' registers directly, so every lane of the wave
// must be active at the call site (a disabled lane contributes a stale register, not a neutral element)
#ifdef MPB_DEBUG
#include <assert.h>
#define MPB_ASSERT_FULL_WAVE() assert(__builtin_amdgcn_read_exec() == ~0ull)
#else
#define MPB_ASSERT_FULL_WAVE()
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wave_sum_f32(float v) {
    MPB_ASSERT_FULL_WAVE();
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    MPB_ASSERT_FULL_WAVE();
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
// the same reductions over each ROW of sixteen lanes (every lane ends up with its row's result): for values replicated in the
// four rows, these are wave_sum_f32 / wave_max_f32 without the four v_readlane and the combine -- and, the rows holding the same
// values in the same lanes, the same bits as the full-wave forms give when rows 1-3 hold the neutral element
__device__ __forceinline__ float row_sum_f32(float v) {
    MPB_ASSERT_FULL_WAVE();
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}
__device__ __forceinline__ float row_max_f32(float v) {
    MPB_ASSERT_FULL_WAVE();
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return v;
}
// lane L of the caller's row, to every lane of the row (DPP row_newbcast, gfx90a and later): a v_readlane + scalar operand without
// the trip through a scalar register
template <int L>
__device__ __forceinline__ float row_bcast_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + L, 0xF, 0xF, false));
}
// acc = fma(lane k of the caller's row of `a`, b[k], acc) for k = 0 .. N - 1 in ascending order: v_fmac_f32 with its first source through
// DPP row_newbcast (written out: the compiler folds a DPP move into a multiply, not into the accumulating form).  ONE asm statement that
// opens with its own wait states: the hardware does not interlock a DPP read against a vector write of the source register in the two
// instructions before it, nor against a vector write of exec in the five before, and the compiler's hazard recogniser does not look
// into inline assembly -- a separate `s_nop` statement could be scheduled away from the first fmac, or a register copy / reload of `a`
// placed right in front of it.  `s_nop 4` = five wait states covers both rules whatever precedes the block; the fmacs
// themselves do not write `a`.  (tests/test_host_logic.py checks the disassembly: every row_newbcast:0 fmac follows an s_nop 4.)
template <int N>
__device__ __forceinline__ void fmac_row_bcast_seq(float& acc, float a, const float (&b)[N]) {
    static_assert(N == 8 || N == 16, "one asm block per supported length");
    if constexpr (N == 16) {
        asm(
            "s_nop 4\n\t"
            "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
            : "+v"(acc) : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]), "v"(b[12]), "v"(b[13]), "v"(b[14]), "v"(b[15]));
    } else {
        asm(
            "s_nop 4\n\t"
            "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
            : "+v"(acc) : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    }
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    MPB_ASSERT_FULL_WAVE();
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    const unsigned long long u = __double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * k);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * k);
        r[k] = __longlong_as_double(((unsigned long long)hi << 32) | lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}
#endif

// wave_sum_f32 for callers that need the total in ONE lane only: valid in LANE 63.  The four row totals meet through DPP row_bcast:15
// (lane 15 of a row into the next row; rows 1 and 3 take it) and row_bcast:31 (lane 31 into rows 2 and 3; row 3 takes it): (r2 + r3) +
// (r0 + r1) -- the association of wave_sum_f32, hence the same bits -- in two vector instructions where the four row totals cost four
// v_readlane, each a round trip through a scalar register the adds then wait for.
#ifndef MPB_SHFL_REDUCE
__device__ __forceinline__ float wave_sum_f32_lane63(float v) {
    MPB_ASSERT_FULL_WAVE();
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
    return v;
}
#else
__device__ __forceinline__ float wave_sum_f32_lane63(float v) { return wave_sum_f32(v); }
#endif

// A kernel argument RE-READ from the kernarg segment at its point of use (one s_load on the scalar-memory pipe) instead of being held in
// scalar registers from the kernel's entry on.  The big kernels of this library carry 20-30 arguments; the ones a phase touches once per
// iteration (output bases, a matrix used by one phase) are live across everything else, and under the 102-SGPR limit the compiler SPILLS such
// values into the lanes of a vector register -- every reload is then a v_readlane on the vector pipe that binds these kernels.  `off` is the
// argument's offset in the kernarg segment: offsetof in a mirror struct of the kernel's parameter list (explicit arguments are laid out in
// order at their natural alignment).  The empty asm makes the segment pointer opaque, so the load can be neither hoisted nor merged with
// another one: the value lives from here to its last use only.  The kernel must not ALSO use the parameter by name in a long-lived way.
template <class T>
__device__ __forceinline__ T kernarg_reload(unsigned off) {
    typedef const __attribute__((address_space(4))) char* ka_ptr;
    ka_ptr ka = (ka_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *(const __attribute__((address_space(4))) T*)(ka + off);
}
#define MPB_KARG(STRUCT, field) kernarg_reload<decltype(STRUCT::field)>((unsigned)offsetof(STRUCT, field))

// exp(x) and 1 / x on the hardware units, for the softmax algebra of the persistent STOMP kernels (round 5): v_exp_f32 of
// x log2(e) (2 instructions where ocml's expf is ~15; relative error <= 2^-22 + |x| 2^-24, i.e. 1e-6 at logits of -16, below
// which a weight is < 1e-7 of the largest) and v_rcp_f32 (1 ulp; the IEEE division hipcc emits is ~10 instructions).
__device__ __forceinline__ float fast_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

// Philox4x32-R (Salmon et al., SC'11), counter-based: no state, result depends only on (key, counter).
// R = 10 is the library default; R = 7 is the smallest round count the authors report as passing
// BigCrush ("Crush-resistant") and is what the STOMP kernel uses (the generator is ~15 % of its VALU work).
template <int ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one 32 x 32 -> 64 multiply per word pair (v_mad_u64_u32) instead of v_mul_hi_u32 + v_mul_lo_u32: both are
        // quarter-rate instructions, and the multiplies are most of the generator
        const uint64_t p0 = (uint64_t)M0 * ctr.x, p1 = (uint64_t)M1 * ctr.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) { return philox4x32<10>(ctr, key); }
// two uniforms -> two standard normals (Box-Muller on the hardware log2 / sin / cos units)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((a >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1]
    const float u2 = (b >> 8) * (1.0f / 16777216.0f);         // [0,1)
    // sqrt(-2 ln u1) on the raw v_sqrt_f32 (1 ulp): hipcc's IEEE-exact sqrtf expansion costs ~15 more instructions
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __log2f(u1));
    n0 = r * __builtin_amdgcn_cosf(u2);  // v_cos_f32 takes revolutions: cos(2*pi*u2)
    n1 = r * __builtin_amdgcn_sinf(u2);
}

// the same from the top 23 bits of each word dropped straight into the mantissa of a float in [1, 2) (round 4, as the STOMP
// draw does: mpb_stomp_noise.h): radius from u = 2 - m in (0, 1] (exact), the angle is m itself (v_sin / v_cos count in
// revolutions, period 1) -- no integer-to-float conversion, no scaling
__device__ __forceinline__ void box_muller_m23(uint32_t a, uint32_t b, float& n0, float& n1) {
#pragma clang fp contract(off)
    const float u1 = 2.0f - __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, a, 9));
    const float ang = __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, b, 9));
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = r * __builtin_amdgcn_cosf(ang);
    n1 = r * __builtin_amdgcn_sinf(ang);
}

// The first D floats of a waypoint row into q[0..D) (zero beyond): 8-byte loads when the row starts 8-byte aligned
// (`even` = the row stride in floats is even and so is the base: wave-uniform), else element by element.  A lane's row is
// 28-64 contiguous bytes, so every load instruction of a wave touches the same ~28 cache lines whatever its width:
// halving the instruction count halves the work of the memory pipe.
template <int MAXD>
__device__ __forceinline__ void load_row_prefix(const float* __restrict__ row, int D, bool even, float (&q)[MAXD]) {
    if (even) {
        const float2* r2 = reinterpret_cast<const float2*>(row);
#pragma unroll
        for (int i = 0; i < MAXD; i += 2) {
            float2 v = make_float2(0.f, 0.f);
            if (i + 1 < D) v = r2[i >> 1];
            else if (i < D) v.x = row[i];
            q[i] = v.x;
            if (i + 1 < MAXD) q[i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < MAXD; ++i) q[i] = (i < D) ? row[i] : 0.f;
    }
}
