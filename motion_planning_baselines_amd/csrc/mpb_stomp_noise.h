// mpb_stomp_noise.h -- the time-correlated STOMP noise  N = L * eps  of one rollout per wave on the matrix cores
// (shared by the two-kernel path, mpb_kernels.hip, and the persistent fused kernel, mpb_stomp_fused.hip).
//
// L (64 x 64 lower-triangular scale_tril of the precision matrix R) is the A operand of exact-fp32
// v_mfma_f32_16x16x4_f32 tiles, eps (64 x d standard normals) the B operand:
//   A  L[16m+i][4ks+g]   (lane i = l&15, g = l>>4)  from an LDS image laid out so that one ds_read_b128 per lane
//                         delivers four k-steps, conflict-free (stomp_l_image_index);
//   B  eps[c=j][4ks+g]   (lane j = l&15, g = l>>4)  generated in registers (Philox) or loaded;
//   lower-triangular: row tile m only needs k-steps ks <= 4m+3  ->  40 instead of 64 MFMAs.
#pragma once
#include "mpb_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NT_STRIDE 20  // floats per waypoint row of a wave's noise tile: 80 B keeps ds_read_b128 conflict-free

// Standard normals of the H = 64 paths: one Philox4x32-7 call per (particle, sample, channel j, k-group g, quarter q4)
// yields eps[j][k] for k = 16*q4 + 4*r + g, r = 0..3.  The counter holds the GLOBAL particle id, so the noise does not
// depend on how the particles are sharded over GPUs (nor on which kernel draws it).
__device__ __forceinline__ void stomp_eps4(uint32_t p_global, uint32_t s, uint32_t j, uint32_t g, uint32_t q4,
                                           uint32_t iter, uint32_t seed_lo, uint32_t seed_hi, float (&n)[4]) {
    const uint4 rr = philox4x32<7>(make_uint4(p_global, s, (j << 16) | (g << 8) | q4, iter), make_uint2(seed_lo, seed_hi));
    box_muller(rr.x, rr.y, n[0], n[1]);
    box_muller(rr.z, rr.w, n[2], n[3]);
}

// B operand of one rollout: e[ks] = eps[c = j][k = 4 ks + g], ks = 0..15 (zero for the padding channels j >= DCH).
// eps_rollout != nullptr: pre-drawn normals, laid out (d, P, H) for this sample (pointer already at [s]).
// PRIO: what the wave does to its issue priority while it draws (the caller resets it after the matrix product).
//   STOMP_PRIO_PROGRESS  lower it as the wave advances through the four draws (3, 2, 1, 0): the SIMD arbiter serves its
//       oldest wave first, which left alone makes the waves of a SIMD finish one after the other, the last one running
//       alone (see model_group_positions in mpb_geom.h).  For kernels whose waves do VALU work only.
//   STOMP_PRIO_STAGGER   one fixed level per wave, `level` = 0..3, different for the (up to four) waves of a SIMD: the
//       draws are VALU work, the product that follows them runs on the matrix pipe, and the two overlap only when the
//       waves of a SIMD are in DIFFERENT halves -- so here the waves are made to finish one after the other on purpose:
//       the wave with the highest level draws at full rate and multiplies while the others still draw.
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
template <int DCH, int PRIO = STOMP_PRIO_NONE>
__device__ __forceinline__ void stomp_b_operand(float (&e)[16], const float* __restrict__ eps_s, int P, int p, int j, int g,
                                                uint32_t p_global, uint32_t s, uint32_t iter, uint32_t seed_lo, uint32_t seed_hi,
                                                int level = 0) {
    constexpr int H = 64;
    if (eps_s != nullptr) {
        const float* ep = eps_s + ((size_t)(j < DCH ? j : 0) * P + p) * H + g;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) e[ks] = (j < DCH) ? ep[4 * ks] : 0.f;
    } else {
        if (PRIO == STOMP_PRIO_STAGGER) stomp_setprio(level);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float n[4] = {0.f, 0.f, 0.f, 0.f};
            if (PRIO == STOMP_PRIO_PROGRESS) stomp_setprio(3 - q4);
            if (j < DCH) stomp_eps4(p_global, s, (uint32_t)j, (uint32_t)g, (uint32_t)q4, iter, seed_lo, seed_hi, n);
            e[4 * q4 + 0] = n[0]; e[4 * q4 + 1] = n[1]; e[4 * q4 + 2] = n[2]; e[4 * q4 + 3] = n[3];
        }
    }
}

// index of L[row][col] in the permuted LDS image: Lp[(((m*4 + ks4)*4 + g)*16 + i)*4 + kk] = L[16m+i][4*(4*ks4+kk) + g]
__device__ __forceinline__ int stomp_l_image_index(int row, int col) {
    const int m = row >> 4, i = row & 15, ks = col >> 2, gq = col & 3;
    return ((((m * 4 + (ks >> 2)) * 4 + gq) * 16 + i) << 2) + (ks & 3);
}

// acc[m] = rows 16m..16m+15 of L * eps for the wave's rollout (lane (j, g) holds D[row = 4g + rr][col = j] in acc[m][rr])
__device__ __forceinline__ void stomp_noise_product(const float* __restrict__ Lp, const float (&e)[16], int j, int g, f32x4 (&acc)[4]) {
    const f32x4* Lp4 = reinterpret_cast<const f32x4*>(Lp);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks4 = 0; ks4 <= m; ++ks4) {
            const f32x4 a = Lp4[((m * 4 + ks4) * 4 + g) * 16 + j];
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], e[4 * ks4 + 0], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], e[4 * ks4 + 1], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], e[4 * ks4 + 2], acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], e[4 * ks4 + 3], acc[m], 0, 0, 0);
        }
    }
}

// D tiles (lane = channel) -> the wave's LDS tile [waypoint][channel] -> this lane's waypoint row (lane = waypoint).
// Written and read back by the SAME wave (LDS operations of a wave complete in order: no barrier).
template <int DCH>
__device__ __forceinline__ void stomp_noise_rows(float* __restrict__ nt, const f32x4 (&acc)[4], int lane, float (&nz)[16]) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) nt[(16 * m + 4 * g + rr) * NT_STRIDE + j] = acc[m][rr];
    __builtin_amdgcn_wave_barrier();
    const f32x4* row = reinterpret_cast<const f32x4*>(nt + lane * NT_STRIDE);
#pragma unroll
    for (int v = 0; v < (DCH + 3) / 4; ++v) {
        const f32x4 t = row[v];
        nz[4 * v + 0] = t[0]; nz[4 * v + 1] = t[1]; nz[4 * v + 2] = t[2]; nz[4 * v + 3] = t[3];
    }
    __builtin_amdgcn_wave_barrier();
}

// the two halves of stomp_noise_rows for callers that park the noise in the tile between producing and consuming it
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
