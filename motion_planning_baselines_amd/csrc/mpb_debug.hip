// mpb_debug.hip -- test aids of the C-ABI: what the tests need to look INSIDE the product kernels' random-number path.
//   mpb_debug_philox          raw Philox4x32-R words (R = 7: the STOMP and MPPI kernels; R = 10: everything else) for given
//                             counters / keys -- compared with the published Random123 known-answer vectors;
//   mpb_debug_stomp_normals   the standard normals exactly as the STOMP kernels draw them (stomp_normals_lo / _hi: Philox4x32-7 +
//                             Box-Muller on the hardware log2 / sqrt / sin / cos units), laid out (iters, P, S, d, 64 ceil(H / 64))
//                             -- for the statistical tests of the throughput-mode noise (tests/test_gpu_rng.py).
//   mpb_debug_mppi_normals    the standard normals of the MPPI kernel's device draw, in the layout of its injected eps;
//   mpb_debug_occupy          workgroups that each hold a CU's LDS for a given time (the lost-launch tests).
// None is on a product path: this file is the ONLY source of libmpb_hip_debug.so (include/mpb_debug.h), a library of its
// own that the tests load next to the product library; it shares device code with the product through the headers only.
#include <hip/hip_runtime.h>

#include "mpb_common.h"
#include "mpb_stomp_noise.h"
#include "../../include/mpb_debug.h"

// (this library's own last-error buffer: mpb_common.h's helpers write through mpb_err_buf())
static thread_local char g_debug_err[512] = "";
char* mpb_err_buf() { return g_debug_err; }
extern "C" const char* mpb_debug_last_error(void) { return g_debug_err; }

__global__ void debug_philox_kernel(const uint32_t* __restrict__ ctr, const uint32_t* __restrict__ key, uint32_t* __restrict__ out,
                                    int n, int rounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 c = make_uint4(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]);
    const uint2 k = make_uint2(key[2 * i], key[2 * i + 1]);
    const uint4 r = (rounds == 7) ? philox4x32<7>(c, k) : philox4x32<10>(c, k);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

extern "C" int mpb_debug_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out, int n, int rounds, void* stream) {
    if (!ctr || !key || !out || n < 1) return mpb_fail(MPB_E_INVALID, "mpb_debug_philox: bad argument");
    if (rounds != 7 && rounds != 10) return mpb_fail(MPB_E_INVALID, "mpb_debug_philox: rounds must be 7 or 10");
    hipLaunchKernelGGL(debug_philox_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ctr, key, out, n, rounds);
    return mpb_check_launch("mpb_debug_philox");
}

// one thread per (iteration, particle, sample, channel j, k-group g): the sixteen normals of that lane of the STOMP kernels
// (stomp_normals_lo / _hi, mpb_stomp_noise.h: three Philox calls), eps[j][k = stomp_eps_column(g, u >> 2, u & 3)], u = 0..15
#ifdef MPB_DEBUG_RAW_NORMALS
#define DBGQ(x) (x)
#else
#define DBGQ(x) stomp_eps_quantise(x)
#endif
__global__ void debug_stomp_normals_kernel(float* __restrict__ out, int P, int S, int d, int HC, int n_iters, uint32_t seed_lo,
                                           uint32_t seed_hi, uint32_t iter0, uint32_t particle_offset) {
    const size_t n = (size_t)n_iters * P * S * d * HC * 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = i & 3;
    size_t r = i >> 2;
    const uint32_t kc = r % HC; r /= HC;              // 64-column chunk of the horizon (the chunked kernels: call0 = kc << 4)
    const uint32_t j = r % d; r /= d;
    const uint32_t s = r % S; r /= S;
    const uint32_t p = r % P; r /= P;
    const uint32_t it = (uint32_t)r;
    float lo[8], hi[8];
    uint32_t carry[2];
    stomp_normals_lo<STOMP_PRIO_NONE>(particle_offset + p, s, j, g, kc << 4, iter0 + it, seed_lo, seed_hi, lo, carry);
    stomp_normals_hi<STOMP_PRIO_NONE>(particle_offset + p, s, j, g, kc << 4, iter0 + it, seed_lo, seed_hi, carry, hi);
    float* o = out + (((((size_t)it * P + p) * S + s) * d + j) * HC + kc) * 64;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        // (as they enter the product: the two leading bf16 components, mpb_stomp_noise.h)
        o[stomp_eps_column((int)g, u >> 2, u & 3)] = DBGQ(lo[u]);
        o[stomp_eps_column((int)g, 2 + (u >> 2), u & 3)] = DBGQ(hi[u]);
    }
}

extern "C" int mpb_debug_stomp_normals_h(float* out, int P, int S, int d, int H, int n_iters, uint64_t seed, uint32_t iter0,
                                         uint32_t particle_offset, void* stream) {
    if (!out || P < 1 || S < 1 || d < 1 || d > 16 || n_iters < 1 || H < 1 || H > 256)
        return mpb_fail(MPB_E_INVALID, "mpb_debug_stomp_normals: bad argument");
    const int HC = (H + 63) / 64;
    const size_t n = (size_t)n_iters * P * S * d * HC * 4;
    if (n > 0x7FFFFFFFull * 256ull) return mpb_fail(MPB_E_INVALID, "mpb_debug_stomp_normals: too many draws for one launch");
    hipLaunchKernelGGL(debug_stomp_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, P, S,
                       d, HC, n_iters, (uint32_t)seed, (uint32_t)(seed >> 32), iter0, particle_offset);
    return mpb_check_launch("mpb_debug_stomp_normals");
}
extern "C" int mpb_debug_stomp_normals(float* out, int P, int S, int d, int n_iters, uint64_t seed, uint32_t iter0,
                                       uint32_t particle_offset, void* stream) {
    return mpb_debug_stomp_normals_h(out, P, S, d, 64, n_iters, seed, iter0, particle_offset, stream);
}

// the standard normals of the MPPI kernel's device draw (csrc/mpb_mppi.hip: one Philox4x32-7 call per (problem, sample,
// group of four time steps | control dimension << 16, iteration) -> four normals by box_muller_m23), laid out as the injected
// eps of mpb_mppi_step: (n_iters, NP, c, S, T)
__global__ void debug_mppi_normals_kernel(float* __restrict__ out, int NP, int S, int T, int c, int n_iters, uint32_t seed_lo,
                                          uint32_t seed_hi, uint32_t iter0) {
    const int G4 = (T + 3) >> 2;
    const size_t n = (size_t)n_iters * NP * c * S * G4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    size_t r = idx;
    const uint32_t g4 = r % G4; r /= G4;
    const uint32_t s = r % S; r /= S;
    const uint32_t i = r % c; r /= c;
    const uint32_t prob = r % NP; r /= NP;
    const uint32_t it = (uint32_t)r;
    const uint4 w = philox4x32<7>(make_uint4(prob, s, g4 | (i << 16), iter0 + it), make_uint2(seed_lo, seed_hi));
    float nrm[4];
    box_muller_m23(w.x, w.y, nrm[0], nrm[1]);
    box_muller_m23(w.z, w.w, nrm[2], nrm[3]);
    float* o = out + ((((size_t)it * NP + prob) * c + i) * S + s) * T;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if ((int)(4 * g4 + q) < T) o[4 * g4 + q] = nrm[q];
}

extern "C" int mpb_debug_mppi_normals(float* out, int NP, int S, int T, int c, int n_iters, uint64_t seed, uint32_t iter0, void* stream) {
    if (!out || NP < 1 || S < 1 || T < 1 || c < 1 || n_iters < 1) return mpb_fail(MPB_E_INVALID, "mpb_debug_mppi_normals: bad argument");
    const size_t n = (size_t)n_iters * NP * c * S * ((T + 3) >> 2);
    if (n > 0x7FFFFFFFull * 256ull) return mpb_fail(MPB_E_INVALID, "mpb_debug_mppi_normals: too many draws for one launch");
    hipLaunchKernelGGL(debug_mppi_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, NP, S, T, c,
                       n_iters, (uint32_t)seed, (uint32_t)(seed >> 32), iter0);
    return mpb_check_launch("mpb_debug_mppi_normals");
}

// test aid (mpb_debug_occupy): workgroups that each take a whole CU's LDS and spin for a given time -- the "other stream
// keeps the chip busy" of the time-out tests
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    __shared__ unsigned pad[150 * 256];                       // 150 KB: one such workgroup per CU
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (pad[(threadIdx.x + 1) & 63] == 0xFFFFFFFFu) sink[0] = 1u;   // (keeps the array)
}

extern "C" int mpb_debug_occupy(int n_blocks, uint64_t usec, uint32_t* sink, void* stream) {
    if (n_blocks < 1 || !sink) return mpb_fail(MPB_E_INVALID, "mpb_debug_occupy: bad argument");
    hipLaunchKernelGGL(occupy_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, usec * 100ull, sink);
    return mpb_check_launch("mpb_debug_occupy");
}
