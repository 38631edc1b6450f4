/* mpb_debug.h -- TEST AIDS, not the product ABI.

   Exported by a separate library, motion_planning_baselines_amd/csrc/libmpb_hip_debug.so (csrc/mpb_debug.hip), which the
   tests load next to libmpb_hip.so; the product library exports none of these.  They look INSIDE the product kernels'
   random-number path (same device functions, csrc/mpb_stomp_noise.h) and provide the "another stream keeps the chip
   busy" load of the lost-launch tests. */
#ifndef MPB_DEBUG_H
#define MPB_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *mpb_debug_last_error(void);
/* Test aid: n_blocks workgroups that each take a whole CU's LDS and idle for `usec` microseconds (the "another stream
 * keeps the chip busy" of the time-out tests); `sink` is one device word (never written in practice). */
int mpb_debug_occupy(int n_blocks, uint64_t usec, uint32_t *sink, void *stream);
/* Test aids for the random-number path (csrc/mpb_debug.hip; not on a product path).
 * mpb_debug_philox: out[4i..4i+3] = Philox4x32-`rounds`(ctr[4i..4i+3], key[2i..2i+1]), rounds = 7 (the STOMP kernels and, since
 * ABI 4, the MPPI kernel) or 10 (every other kernel), device pointers -- for the Random123 known-answer vectors.
 * mpb_debug_stomp_normals: the standard normals of n_iters STOMP iterations exactly as mpb_stomp_step / mpb_stomp_run draw
 * them in throughput mode (eps == NULL): out (n_iters, P, S, d, 64), element [it][p][s][c][k] = eps of iteration
 * iter0 + it, global particle particle_offset + p, sample s, channel c, waypoint k.
 * mpb_debug_stomp_normals_h: the same for any horizon H <= 256 (the chunked kernels draw 64 columns per chunk kc = k / 64):
 * out (n_iters, P, S, d, 64 * ceil(H / 64)); the columns k >= H are drawn by the kernels too and meet zero columns of L. */
int mpb_debug_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, int n, int rounds, void *stream);
int mpb_debug_stomp_normals(float *out, int P, int S, int d, int n_iters, uint64_t seed, uint32_t iter0,
                            uint32_t particle_offset, void *stream);
int mpb_debug_stomp_normals_h(float *out, int P, int S, int d, int H, int n_iters, uint64_t seed, uint32_t iter0,
                              uint32_t particle_offset, void *stream);
/* mpb_debug_mppi_normals: the standard normals mpb_mppi_step draws in throughput mode (eps == NULL), laid out as its injected
 * eps: out (n_iters, NP, c, S, T), element [it][problem][dim][s][t] -- Philox4x32-7, counter (problem, s, (t / 4) | dim << 16,
 * iter0 + it), Box-Muller on 23-bit uniforms (csrc/mpb_common.h box_muller_m23). */
int mpb_debug_mppi_normals(float *out, int NP, int S, int T, int c, int n_iters, uint64_t seed, uint32_t iter0, void *stream);

#ifdef __cplusplus
}
#endif
#endif
