// mpb_mppi.hip -- MPPI on point-particle dynamics: one workgroup per problem, one WAVE per control sample,
// one LANE per time step.
//
// Replaces MPPI.optimize's loop body (mppi.py:145-152): ControlTrajectoryGaussian.sample
// (priors/gaussian.py:276-298: per control dim, mean + scale_tril @ eps), the Euler rollout
// (mppi.py:205-209 over PointParticleDynamics.dynamics, point.py:102-140, deterministic), traj_cost
// (point.py:154-226 incl. quirks Q6 / Q8), the importance-sampling term (mppi.py:125-128) and
// update_controller (mppi.py:72-86).  All `n_iters` iterations run inside one launch.
//
// The reference walks the horizon in a Python loop (one dynamics + cost call per step).  Velocity-controlled
// point dynamics are x_{t+1} = x_t + clamp(u_t) dt, i.e. a prefix sum over time, so here the horizon is the
// lane axis: the T x T noise product runs on the matrix pipe for all samples at once (T <= 64; beyond that T fma per
// lane against an LDS-resident transposed scale_tril: see MPPI_NOISE_*), the rollout is a wave scan, the per-step costs (incl. the collision field) are evaluated by all lanes at once
// and reduced with DPP; the sequential depth per iteration drops from O(T * (T + cost)) to O(T + log T).
//
// Only velocity control is served (state_dim == control_dim): with control_type='acceleration' the
// reference's dynamics slices an empty tensor (point.py:114-118 uses the doubled self.state_dim) and
// cannot run, so there is nothing to match.
#include "mpb_common.h"
#include "mpb_geom.h"
#include "mpb_stomp_noise.h"   // the permuted MFMA image of a 64 x 64 lower-triangular factor (stomp_l_image_index)

#define MPPI_MAX_C 4
#define MPPI_E_STRIDE 68   // words between the samples of the normals slab: 64 + 4, so that the 16-byte operand reads of the lanes of a tile row land on different banks

__device__ __forceinline__ float block_sum(float v, float* red, int lane, int wave, int nw) {
    v = wave_sum_f32(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red, int lane, int wave, int nw) {
    v = wave_max_f32(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// inclusive prefix sum over the 64 lanes on the DPP path: Hillis-Steele inside each 16-lane row (row_shr 1, 2, 4, 8,
// zeros shifted in), then the totals of the earlier rows (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3)
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
    (void)lane;
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return v;
}

struct MppiLds {
    float* wvec;   // c*T   : Cov_inv[i] @ mean[:, i]
    float* red;    // 64    : scratch of the block reductions
    float* wts;    // S     : sample weights
    float* cst;    // S     : sample costs
    float* coll;   // S     : collision cost of every sample (summed into the Q6 scalar in sample order)
    float* Us;     // S*c*T : controls of every sample, [s][i][t]
    float* epsw;   // W*c*T : standard normals of the sample a wave is working on, [wave][i][t]
    float* trilT;  // c*T*T : scale_tril transposed [i][k][t] (only when it fits); matrix path: c*4096, the MFMA images
    float* E;      // matrix path: c*Spad*MPPI_E_STRIDE standard normals of EVERY sample, [i][s][k & 3][k >> 2] (zero beyond T / S)
    unsigned* gridw;   // grid path (ONE grid-backed collision field): the broad-phase grid, grid_words words ...
    float4* otab;      // ... and the obstacle table (n_sph + 1 entries, the last one the far dummy)
};

// How U = mean + scale_tril @ eps is formed (mpb_mppi_step picks):
//   MPPI_NOISE_MATRIX  T <= 64 and everything fits LDS: the normals of all samples are drawn first, then the c products
//                      (T x T lower triangular) x (T x S) run as v_mfma_f32_16x16x4_f32 tiles, one 16 x 16 tile of
//                      (time steps x samples) per wave -- every word of the factor leaves LDS once per tile instead of once
//                      per (sample, lane, k): the per-lane form below spends two LDS reads per multiply-add, ~420 LDS
//                      instructions per wave and iteration at S = 32, T = 64, c = 2, and was LDS-bound;
//   MPPI_NOISE_LDS     the per-lane product against the transposed factor in LDS;
//   MPPI_NOISE_GLOBAL  the same against the factor in global memory (it does not fit).
#define MPPI_NOISE_GLOBAL 0
#define MPPI_NOISE_LDS 1
#define MPPI_NOISE_MATRIX 2

// CC: the control dimension as a compile-time constant (2: the reference example's point mass; 0: run-time c <= MPPI_MAX_C);
// GRID: collision through the broad-phase grid (ONE grid-backed field) -- the exhaustive evaluator, with its blocks of obstacles
// in scalar registers, lives in the other instantiation (one kernel holding both spilled 48 VGPRs and 292 SGPRs)
// MATRIX: the noise product of all samples on the matrix pipe (MPPI_NOISE_MATRIX) as a compile-time fact -- the instantiation then
// carries neither the per-lane product nor its operands (round 5: the kernel spilled 155 SGPRs, a fifth of its vector
// instructions were v_readlane / v_writelane of spilled scalars)
// INJ: injected normals supported (eps != NULL); the device-noise instantiation does not carry that path.
// mirror of mppi_kernel's parameter list: the offsets of its arguments in the kernarg segment (mpb_common.h, kernarg_reload)
struct MppiKernargs {
    float* mean; const float* eps; const float* tril; const float* cov_inv; const float* state0; const float* goal;
    const float* ctrl_min; const float* ctrl_max; const float* discount; const float* cw; const float* geom; float* controls;
    float* states; float* costs; float* weights; float* best_cost; float* best_states; int S, T, c_rt;
    float dt, k_sigma, weight, temp, step_size; int n_iters; uint32_t seed_lo, seed_hi, iter0; int noise_mode, grid_words;
};
static_assert(sizeof(MppiKernargs) == 17 * 8 + 14 * 4, "mirror of the kernel's explicit arguments");
#define MPPI_KARG(field) MPB_KARG(MppiKernargs, field)

template <int CC, bool GRID, bool MATRIX, bool INJ>
__global__ __launch_bounds__(1024) void mppi_kernel(
    float* __restrict__ mean, const float* __restrict__ eps, const float* __restrict__ tril,
    const float* __restrict__ cov_inv, const float* __restrict__ state0, const float* __restrict__ goal,
    const float* __restrict__ ctrl_min, const float* __restrict__ ctrl_max, const float* __restrict__ discount,
    const float* __restrict__ cw, const float* __restrict__ geom, float* __restrict__ controls,
    float* __restrict__ states, float* __restrict__ costs, float* __restrict__ weights,
    float* __restrict__ best_cost, float* __restrict__ best_states, int S, int T, int c_rt,
    float dt, float k_sigma, float weight, float temp, float step_size, int n_iters, uint32_t seed_lo,
    uint32_t seed_hi, uint32_t iter0, int noise_mode, int grid_words) {
    extern __shared__ float lds[];
    if (!INJ) eps = nullptr;
    const int c = CC ? CC : c_rt;
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    MppiLds M;
    // the mean of the problem lives in LDS for the whole launch (round 5: it was read from / written to global memory in three
    // phases of every iteration, each a round trip on the iteration's critical path) and goes back once at the end
    float* m = lds;
    M.wvec = lds + ((c * T + 3) & ~3);
    M.red = M.wvec + c * T;
    M.wts = M.red + 64;
    M.cst = M.wts + ((S + 3) & ~3);
    M.coll = M.cst + ((S + 3) & ~3);
    M.Us = M.coll + ((S + 3) & ~3);
    M.epsw = M.Us + (size_t)S * (c * T + 4);
    M.trilT = M.epsw + (size_t)nw * c * T;
    const bool matrix = MATRIX;                        // (the launcher picks the instantiation by noise_mode)
    const bool tril_in_lds = !MATRIX && noise_mode == MPPI_NOISE_LDS;
    const int us_stride = c * T + 4;                   // words between the samples of the controls slab (see the matrix product)
    const int Spad = (S + 15) & ~15;
    if (matrix) {                      // (no per-wave normals: the slab holds every sample's)
        M.trilT = lds + (((M.epsw - lds) + 3) & ~(ptrdiff_t)3);                       // 16-byte rows for ds_read_b128
        M.E = M.trilT + (size_t)c * 4096;
    }
    // collision through the broad-phase grid: the launcher reserved grid_words + 4 (MPB_GRID_MAX_SPH + 1) words behind
    // everything else when geom_flags promised ONE grid-backed field; the device header has the last word
    GeomView G0;
    bool use_grid = false;
    if (geom != nullptr) {
        G0 = geom_view(geom);
        use_grid = GRID && grid_words > 0 && G0.next == 0 && G0.kind == MPB_KIND_POINT && grid_usable(G0) && G0.n_cells <= grid_words;
        if (use_grid) {
            float* tail = matrix ? M.E + (size_t)c * Spad * MPPI_E_STRIDE : (tril_in_lds ? M.trilT + (size_t)c * T * T : M.trilT);
            M.gridw = reinterpret_cast<unsigned*>(lds + (((tail - lds) + 3) & ~(ptrdiff_t)3));
            M.otab = reinterpret_cast<float4*>(M.gridw + ((grid_words + 3) & ~3));
            grid_stage_offsets(G0, M.gridw, M.otab, threadIdx.x, blockDim.x);     // (offset words: mpb_geom.h)
        }
    }
    // what the collision walk of a step needs of the field, formed ONCE (round 5: grid_addr() inside the step loop moved the three
    // cell sizes into vector registers on every trip, and the view's header fields stayed live -- spilled -- across the whole loop)
    GridAddr GA{};
    float rl_m = 0.f, fscale0 = 0.f;
    bool z_on = false;
    if (GRID && use_grid) {
        GA = grid_addr(G0);
        rl_m = G0.margin + G0.links[4];        // margin + r_l: spheres_hinge_grid's RLM form, the same association
        fscale0 = G0.fscale;
        z_on = G0.n_dof > 2;
    }
    const int prob = blockIdx.x;
    float* mean_g = mean + (size_t)prob * T * c;
    for (int e = threadIdx.x; e < T * c; e += blockDim.x) m[e] = mean_g[e];
    const float w_pos = cw[0], w_ctrl = cw[2], w_posT = cw[3];
    if (tril_in_lds) {
        for (int e = threadIdx.x; e < c * T * T; e += blockDim.x) {       // coalesced read, transposed write
            const int i = e / (T * T), r = e - i * T * T, t = r / T, k = r - t * T;
            M.trilT[((size_t)i * T + k) * T + t] = tril[e];
        }
    }
    if (matrix) {
        for (int e = threadIdx.x; e < c * 4096; e += blockDim.x) {
            const int i = e >> 12, row = (e >> 6) & 63, col = e & 63;
            M.trilT[i * 4096 + stomp_l_image_index(row, col)] = (row < T && col < T) ? tril[((size_t)i * T + row) * T + col] : 0.f;
        }
        for (int e = threadIdx.x; e < c * Spad * MPPI_E_STRIDE; e += blockDim.x) M.E[e] = 0.f;      // padding samples / steps stay zero
    }
    float gl[MPPI_MAX_C], x0[MPPI_MAX_C], umin[MPPI_MAX_C], umax[MPPI_MAX_C];
#pragma unroll
    for (int i = 0; i < MPPI_MAX_C; ++i) {
        const bool on = i < c;
        gl[i] = on ? goal[(size_t)prob * c + i] : 0.f;
        x0[i] = on ? state0[(size_t)prob * c + i] : 0.f;
        umin[i] = on ? ctrl_min[i] : 0.f;
        umax[i] = on ? ctrl_max[i] : 0.f;
    }
    float* ew = M.epsw + (size_t)wave * c * T;
    const float inv_temp = 1.0f / temp;
    const float dsc0 = (lane < T) ? discount[lane] : 0.f;        // the discount of this lane's step in the first 64-step chunk
    const bool has_best = best_cost != nullptr;
    float best_c = has_best ? best_cost[prob] : 0.f;   // running best over all iterations (and calls)

    for (int it = 0; it < n_iters; ++it) {
        const bool last = it == n_iters - 1;
        __syncthreads();
        // ---- w_i = Cov_inv[i] @ mean_i (vector of the importance-sampling term)
        // eight lanes per output row: each reads a contiguous eighth of the row (coalesced), the partial sums meet in
        // a three-step lane exchange.  (One thread per row walked the row with T dependent, uncoalesced global loads.)
        for (int i = 0; i < c; ++i)
        for (int t0 = 0; t0 < T; t0 += blockDim.x >> 3) {
            const int t = t0 + (threadIdx.x >> 3), seg = threadIdx.x & 7;
            const int e = i * T + t;
            float a = 0.f;
            if (t < T) {
                const float* row = MPPI_KARG(cov_inv) + ((size_t)i * T + t) * T;     // (re-read per phase: kernarg_reload)
                for (int k = seg; k < T; k += 8) a = fmaf(row[k], m[k * c + i], a);
            }
            // (the eight partial sums meet on the DPP path: quad_perm [1,0,3,2], [2,3,0,1], then row_half_mirror brings the other
            // quad of the eight-lane group -- the association of the xor-1, -2, -4 butterfly, without its three trips through
            // the LDS crossbar)
            a += dpp_f32<0xB1>(a);
            a += dpp_f32<0x4E>(a);
            a += dpp_f32<0x141>(a);
            if (t < T && seg == 0) M.wvec[e] = a;
        }
        // (matrix path: the draw below does not need w; the barriers behind the draw and behind the product order it before the rollouts)
        if (!matrix) __syncthreads();
        if (matrix) {
            // ---- the standard normals of EVERY sample (the same stream as below: one Philox call per (sample, dimension,
            //      group of four steps)), then U = mean + L eps on the matrix pipe, one (16 steps x 16 samples) tile per wave
            const int G4 = (T + 3) >> 2;
            if (eps != nullptr) {
                for (int e = threadIdx.x; e < c * S * T; e += blockDim.x) {
                    const int i = e / (S * T), r = e - i * S * T, ss = r / T, t = r - ss * T;
                    M.E[((size_t)i * Spad + ss) * MPPI_E_STRIDE + (t & 3) * 16 + (t >> 2)] =
                        eps[((((size_t)it * gridDim.x + prob) * c + i) * S + ss) * T + t];
                }
            } else {
                // (which thread draws which (sample, group of four steps): groups fastest only over FOUR values, then eight
                // samples -- 32 consecutive lanes then write banks 4 ss + g4 = 0..31 of the slab (its sample stride is 4 mod 32),
                // where groups fastest over all sixteen put two samples on the same banks: the 2-way conflict that was left
                // of round 3's 16-way one.  The counters do not depend on the lane: same stream)
                // (round 5: a WAVE takes a (dimension, high group) pair -- wave-uniform, scalar arithmetic -- and its lanes run over
                // (sample, low group): the flat index of round 4 cost four integer divisions by run-time values per draw, ~100
                // vector instructions where the draw itself is ~110)
                const int G4q = (G4 + 3) >> 2;
                const int wave_s = __builtin_amdgcn_readfirstlane(wave);
                for (int pr = wave_s; pr < c * G4q; pr += nw) {
                    const int i = pr / G4q, ghi = pr - i * G4q;              // (scalar)
                for (int ql = lane; ql < 4 * S; ql += 64) {
                    const int glo = ql & 3, ss = ql >> 2;
                    const int g4 = 4 * ghi + glo;
                    if (g4 >= G4) continue;
                    const uint4 rr = philox4x32<7>(make_uint4((uint32_t)prob, (uint32_t)ss, (uint32_t)g4 | ((uint32_t)i << 16),
                                                              iter0 + (uint32_t)it),
                                                   make_uint2(seed_lo, seed_hi));
                    float n[4];
                    box_muller_m23(rr.x, rr.y, n[0], n[1]);
                    box_muller_m23(rr.z, rr.w, n[2], n[3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * g4 + q < T) M.E[((size_t)i * Spad + ss) * MPPI_E_STRIDE + q * 16 + g4] = n[q];
                }
                }
            }
            __syncthreads();
            const int NT = Spad >> 4, j = lane & 15, g = lane >> 4;
            const f32x4* L4 = reinterpret_cast<const f32x4*>(M.trilT);
            // (tile = (i * 4 + mt) * NT + nt, every nw-th one this wave's: walked with scalar counters, no division)
            const int wave_t = __builtin_amdgcn_readfirstlane(wave);
            int tile_idx = 0, tile_next = wave_t;
            for (int i = 0; i < c; ++i)
            for (int mt = 0; mt < 4; ++mt)
            for (int nt = 0; nt < NT; ++nt) {
                if (tile_idx++ != tile_next) continue;
                tile_next += nw;
                if (16 * mt >= T) continue;                                              // (wave-uniform)
                const f32x4* e4 = reinterpret_cast<const f32x4*>(M.E + ((size_t)i * Spad + 16 * nt + j) * MPPI_E_STRIDE + g * 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int ks4 = 0; ks4 <= mt; ++ks4) {
                    const f32x4 a = L4[i * 1024 + ((mt * 4 + ks4) * 4 + g) * 16 + j];
                    const f32x4 b = e4[ks4];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], a[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[1], a[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], a[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[3], a[3], acc, 0, 0, 0);
                }
                // issued TRANSPOSED (A = eps, B = L^T: same operand registers): acc[r] = (L eps)[t = 16 mt + j][sample 16 nt + 4 g + r]
                // -- the 16 lanes of a row write 16 consecutive steps of one sample, and the sample stride of the slab
                // (c T + 4 words) puts the four rows on different banks: conflict-free (the untransposed form wrote 16
                // samples at the same step: a 16-way conflict, 66 % of the kernel's LDS cycles in round 3)
                const int t = 16 * mt + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ss = 16 * nt + 4 * g + q;
                    if (ss < S && t < T) M.Us[(size_t)ss * us_stride + i * T + t] = m[t * c + i] + acc[q];
                }
            }
            __syncthreads();
        }
        for (int s = wave; s < S; s += nw) {
            // ---- standard normals of sample s: injected (reference draw order (c, S, T)) or Philox
            if (matrix) {
            } else if (eps != nullptr) {
                for (int i = 0; i < c; ++i)
                    for (int t = lane; t < T; t += 64)
                        ew[i * T + t] = eps[((((size_t)it * gridDim.x + prob) * c + i) * S + s) * T + t];
            } else {
                // one Philox call yields the normals of four consecutive time steps of one control dimension: the
                // c * ceil(T/4) calls of the sample are spread over the lanes (counter = (problem, sample,
                // step group | dim << 16, iteration), the same stream as one call per lane and step would give)
                const int G4 = (T + 3) >> 2;
                for (int l = lane; l < c * G4; l += 64) {
                    const int i = l / G4, g4 = l - i * G4;
                    const uint4 r = philox4x32<7>(make_uint4((uint32_t)prob, (uint32_t)s, (uint32_t)g4 | ((uint32_t)i << 16),
                                                             iter0 + (uint32_t)it),
                                                  make_uint2(seed_lo, seed_hi));
                    float n[4];
                    box_muller_m23(r.x, r.y, n[0], n[1]);
                    box_muller_m23(r.z, r.w, n[2], n[3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (4 * g4 + q < T) ew[i * T + 4 * g4 + q] = n[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float carry[MPPI_MAX_C];
#pragma unroll
            for (int i = 0; i < MPPI_MAX_C; ++i) carry[i] = 0.f;
            float pos_l = 0.f, ctl_l = 0.f, is_l = 0.f, term_l = 0.f, coll_l = 0.f;
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                const bool on = t < T;
                const int kend = min(base + 64, T);            // entries above the diagonal are zero: no lane mask
                float u[MPPI_MAX_C], x[MPPI_MAX_C];
#pragma unroll
                for (int i = 0; i < MPPI_MAX_C; ++i) {
                    u[i] = 0.f;
                    if (matrix) {
                        if (i < c && on) u[i] = M.Us[(size_t)s * us_stride + i * T + t];
                    } else if (i < c && on) {
                        // U = mean + L eps (gaussian.py:276-298), ascending k like the matmul row
                        float a = 0.f;
                        if (tril_in_lds) {
                            // eight LDS reads in flight per trip, then the eight fma in ascending k (the order is the
                            // matmul row's; only the load latency is paid once per trip instead of once per term)
                            const float* col = M.trilT + (size_t)i * T * T + t;
                            const float* er = ew + i * T;
                            int k = 0;
                            for (; k + 8 <= kend; k += 8) {
                                float cv[8], ev[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    cv[u] = col[(size_t)(k + u) * T];
                                    ev[u] = er[k + u];
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) a = fmaf(cv[u], ev[u], a);
                            }
                            for (; k < kend; ++k) a = fmaf(col[(size_t)k * T], er[k], a);
                        } else {
                            const float* row = tril + ((size_t)i * T + t) * T;
                            for (int k = 0; k < kend; ++k) a = fmaf(row[k], ew[i * T + k], a);
                        }
                        u[i] = m[t * c + i] + a;
                        M.Us[(size_t)s * us_stride + i * T + t] = u[i];
                    }
                }
                // ---- Euler rollout (mppi.py:205-209, point.py:112, :139): x_t = x_0 + sum_{k<t} clamp(u_k) dt
#pragma unroll
                for (int i = 0; i < MPPI_MAX_C; ++i) {
                    const float v = (i < c && on) ? fminf(fmaxf(u[i], umin[i]), umax[i]) * dt : 0.f;
                    const float inc = wave_scan_incl(v, lane);
                    x[i] = x0[i] + (carry[i] + (inc - v));
                    carry[i] += readlane_f32(inc, 63);
                }
                if (on) {
                    // ---- quadratic cost terms of step t (point.py:198-226)
                    float pc = 0.f, cc = 0.f, tc = 0.f, is = 0.f;
#pragma unroll
                    for (int i = 0; i < MPPI_MAX_C; ++i) {
                        if (i < c) {
                            const float dx = x[i] - gl[i];
                            pc += dx * dx * w_pos;
                            tc += dx * dx * w_posT;
                            cc += u[i] * u[i] * w_ctrl;
                            is = fmaf(u[i], M.wvec[i * T + t], is);
                        }
                    }
                    const float dsc = (base == 0) ? dsc0 : MPPI_KARG(discount)[t];
                    pos_l += pc * dsc;
                    ctl_l += cc * dsc;
                    is_l += is;
                    if (t == T - 1) term_l = tc * dsc;
                    if (geom != nullptr && t >= 1) {
                        float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
#pragma unroll
                        for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < MPPI_MAX_C && i < c) ? x[i < MPPI_MAX_C ? i : 0] : 0.f;
                        // (same bits either way: the grid only culls, tests/test_gpu_parity_gpmp2_mppi.py)
                        if (GRID) {
                            // (the launcher picked this instantiation from geom_flags; a device header that disagrees poisons
                            // the cost instead of being mis-read)
                            if (use_grid) {      // the point branch of waypoint_cost_grid (mpb_geom.h), same expressions
                                const float px[1] = {q[0]}, py[1] = {q[1]}, pz[1] = {z_on ? q[2] : 0.f}, rl[1] = {rl_m};
                                float cg = 0.f;
                                spheres_hinge_grid<1, true, false, true>(G0, M.gridw, M.otab, px, py, pz, rl, cg, GA);
                                coll_l += fscale0 * cg;
                            } else {
                                coll_l += __uint_as_float(0x7FC00000u);
                            }
                        } else {
                            coll_l += waypoint_cost_chain<false>(geom, q, dq);
                        }
                    }
                    if (last) {                                 // API-visible outputs of the last iteration
                        float* Ug = MPPI_KARG(controls) + (((size_t)prob * S + s) * T + t) * c;
                        float* Xg = MPPI_KARG(states) + (((size_t)prob * S + s) * T + t) * c;   // velocity control: state_dim == c
#pragma unroll
                        for (int i = 0; i < MPPI_MAX_C; ++i)
                            if (i < c) {
                                Ug[i] = u[i];
                                Xg[i] = x[i];
                            }
                    }
                }
            }
            // the sample's cost = position + (velocity: empty slice, quirk Q8) + control + terminal + temp * importance term
            // (point.py:198-226, mppi.py:125-128): the four per-step contributions are added PER LANE and reduced once (round 5:
            // four wave reductions of ~15 vector instructions each were a tenth of the kernel's vector work); the collision
            // part stays separate (quirk Q6 sums it over the samples)
            // (the totals in lane 63 only: wave_sum_f32_lane63, mpb_common.h -- the same association without the trip through scalar registers)
            const float cost_s = wave_sum_f32_lane63((pos_l + ctl_l) + (term_l + temp * is_l));
            const float coll_s = (geom != nullptr) ? wave_sum_f32_lane63(coll_l) : 0.f;
            if (lane == 63) {
                M.cst[s] = cost_s;
                M.coll[s] = coll_s;
            }
        }
        // ---- quirk Q6: the per-sample collision costs collapse into ONE scalar added to every sample
        float total = 0.f;
        __syncthreads();
        // (S <= 64: only the waves of the two phases below that read it -- wave 0, save-best, and the softmax's wave -- form it)
        // the softmax runs BESIDE save-best on a wave of its own (S <= 64) -- when the workgroup has a second wave (S = 1 or
        // MPB_MPPI_WAVES=1 launch one: wave 0 then does both, one after the other)
        const int sm_wave = (has_best && nw > 1) ? 1 : 0;
        if (geom != nullptr && (S > 64 || wave <= sm_wave)) {
            // summed by every wave for itself in an order that depends on S alone (lane-strided partial sums, then the
            // wave reduction): the scalar -- and with it every cost and weight -- is the same however many waves the
            // problem was given (mpb_mppi_step picks 8 or 16 by the number of problems)
            float a = 0.f;
            for (int ss = lane; ss < S; ss += 64) a += M.coll[ss];
            total = weight * (k_sigma * wave_sum_f32(a));
        }
        // ---- MPPI._save_best (mppi.py:164-168), called every iteration before update_controller (mppi.py:148): the
        //      cheapest sample so far (first index on ties, like torch.argmin) and its state trajectory.  The states of
        //      every sample are not kept, so the winner's rollout is redone from its controls in LDS (same arithmetic
        //      as above, bit-identical); only wave 0 works, the others go on to the softmax
        if (has_best && wave == 0) {
            float bv = 3.0e38f;
            int bi = 0;
            for (int ss = lane; ss < S; ss += 64) {
                const float v = M.cst[ss] + total;
                if (v < bv) { bv = v; bi = ss; }
            }
            // the wave's minimum, then the smallest index among the lanes that hold it (torch.argmin: first index on ties; indices are
            // exact in fp32): two DPP reductions.  (A butterfly of (value, index) pairs was twelve ds_bpermute round trips on the path
            // of the one wave every other wave then waits for at the barrier below.)
            const float mn = -wave_max_f32(-bv);
            bi = (int)(-wave_max_f32(bv == mn ? -(float)bi : -3.0e38f));
            bv = mn;
            if (bv < best_c) {                                   // wave-uniform
                best_c = bv;
                if (lane == 0) MPPI_KARG(best_cost)[prob] = bv;
                float carry[MPPI_MAX_C];
#pragma unroll
                for (int i = 0; i < MPPI_MAX_C; ++i) carry[i] = 0.f;
                for (int base = 0; base < T; base += 64) {
                    const int t = base + lane;
                    const bool on = t < T;
#pragma unroll
                    for (int i = 0; i < MPPI_MAX_C; ++i) {
                        const float u = (i < c && on) ? M.Us[(size_t)bi * us_stride + i * T + t] : 0.f;
                        const float v = (i < c && on) ? fminf(fmaxf(u, umin[i]), umax[i]) * dt : 0.f;
                        const float inc = wave_scan_incl(v, lane);
                        const float x = x0[i] + (carry[i] + (inc - v));
                        carry[i] += readlane_f32(inc, 63);
                        if (i < c && on) MPPI_KARG(best_states)[((size_t)prob * T + t) * c + i] = x;
                    }
                }
            }
        }
        // ---- softmax over samples (mppi.py:73-76)
        if (S <= 64) {
            // ONE wave (lane = sample, wave reductions): no block-wide reduction, one barrier.  (Until round 5 every wave repeated it and
            // wave 0 stored it -- behind save-best, with the other waves' copies competing for its SIMD.)
            if (wave == sm_wave) {
            const float cs = (lane < S) ? M.cst[lane] + total : 0.f;
            // (v_exp / v_rcp: mpb_common.h fast_expf / fast_rcpf, as in the persistent STOMP kernels; the sum holds exp(0) = 1)
            const float xs = (lane < S) ? -cs * inv_temp : -3.0e38f;
            const float mx = wave_max_f32(xs);
            const float ex = (lane < S) ? fast_expf(xs - mx) : 0.f;
            const float w = ex * fast_rcpf(wave_sum_f32(ex));
            if (lane < S) {
                M.wts[lane] = w;
                if (last) {
                    MPPI_KARG(costs)[(size_t)prob * S + lane] = cs;
                    MPPI_KARG(weights)[(size_t)prob * S + lane] = w;
                }
            }
            }
        } else {
            float mx = -3.0e38f;
            for (int ss = threadIdx.x; ss < S; ss += blockDim.x) mx = fmaxf(mx, -(M.cst[ss] + total) / temp);
            mx = block_max(mx, M.red, lane, wave, nw);
            float z = 0.f;
            for (int ss = threadIdx.x; ss < S; ss += blockDim.x) z += expf(-(M.cst[ss] + total) / temp - mx);
            z = block_sum(z, M.red, lane, wave, nw);
            for (int ss = threadIdx.x; ss < S; ss += blockDim.x) {
                const float cs = M.cst[ss] + total;
                const float w = expf(-cs / temp - mx) / z;
                M.wts[ss] = w;
                if (last) {
                    MPPI_KARG(costs)[(size_t)prob * S + ss] = cs;
                    MPPI_KARG(weights)[(size_t)prob * S + ss] = w;
                }
            }
        }
        __syncthreads();
        // ---- mean += step * sum_s w_s (U_s - mean)   (mppi.py:79-84).  Sixteen elements (i, t) per wave, FOUR lanes per element
        //      (one quad): lane p of the quad sums the samples s = p, p + 4, p + 8, ... in ascending order, the four partial
        //      sums meet as (p0 + p1) + (p2 + p3) -- an association that depends on S alone, so the result does not depend on
        //      how many waves the problem was given.  (One thread per element walked all S samples: 128 busy threads of 512,
        //      two dependent LDS reads per sample each.)
        for (int i = 0; i < c; ++i)
        for (int t0 = 0; t0 < T; t0 += 16 * nw) {
            // (the four partial sums of an element sit in one QUAD: they meet on DPP quad permutes -- [1,0,3,2], then [2,3,0,1] -- where
            // a lane-row per partial sum needed two ds_bpermute round trips; the same association)
            const int tt = t0 + 16 * wave + (lane >> 2), part = lane & 3;
            const bool on = tt < T;
            const int t = on ? tt : 0;
            const float mu = m[t * c + i];
            float a = 0.f;
            for (int ss = part; ss < S; ss += 4) a += M.wts[ss] * (M.Us[(size_t)ss * us_stride + i * T + t] - mu);
            a += dpp_f32<0xB1>(a);
            a += dpp_f32<0x4E>(a);
            if (on && part == 0) m[t * c + i] = mu + step_size * a;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * c; e += blockDim.x) mean_g[e] = m[e];
}

extern "C" int mpb_mppi_step(float* mean, const float* eps, const float* scale_tril, const float* cov_inv,
                             const float* state0, const float* goal, const float* ctrl_min, const float* ctrl_max,
                             const float* discount, const float* c_weights, const float* geom, int geom_flags, float* controls,
                             float* states, float* costs, float* weights, float* best_cost, float* best_states,
                             int NP, int S, int T, int c, int control_type,
                             float dt, float k_sigma, float weight, float temp, float step_size, int n_iters,
                             uint64_t seed, uint32_t iter0, void* stream) {
    if (!mean || !scale_tril || !cov_inv || !state0 || !goal || !ctrl_min || !ctrl_max || !discount || !c_weights ||
        !controls || !states || !costs || !weights)
        return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: null pointer");
    if (NP < 0 || S < 1 || S > 1024 || T < 2 || T > MPB_MAX_H || c < 1 || c > MPPI_MAX_C || n_iters < 0)
        return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: bad shape");
    if (!(temp > 0.f)) return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: temp must be > 0");
    if ((best_cost == nullptr) != (best_states == nullptr))
        return mpb_fail(MPB_E_INVALID, "mpb_mppi_step: best_cost and best_states must be given together");
    if (control_type != 0)
        return mpb_fail(MPB_E_UNSUPPORTED, "mpb_mppi_step: only velocity control (the reference's acceleration mode cannot run)");
    if (NP == 0 || n_iters == 0) return MPB_OK;
    static const int force_nw = getenv("MPB_MPPI_WAVES") ? atoi(getenv("MPB_MPPI_WAVES")) : 0;       // tuning aid
    // one wave per sample, at most 16 waves per problem.  With at least two problems per CU, workgroups of 8 waves (two
    // resident per CU: one problem's barriers and serial sections run under the other's rollouts) are 16 % faster than
    // 16 (NP = 1 024, S = 32, T = 64: 68.7 against 81.5 us per iteration); a single problem is fastest on 16
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    int nw = S < 16 ? S : 16;
    if (NP >= 2 * n_cu && S >= 16) nw = 8;
    if (force_nw > 0 && force_nw <= 16 && force_nw <= S) nw = force_nw;
    const size_t base = (size_t)((c * T + 3) & ~3) + (size_t)c * T + 64 + 3 * (size_t)((S + 3) & ~3) + (size_t)S * (c * T + 4) + (size_t)nw * c * T;
    const size_t with_tril = base + (size_t)c * T * T;
    const size_t budget = 150 * 1024 / sizeof(float);
    if (base > budget) return mpb_fail(MPB_E_UNSUPPORTED, "mpb_mppi_step: S*T*c too large for the LDS controls slab");
    // the matrix path swaps the per-wave normals + transposed factor for the MFMA images + every sample's normals
    const size_t with_matrix = base - (size_t)nw * c * T + 4 + (size_t)c * 4096 + (size_t)c * ((S + 15) & ~15) * MPPI_E_STRIDE;
    static const int force_mode = getenv("MPB_MPPI_NOISE") ? atoi(getenv("MPB_MPPI_NOISE")) : -1;     // tuning / tests: 0, 1, 2
    int noise_mode = (T <= 64 && with_matrix <= budget) ? MPPI_NOISE_MATRIX : (with_tril <= budget ? MPPI_NOISE_LDS : MPPI_NOISE_GLOBAL);
    if (force_mode == MPPI_NOISE_GLOBAL) noise_mode = MPPI_NOISE_GLOBAL;
    if (force_mode == MPPI_NOISE_LDS && with_tril <= budget) noise_mode = MPPI_NOISE_LDS;
    size_t lds_words = noise_mode == MPPI_NOISE_MATRIX ? with_matrix : noise_mode == MPPI_NOISE_LDS ? with_tril : base;
    // ONE grid-backed collision field (geom_flags bit 8, bits 16-28 = its cells): the grid + obstacle table ride in LDS when
    // they fit next to the rest (and, with two workgroups per CU, leave room for the second one)
    int grid_words = 0;
    static const int no_grid = getenv("MPB_MPPI_NO_GRID") ? atoi(getenv("MPB_MPPI_NO_GRID")) : 0;           // tuning / tests
    if (geom && (geom_flags & 0x1500) == 0x1500 && !no_grid) {     // ONE grid-backed field, point robot
        const int cells = (geom_flags >> 16) & 0x1FFF;
        const size_t extra = 4 + (size_t)((cells + 3) & ~3) + 4 * (MPB_GRID_MAX_SPH + 1);
        const size_t cap = (nw <= 8 ? 78 : 150) * 1024 / sizeof(float);
        if (cells > 0 && lds_words + extra <= cap) {
            grid_words = cells;
            lds_words += extra;
        }
    }
    const size_t lds = lds_words * sizeof(float);
#define MPPI_LAUNCH(CC, GRID)                                          \
    do {                                                               \
        if (noise_mode == MPPI_NOISE_MATRIX && eps) MPPI_LAUNCH_(CC, GRID, true, true);    \
        else if (noise_mode == MPPI_NOISE_MATRIX) MPPI_LAUNCH_(CC, GRID, true, false);    \
        else MPPI_LAUNCH_(CC, GRID, false, true);                      \
    } while (0)
#define MPPI_LAUNCH_(CC, GRID, MATRIX, INJ)                                                                                                   \
    hipLaunchKernelGGL((mppi_kernel<CC, GRID, MATRIX, INJ>), dim3(NP), dim3(64 * nw), lds, (hipStream_t)stream, mean, eps, scale_tril, cov_inv, \
                       state0, goal, ctrl_min, ctrl_max, discount, c_weights, geom, controls, states, costs, weights, best_cost,  \
                       best_states, S, T, c, dt, k_sigma, weight, temp, step_size, n_iters, (uint32_t)seed,                        \
                       (uint32_t)(seed >> 32), iter0, noise_mode, grid_words)
    if (c == 2 && grid_words > 0) MPPI_LAUNCH(2, true);
    else if (c == 2) MPPI_LAUNCH(2, false);
    else if (grid_words > 0) MPPI_LAUNCH(0, true);
    else MPPI_LAUNCH(0, false);
#undef MPPI_LAUNCH
#undef MPPI_LAUNCH_
    return mpb_check_launch("mpb_mppi_step");
}

// ------------------------------------------------------------------------------------------------
// The point-particle system as stand-alone entry points (dynamics/point.py): code written against the reference's
// `system.dynamics(x, u)` / `system.traj_cost(X, U)` runs on these; the MPPI kernel above fuses the same arithmetic.
// ------------------------------------------------------------------------------------------------
// x_next = x + (clamp(u, ctrl_min, ctrl_max) + dyn_std * noise) * dt   (point.py:102-140; the reference's xdot = cat(x[..., state_dim:], u)
// has an EMPTY first part: the state holds state_dim entries), n rows of dim entries; noise NULL: deterministic
__global__ void point_dynamics_kernel(const float* __restrict__ x, const float* __restrict__ u, const float* __restrict__ ctrl_min,
                                      const float* __restrict__ ctrl_max, const float* __restrict__ dyn_std,
                                      const float* __restrict__ noise, float* __restrict__ x_next, size_t n_elem, int dim, float dt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elem) return;
    const int d = (int)(i % (size_t)dim);
    float uc = fminf(fmaxf(u[i], ctrl_min[d]), ctrl_max[d]);
    if (noise) uc += dyn_std[d] * noise[i];
    x_next[i] = fmaf(uc, dt, x[i]);
}

extern "C" int mpb_point_dynamics(const float* x, const float* u, const float* ctrl_min, const float* ctrl_max, const float* dyn_std,
                                  const float* noise, float* x_next, size_t n, int dim, float dt, void* stream) {
    if (dim < 1 || dim > 64) return mpb_fail(MPB_E_INVALID, "mpb_point_dynamics: bad dimension");
    if (n == 0) return MPB_OK;                   // (an empty batch has no buffers to speak of)
    if (!x || !u || !ctrl_min || !ctrl_max || !x_next || (noise && !dyn_std)) return mpb_fail(MPB_E_INVALID, "mpb_point_dynamics: null pointer");
    const size_t ne = n * (size_t)dim;
    hipLaunchKernelGGL(point_dynamics_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, u, ctrl_min,
                       ctrl_max, dyn_std, noise, x_next, ne, dim, dt);
    return mpb_check_launch("mpb_point_dynamics");
}

// quadratic trajectory cost of B rollouts (point.py:154-226): X (T, B, sd), U (T, B, cd) in the reference's time-major
// layout, one thread per rollout (adjacent threads read adjacent rows: coalesced);
//   cost_b = sum_t disc_t (w_pos |X_tb - goal|^2 + w_ctrl |U_tb|^2) + w_posT disc_{T-1} |X_{T-1,b} - goal|^2 + energy
// the velocity term of the reference slices dX[..., state_dim:control_dim], which is empty (quirk Q8): it contributes 0.
__global__ void point_traj_cost_kernel(const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ goal,
                                       const float* __restrict__ discount, float w_pos, float w_ctrl, float w_posT, float energy,
                                       float* __restrict__ out, int T, int B, int sd, int cd) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float pos = 0.f, ctrl = 0.f, last = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* xr = X + ((size_t)t * B + b) * sd;
        const float* ur = U + ((size_t)t * B + b) * cd;
        float sx = 0.f, su = 0.f;
        for (int d = 0; d < sd; ++d) { const float e = xr[d] - goal[d]; sx = fmaf(e, e, sx); }
        for (int d = 0; d < cd; ++d) su = fmaf(ur[d], ur[d], su);
        const float dsc = discount[t];
        pos = fmaf(sx * w_pos, dsc, pos);
        ctrl = fmaf(su * w_ctrl, dsc, ctrl);
        last = sx;
    }
    out[b] = pos + 0.f + ctrl + (last * w_posT) * discount[T - 1] + energy;   // (pos + vel + control + terminal + energy, point.py:225)
}

extern "C" int mpb_point_traj_cost(const float* X, const float* U, const float* goal, const float* discount, float w_pos, float w_vel,
                                   float w_ctrl, float w_pos_T, float energy, float* costs, int T, int B, int state_dim, int ctrl_dim,
                                   void* stream) {
    (void)w_vel;     // (quirk Q8: the velocity slice is empty)
    if (T < 1 || B < 0 || state_dim < 1 || state_dim > 64 || ctrl_dim < 1 || ctrl_dim > 64)
        return mpb_fail(MPB_E_INVALID, "mpb_point_traj_cost: bad shape");
    if (B == 0) return MPB_OK;
    if (!X || !U || !goal || !discount || !costs) return mpb_fail(MPB_E_INVALID, "mpb_point_traj_cost: null pointer");
    hipLaunchKernelGGL(point_traj_cost_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, X, U, goal, discount,
                       w_pos, w_ctrl, w_pos_T, energy, costs, T, B, state_dim, ctrl_dim);
    return mpb_check_launch("mpb_point_traj_cost");
}

