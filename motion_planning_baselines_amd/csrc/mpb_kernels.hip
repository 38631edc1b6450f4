// mpb_kernels.hip -- gfx950 kernels + C-ABI (include/mpb.h) for the STOMP / CHOMP / collision-cost paths.
//
// Work mapping (CDNA4: 64-lane waves, 4 SIMDs per CU, 256 CUs):
//   * one WAVE per rollout (trajectory), one LANE per waypoint -- H = 64 fills a wave exactly; longer
//     horizons loop in 64-waypoint chunks.  FK + SDF of a waypoint run entirely in registers with all
//     geometry constants as scalar (SGPR) operands; the per-trajectory cost is a wave reduction.
//   * B = P*S rollouts -> B waves; C3 (B = 4096) is exactly 4 waves per SIMD over the whole chip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mpb_common.h"
#include <hip/hip_ext.h>
#include "mpb_geom.h"
#include "mpb_stomp_noise.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
char* mpb_err_buf() { return g_err; }

static int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#define MPB_REQUIRE(cond, msg)                                                     \
    do {                                                                           \
        if (!(cond)) return fail(MPB_E_INVALID, "%s: requirement failed: " msg " [%ld,%ld]", __func__, 0, 0); \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return MPB_E_HIP;
    }
    return MPB_OK;
}

#ifdef MPB_TUNING_BUILD
extern "C" int mpb_version(void) { return MPB_ABI_VERSION | MPB_VERSION_TUNING_BUILD; }
#else
extern "C" int mpb_version(void) { return MPB_ABI_VERSION; }
#endif
extern "C" const char* mpb_last_error(void) { return g_err; }

// A buffer tagged with a compile-time robot model (header word 29) must carry exactly that model's tables: the joint
// transforms bit for bit, and as link table the model's collision spheres selected by the keep mask (word 30), of
// which only spheres on frame 1 may be missing.  The model kernels never read these tables -- they trust the tag.
template <class M>
static int model_check_as(const float* g, const char* who) {
    const int32_t* gi = reinterpret_cast<const int32_t*>(g);
    const uint32_t keep = (uint32_t)gi[30];
    if (gi[2] != MPB_KIND_CHAIN || gi[3] != M::N_DOF || gi[4] != M::N_TF) return fail(MPB_E_INVALID, "%s: model tag does not match the robot", who);
    if (memcmp(g + gi[9], M::TF, sizeof(float) * 12 * M::N_TF) != 0) return fail(MPB_E_INVALID, "%s: joint transforms differ from the tagged model", who);
    int n = 0;
    for (int l = 0; l < M::N_LINKS; ++l) {
        if (!((keep >> l) & 1u)) {
            if (M::LINK_FRAME[l] != 1) return fail(MPB_E_INVALID, "%s: only frame-1 spheres of a model may be pruned", who);
            continue;
        }
        if (n >= gi[5]) return fail(MPB_E_INVALID, "%s: keep mask and link table disagree", who);
        const float* lk = g + gi[10] + 8 * n;
        if (reinterpret_cast<const int32_t*>(lk)[0] != M::LINK_FRAME[l] || memcmp(lk + 1, M::LINK[l], 4 * sizeof(float)) != 0)
            return fail(MPB_E_INVALID, "%s: link table differs from the tagged model", who);
        ++n;
    }
    if (n != gi[5] || (M::N_LINKS < 32 && (keep >> M::N_LINKS) != 0u)) return fail(MPB_E_INVALID, "%s: keep mask and link table disagree", who);
    // the model kernels clamp a hinge to [0, 1] (mpb_geom.h, UNIT): margin + largest collision sphere + deepest possible
    // penetration (sphere radius / smallest half extent of a box) must stay below 1
    float rl_max = 0.f, deepest = 0.f;
    for (int l = 0; l < M::N_LINKS; ++l) rl_max = M::LINK[l][3] > rl_max ? M::LINK[l][3] : rl_max;
    for (int o = 0; o < gi[6]; ++o) deepest = g[gi[11] + 4 * o + 3] > deepest ? g[gi[11] + 4 * o + 3] : deepest;
    for (int o = 0; o < gi[7]; ++o) {
        const float* h = g + gi[12] + 8 * o + 4;
        const float m = h[0] < h[1] ? (h[0] < h[2] ? h[0] : h[2]) : (h[1] < h[2] ? h[1] : h[2]);
        deepest = m > deepest ? m : deepest;
    }
    if (!(g[8] + rl_max + deepest < 1.0f)) return fail(MPB_E_INVALID, "%s: a model-tagged scene must keep every hinge below 1 (margin + radii)", who);
    return MPB_OK;
}

static int model_check(const float* g, const char* who) {
    const int model = reinterpret_cast<const int32_t*>(g)[29];
    if (model == 0) return MPB_OK;
    if (model == PandaModel::ID) return model_check_as<PandaModel>(g, who);
    return fail(MPB_E_INVALID, "%s: unknown robot model id", who);
}

static int geom_check_one(const float* g, int n_words, const char* who) {
    if (!g || n_words < MPB_GEOM_HEADER_WORDS) return fail(MPB_E_INVALID, "%s: geometry buffer too small", who);
    const int32_t* gi = reinterpret_cast<const int32_t*>(g);
    if (gi[0] != MPB_GEOM_MAGIC || (gi[1] != MPB_GEOM_VERSION && gi[1] != MPB_GEOM_VERSION_LIST)) return fail(MPB_E_INVALID, "%s: bad magic/version", who);
    const bool list = gi[1] == MPB_GEOM_VERSION_LIST;       // version 7: the grid section is a list grid (mpb_geom.h, spheres_hinge_list)
    const int kind = gi[2], n_dof = gi[3], n_tf = gi[4], n_links = gi[5], n_sph = gi[6], n_box = gi[7];
    if (kind != MPB_KIND_POINT && kind != MPB_KIND_CHAIN) return fail(MPB_E_INVALID, "%s: unknown robot kind", who);
    if (n_dof < 1 || n_dof > MPB_MAX_DOF) return fail(MPB_E_INVALID, "%s: n_dof out of range", who);
    if (kind == MPB_KIND_POINT && (n_dof < 2 || n_dof > 3 || n_links != 1)) return fail(MPB_E_INVALID, "%s: point robot must be 2-D/3-D with one sphere", who);
    if (kind == MPB_KIND_CHAIN && n_tf != n_dof + 1) return fail(MPB_E_INVALID, "%s: chain needs n_dof+1 transforms", who);
    if (n_links < 1 || n_sph < 0 || n_box < 0 || n_sph + n_box < 1) return fail(MPB_E_INVALID, "%s: empty link/obstacle set", who);
    const int off_tf = gi[9], off_links = gi[10], off_sph = gi[11], off_box = gi[12], total = gi[13];
    const int off_cull = gi[14], off_fs = gi[15], off_grid = gi[16];
    const int gnx = gi[17], gny = gi[18], gnz = gi[19], n_cells = gi[26];
    const int n_sph_pad = (n_sph + 3) / 4 * 4;
    const int n_frames = n_tf > 1 ? n_tf : 1;
    const int n_fs = (n_frames + 1 + 3) / 4 * 4;
    if (off_tf != MPB_GEOM_HEADER_WORDS || off_links != off_tf + 12 * n_tf || off_sph != off_links + 8 * n_links ||
        off_box != off_sph + 4 * n_sph || off_cull != off_box + 8 * n_box || off_fs != off_cull + 8 * n_sph_pad ||
        off_grid != off_fs + n_fs || total > n_words)
        return fail(MPB_E_INVALID, "%s: inconsistent section offsets", who);
    const int off_cand = off_grid + (n_cells + MPB_GRID_PAD - 1) / MPB_GRID_PAD * MPB_GRID_PAD;
    if (list ? (total < off_cand + 4 || ((total - off_cand) & 3)) : total != off_cand) return fail(MPB_E_INVALID, "%s: inconsistent section offsets", who);
    if (list && (n_cells < 1 || n_cells > MPB_GRID_MAX_CELLS || n_sph > MPB_LIST_MAX_SPH || n_box > MPB_LIST_MAX_BOX ||
                 4 * (total - off_cand) > MPB_LIST_MAX_CAND + 16))
        return fail(MPB_E_INVALID, "%s: a list grid needs 1..4096 cells, <= 255 spheres, <= 127 boxes, <= 16 KB of candidates", who);
    if ((off_links | off_sph | off_box | off_cull | off_fs | off_grid) & 3) return fail(MPB_E_INVALID, "%s: sections must be 16-byte aligned", who);
    if (n_cells < 0 || (n_cells > 0 && (gnx < 1 || gny < 1 || gnz < 1 || gnx * gny * gnz != n_cells)))
        return fail(MPB_E_INVALID, "%s: bad broad-phase grid dims", who);
    if (n_cells > 0) {
        // version 6: cells on a lattice through the origin -- lo = (K - 1/2) h per axis with integer K, and header word 31 =
        // Kx + gnx (Ky + gny Kz): what grid_cell_rel (mpb_geom.h) turns round(x / h) into a cell with
        long K[3];
        for (int a = 0; a < 3; ++a) {
            if (!(g[23 + a] > 0.f)) return fail(MPB_E_INVALID, "%s: bad grid cell size", who);
            const double k = (double)g[20 + a] * (double)g[23 + a] + 0.5;
            K[a] = lrint(k);
            // (lo and 1/h are fp32: their product carries ~|K| 1.2e-7 of rounding, so the tolerance scales with |K| -- an absolute
            // 1e-3 refused the grids build_grid makes beyond |K| ~ 16 700)
            if (fabs(k - (double)K[a]) > 1e-3 + 4e-7 * fabs((double)K[a]) || labs(K[a]) > 100000) return fail(MPB_E_INVALID, "%s: grid origin is not on the cell lattice", who);
        }
        if ((long)gi[31] != K[0] + (long)gnx * (K[1] + (long)gny * K[2])) return fail(MPB_E_INVALID, "%s: grid lattice index (word 31) does not match the origin", who);
        if (labs((long)gi[31]) + (long)n_cells >= (1L << 21)) return fail(MPB_E_INVALID, "%s: grid too far from the origin for the fp32 cell index", who);
    }
    if (list) {   // every cell's candidate range must lie inside the candidate bytes and name existing obstacles
        const unsigned char* cand = reinterpret_cast<const unsigned char*>(g + off_cand);
        const int n_cand = 4 * (total - off_cand);
        for (int i = 0; i < n_cells; ++i) {
            const uint32_t w = (uint32_t)gi[off_grid + i];
            if (w & 0x80000000u) continue;                    // overflowing cell: the kernels test every obstacle
            const int start = (int)(w & 0x7FFFu), ns = (int)((w >> 15) & 0x7Fu), nb = (int)((w >> 22) & 0x3Fu);
            if ((w & 0x70000000u) || ns > MPB_LIST_CELL_MAX_SPH || nb > MPB_LIST_CELL_MAX_BOX || start + ns + nb > n_cand)
                return fail(MPB_E_INVALID, "%s: list-grid cell out of range", who);
            for (int k = 0; k < ns; ++k)
                if ((int)cand[start + k] >= n_sph) return fail(MPB_E_INVALID, "%s: grid cell references a missing obstacle", who);
            for (int k = 0; k < nb; ++k)
                if ((int)cand[start + ns + k] >= n_box) return fail(MPB_E_INVALID, "%s: grid cell references a missing obstacle", who);
        }
    } else
    for (int i = 0; i < n_cells; ++i) {   // every packed obstacle index must exist
        const uint32_t w = (uint32_t)gi[off_grid + i];
        if (w == 0xFFFFFFFEu) continue;
        for (int k = 0; k < 4; ++k) {
            const uint32_t idx = (w >> (8 * k)) & 0xFFu;      // n_sph = unused slot (the far dummy the kernels append)
            if ((int)idx > n_sph) return fail(MPB_E_INVALID, "%s: grid cell references a missing obstacle", who);
        }
    }
    // frame -> link ranges must be monotone and end at n_links
    for (int j = 0; j < n_frames; ++j)
        if (gi[off_fs + j] < 0 || gi[off_fs + j] > gi[off_fs + j + 1] || gi[off_fs + j + 1] > n_links)
            return fail(MPB_E_INVALID, "%s: bad frame_start table", who);
    if (gi[off_fs] != 0 || gi[off_fs + n_frames] != n_links) return fail(MPB_E_INVALID, "%s: frame_start must cover all links", who);
    int prev = 1;
    for (int l = 0; l < n_links; ++l) {
        const int f = gi[off_links + 8 * l];
        if (kind == MPB_KIND_CHAIN && (f < prev || f > n_dof + 1)) return fail(MPB_E_INVALID, "%s: link frames must be sorted in [1, n_dof+1]", who);
        prev = f > prev ? f : prev;
    }
    if (!(g[28] >= 0.f)) return fail(MPB_E_INVALID, "%s: field scale must be >= 0", who);
    return model_check(g, who);
}


extern "C" int mpb_geom_check(const float* g, int n_words) {
    // a buffer may chain up to MPB_MAX_FIELDS fields: header word 27 = words from this header to the next one
    int off = 0;
    for (int f = 0; f < MPB_MAX_FIELDS; ++f) {
        if (!g || n_words - off < MPB_GEOM_HEADER_WORDS) return fail(MPB_E_INVALID, "%s: geometry buffer too small", __func__);
        const int rc = geom_check_one(g + off, n_words - off, "mpb_geom_check");
        if (rc) return rc;
        const int32_t* gi = reinterpret_cast<const int32_t*>(g + off);
        if (f > 0 && (gi[2] != reinterpret_cast<const int32_t*>(g)[2] || gi[3] != reinterpret_cast<const int32_t*>(g)[3]))
            return fail(MPB_E_INVALID, "%s: chained fields must share the robot", __func__);
        const int next = gi[27];
        if (next == 0) return MPB_OK;
        if (next < gi[13] || (next & 3)) return fail(MPB_E_INVALID, "%s: bad offset to the next field", __func__);
        off += next;
    }
    return fail(MPB_E_INVALID, "%s: more than MPB_MAX_FIELDS chained fields", __func__);
}

extern "C" int mpb_geom_flags(const float* g, int n_words, int* flags) {
    if (!flags) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    *flags = 0;
    const int rc = mpb_geom_check(g, n_words);
    if (rc) return rc;
    int model = -1, max_cells = 0, n_fields = 0;
    bool all_grids = true, all_lists = true;
    for (int off = 0;;) {
        const int32_t* gi = reinterpret_cast<const int32_t*>(g + off);
        const bool is_list = gi[1] == MPB_GEOM_VERSION_LIST;                                                // (checked above: usable as it stands)
        const bool grid_ok = !is_list && gi[26] > 0 && gi[26] <= MPB_GRID_MAX_CELLS && gi[6] <= MPB_GRID_MAX_SPH;   // grid_usable()
        all_grids = all_grids && grid_ok;
        all_lists = all_lists && is_list;
        if (gi[26] > max_cells) max_cells = gi[26];
        ++n_fields;
        const int m = (grid_ok || is_list) ? gi[29] : 0;
        model = (model < 0 || model == m) ? m : 0;
        if (gi[27] == 0) break;
        off += gi[27];
    }
    const int32_t* g0 = reinterpret_cast<const int32_t*>(g);
    const bool point_small = g0[2] == MPB_KIND_POINT && g0[27] == 0 && g0[6] <= 32 && g0[7] <= 8;
    *flags = (model > 0 ? (model & 0xFF) : 0) | (all_grids ? 0x100 : 0) | (point_small ? 0x200 : 0) |
             (g0[2] == MPB_KIND_POINT ? 0x400 : 0) | (n_fields == 1 ? 0x1000 : 0) | (all_lists ? 0x2000 : 0) |
             ((all_grids || all_lists) ? (max_cells & 0x1FFF) << 16 : 0);
    return MPB_OK;
}

#define MPB_MAX_D 16       // channels of a STOMP rollout (one matrix-core tile of channels): D <= 8 with velocities, D <= 12 position only

// Measurement aid (mpb_stomp_step_profile): while these are set, the STOMP kernel launches record the pair on the
// dispatch itself (hipExtLaunchKernelGGL: kernel begin / end timestamps, the quantity rocprofv3 --kernel-trace reports).
static thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
#define MPB_LAUNCH(kernel, grid, block, lds, st, ...)                                                   \
    do {                                                                                                \
        if (t_ev0) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t_ev0, t_ev1, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                             \
    } while (0)

// ------------------------------------------------------------------------------------------------
// STOMP kernel A, H = 64 fast path: the time-correlated noise  N = L * eps  (64x64 lower-triangular L,
// the scale_tril of the precision matrix R; eps 64 x d) runs on the matrix cores as exact-fp32
// v_mfma_f32_16x16x4_f32 tiles, one rollout per wave:
//   A operand  L[16m+i][4ks+g]   (lane i = l&15, g = l>>4)  from an LDS image laid out so that one
//                                 ds_read_b128 per lane delivers four k-steps, conflict-free;
//   B operand  eps[c=j][4ks+g]   (lane j = l&15, g = l>>4)  generated in registers (Philox) or loaded;
//   lower-triangular: row tile m only needs k-steps ks <= 4m+3  ->  40 instead of 64 MFMAs.
// The D tiles (lane = channel) go through a padded LDS tile to the lane = waypoint layout the
// FK + SDF cost evaluation wants.  DCH = d is a compile-time channel count (no per-channel branches).
// ------------------------------------------------------------------------------------------------

#ifdef MPB_STAMPS  // diagnostic build only: per-wave s_memtime stamps of the phases of kernel A
__device__ unsigned long long g_stamps[4096 * 10];   // slots 0-7: s_memtime (shader clock) per phase; 8, 9: s_memrealtime (100 MHz) at entry / exit
#define MPB_STAMP(k)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 4096) {             \
            g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + (k)] = t_;                      \
            if ((k) == 0 || (k) == 7)                                                             \
                g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 10 + ((k) == 0 ? 8 : 9)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                         \
    } while (0)
extern "C" int mpb_debug_read_stamps(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 3;
}
#else
#define MPB_STAMP(k)
#endif

#ifndef MPB_A_WPB
#define MPB_A_WPB 4      // waves (rollouts) per block of the H = 64 kernel.  Measured at C3 / P = 4096: 2 waves 42 / 878 us (the
                         // L and grid staging is per block), 4 waves 30.0 / 638, 8 waves 29.6 / 670, 16 waves 30.0 / 779
#endif
// LDS of one block: the permuted L image (16 KB), which is dead once the noise product is done and then holds the
// per-wave transpose / pack tiles (5 KB per wave), and -- in a region of its own, so that it is staged at kernel entry
// together with L instead of in a serial phase between two more barriers -- the broad-phase grid + obstacle table of
// the (first) collision field.  37 KB per block: four blocks (16 waves) per CU, the whole C3 batch in one round.
#define MPB_A_TILE_FLOATS (64 * NT_STRIDE * (MPB_A_WPB > 4 ? MPB_A_WPB : 4))
static_assert(MPB_A_TILE_FLOATS >= STOMP_LIMG_WORDS, "the L image shares the tile region");
#define MPB_A_GRID_ROUNDS (MPB_GRID_MAX_CELLS / 4 / (64 * MPB_A_WPB))   // uint4 per thread for the largest grid
// MODEL: 0 = generic table-driven chain walk (any robot, grid or exhaustive obstacle loop per field); > 0 = the
// compile-time robot model of that id (mpb_model_*.h) -- separate instantiations, because the register allocation of
// one kernel holding all three evaluators spills.
template <int DCH, bool WITH_COST, int MODEL>
__global__ __launch_bounds__(64 * MPB_A_WPB, 16 / MPB_A_WPB) void stomp_sample_cost_h64_kernel(
    const float* __restrict__ means, const float* __restrict__ eps, float* __restrict__ samples,
    float* __restrict__ costs, const float* __restrict__ Lmat, const float* __restrict__ geom,
    int P, int S, float k_sigma, float weight, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter,
    uint32_t particle_offset) {
    constexpr int H = 64;
    constexpr int NTHR = 64 * MPB_A_WPB;
    MPB_STAMP(0);
    __shared__ __attribute__((aligned(16))) float Lp[MPB_A_TILE_FLOATS];  // permuted L (first 16 KB), then the wave tiles
    __shared__ __attribute__((aligned(16))) unsigned gridw[WITH_COST ? MPB_GRID_MAX_CELLS : 4];   // broad-phase grid of the first field
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];                         // its obstacle table
    // L and the grid are read here (coalesced, kept in registers) and written to LDS only after the noise has been
    // drawn: the Philox + Box-Muller phase below needs neither, so the load latency (every block of the launch hits
    // L2 at the same moment) hides behind it
    constexpr int L_PER_THREAD = H * H / 4 / NTHR;
    f32x4 lreg[L_PER_THREAD];
#pragma unroll
    for (int u = 0; u < L_PER_THREAD; ++u) lreg[u] = reinterpret_cast<const f32x4*>(Lmat)[threadIdx.x + NTHR * u];
    static_assert(MPB_GRID_PAD % (4 * 64 * MPB_A_WPB) == 0, "the grid section is padded to whole rounds of the block's 16-byte loads");
    float4 oreg = make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
    bool grid0 = false;
    int g_nsph = 0;
    // the (first) field's header is read HERE, before this kernel's first global store: scalar loads into SGPRs that the
    // cost section below reuses.  Re-reading it there, behind the sample stores, costs a chain of vector loads +
    // v_readfirstlane (the compiler cannot use the scalar cache for memory the kernel may have written): ~3 k cycles
    GeomView G0;
    if (WITH_COST) {
        G0 = geom_view(geom);
        grid0 = grid_usable(G0);
        if (grid0) {
            // the grid goes global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write; LDS address = wave-uniform
            // base + 16 * lane, i.e. a straight copy); the section is padded to whole rounds (MPB_GRID_PAD words): no tail.
            // The loads are in flight during the Philox phase; the barrier behind the L image drains them (vmcnt(0)).
            const int g_rounds = (G0.n_cells + 4 * NTHR - 1) / (4 * NTHR);
            g_nsph = G0.n_sph;
            const uint4* g4 = reinterpret_cast<const uint4*>(G0.grid);
#pragma unroll
            for (int u = 0; u < MPB_A_GRID_ROUNDS; ++u)
                if (u < g_rounds)
                    __builtin_amdgcn_global_load_lds(g4 + threadIdx.x + NTHR * u,
                                                     (__attribute__((address_space(3))) unsigned*)(gridw + 4 * (NTHR * u + (threadIdx.x & ~63u))),
                                                     16, 0, 0);
            if ((int)threadIdx.x < g_nsph) oreg = reinterpret_cast<const float4*>(G0.sph)[threadIdx.x];
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware block -> rollout map (speed only, any map is correct): workgroups are dealt round-robin
    // over the 8 XCDs, so blocks with equal blockIdx % 8 share an L2.  All S/4 blocks of particle p are
    // given blockIdx % 8 == p % 8; the update kernel's block p lands on the same XCD and finds the samples
    // this kernel wrote still resident in that L2 instead of fetching them across the fabric.
    int lb = blockIdx.x;
    {
        const int nb = S / MPB_A_WPB;                // blocks per particle
        if ((S % MPB_A_WPB) == 0 && (P & 7) == 0) {
            const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
            lb = (8 * (q / nb) + x) * nb + (q % nb); // particle 8*(q/nb)+x, its (q%nb)-th block
        }
    }
    const int r = lb * MPB_A_WPB + wave;  // rollout index
    const bool live = r < P * S;
    const int p = live ? r / S : 0, s = live ? r - p * S : 0;
    const int j = lane & 15, g = lane >> 4;
    // the lane's row of the particle mean (lane = waypoint further down): fetched now, used after the noise product
    // (the generic instantiation, which holds two evaluators, fetches it where it is used: it has no registers to spare)
    float mu[DCH];
    if (MODEL != 0) {
        const float* mrow = means + ((size_t)p * H + lane) * DCH;
        if (DCH % 2 == 0) {
#pragma unroll
            for (int c = 0; c < DCH; c += 2) {
                const float2 mv = *reinterpret_cast<const float2*>(mrow + c);
                mu[c] = mv.x; mu[c + 1] = mv.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < DCH; ++c) mu[c] = mrow[c];
        }
    }

    MPB_STAMP(1);
    // ---- eps[c = j][k = 32 kb + 8 g + e] of both column blocks, drawn (or loaded) BEFORE L is needed
    float ev[2][8];
    uint32_t carry[2] = {0u, 0u};
    const float* eps_s = eps ? eps + (size_t)s * DCH * P * H : nullptr;
    stomp_eps8<DCH, 0, STOMP_PRIO_NONE>(ev[0], carry, eps_s, P, p, j, g, particle_offset + (uint32_t)p, (uint32_t)s, iter, seed_lo, seed_hi);
    stomp_eps8<DCH, 1, STOMP_PRIO_NONE>(ev[1], carry, eps_s, P, p, j, g, particle_offset + (uint32_t)p, (uint32_t)s, iter, seed_lo, seed_hi);
    MPB_STAMP(2);
    // L as the three-component bf16 MFMA image (mpb_stomp_noise.h)
    unsigned* Limg = reinterpret_cast<unsigned*>(Lp);
#pragma unroll
    for (int u = 0; u < L_PER_THREAD; ++u) {
        const int v4 = threadIdx.x + NTHR * u;
        stomp_l_image_store(Limg, v4 >> 4, (v4 & 15) << 2, lreg[u]);
    }
    if (WITH_COST && grid0 && (int)threadIdx.x <= g_nsph && threadIdx.x <= MPB_GRID_MAX_SPH)
        otab[threadIdx.x] = oreg;                                                            // entry n_sph: the far dummy
    __syncthreads();
    // ---- N = L * eps on the matrix cores (mpb_stomp_noise.h: the product the persistent kernel runs too, same bits)
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        StompEps8 b;
        stomp_split_product<0>(Limg, ev[0], b, eps_s != nullptr, j, g, acc);
        stomp_split_product<1>(Limg, ev[1], b, eps_s != nullptr, j, g, acc);
    }
    MPB_STAMP(3);
    __syncthreads();  // every wave has read its A operands: the L image is dead, its space becomes the wave tiles
    MPB_STAMP(4);
    float* nt = Lp + wave * (H * NT_STRIDE);
    // D[row = 4g + rr][col = j] -> noise tile [waypoint][channel]; written by lane = channel and read back by lane =
    // waypoint of the SAME wave (LDS operations of a wave complete in order: no barrier)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) nt[(16 * m + 4 * g + rr) * NT_STRIDE + j] = acc[m][rr];
    __builtin_amdgcn_wave_barrier();
    // ---- lane = waypoint h
    const int h = lane;
    float nz[16];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(nt + h * NT_STRIDE);
#pragma unroll
        for (int v = 0; v < (DCH + 3) / 4; ++v) {
            const f32x4 t = row[v];
            nz[4 * v + 0] = t[0]; nz[4 * v + 1] = t[1]; nz[4 * v + 2] = t[2]; nz[4 * v + 3] = t[3];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bool edge = (h == 0) || (h == H - 1);
    if (MODEL == 0) {
        const float* mrow = means + ((size_t)p * H + h) * DCH;
#pragma unroll
        for (int c = 0; c < DCH; ++c) mu[c] = mrow[c];
    }
    float x[DCH];
#pragma unroll
    for (int c = 0; c < DCH; ++c) x[c] = mu[c] + (edge ? 0.f : nz[c]);
    // sample tile out through LDS: the wave's (64 x DCH) tile is contiguous in HBM, so pack it in the
    // (already consumed) noise tile and write 16-byte lanes, 1 KB per store instruction
    {
        float* pk = nt;
#pragma unroll
        for (int c = 0; c < DCH; ++c) pk[h * DCH + c] = x[c];
        __builtin_amdgcn_wave_barrier();
        const f32x4* pk4 = reinterpret_cast<const f32x4*>(pk);
        f32x4* out4 = reinterpret_cast<f32x4*>(samples + ((size_t)p * S + s) * H * DCH);
#pragma unroll
        for (int k = 0; k < (16 * DCH + 63) / 64; ++k) {
            const int idx = lane + 64 * k;
            if (idx < 16 * DCH && live) out4[idx] = pk4[idx];
        }
    }
    MPB_STAMP(5);
    if (WITH_COST) {
        float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
        // (a row of d channels holds D = d or D = d / 2 joint positions: with more channels than MPB_MAX_DOF it must be d / 2 -- the
            // joints beyond that are zeros the compiler can fold, which keeps the d = 14 kernels at the registers they had with 8)
            constexpr int DQ_ = (DCH > MPB_MAX_DOF) ? DCH / 2 : DCH;
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < DQ_) ? x[i < DQ_ ? i : 0] : 0.f;
        float c = 0.f;
        bool bad = false;   // geom_flags and the device header disagree: the cost is poisoned (NaN bits), never mis-read
        // one pass per chained collision field (the reference sums one CostCollision per field); the first field's
        // grid is already in LDS, a later field's replaces it (all waves of the block take the same branches: G is
        // block-uniform)
        GeomView G = G0;
        for (const float* gp = geom;;) {
            if (grid_usable(G)) {
                if (gp != geom) {
                    __syncthreads();
                    grid_stage(G, gridw, otab, threadIdx.x, NTHR);
                    __syncthreads();
                }
                MPB_STAMP(6);
                if (live && h >= 1) {
                    if (MODEL == PandaModel::ID) {
                        // the launcher picked this instantiation from the caller's geom_flags; the device header has the
                        // last word: a buffer that is not tagged with the model poisons the cost instead of being mis-read
                        // (`bad` is set below, wave-uniformly: lane 0 -- waypoint 0, outside the walk -- writes the cost)
                        if (G.model == PandaModel::ID) c = fmaf(G.fscale, waypoint_cost_grid_model<PandaModel>(G, gridw, otab, q), c);
                    } else {
                        c = fmaf(G.fscale, waypoint_cost_grid(G, gridw, otab, q), c);
                    }
                }
                if (MODEL != 0 && G.model != MODEL) bad = true;
            } else if (MODEL != 0) {
                bad = true;                                  // model instantiations are only launched for grid-backed fields
            } else if (live && h >= 1) {
                c = fmaf(G.fscale, waypoint_cost<false>(G, q, dq), c);
            }
            if (G.next == 0) break;
            gp += G.next;
            G = geom_view(gp);
        }
        const double csum = wave_sum_f64((double)c);
        if (live && lane == 0) {
            if (bad) reinterpret_cast<unsigned*>(costs)[r] = 0x7FC00000u;   // (the build is -ffinite-math-only: no NaN arithmetic)
            else costs[r] = weight * (k_sigma * (float)csum);
        }
    }
    MPB_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// STOMP kernel A for any other horizon H <= 64*M (M = 1, 2, 4): the same wave-per-rollout / lane-per-waypoint
// scheme with the horizon cut in 64-waypoint chunks.  noise[hc] = sum_{kc <= hc} L[hc][kc] * eps[kc] over
// the 64 x 64 blocks of the lower-triangular scale_tril, each block product on the matrix cores exactly like
// the H = 64 kernel (blocks are staged one at a time in the same 16 KB permuted LDS image, zero-padded past
// H); the channel count is a run-time argument (the MFMA B operand always carries 16 columns).
// ------------------------------------------------------------------------------------------------
template <bool WITH_COST, int M>
__global__ __launch_bounds__(256) void stomp_sample_cost_hx_kernel(
    const float* __restrict__ means, const float* __restrict__ eps, float* __restrict__ samples,
    float* __restrict__ costs, const float* __restrict__ Lmat, const float* __restrict__ geom,
    int P, int S, int H, int d, float k_sigma, float weight, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter,
    uint32_t particle_offset) {
    __shared__ __attribute__((aligned(16))) unsigned Limg[STOMP_LIMG_WORDS_FULL];   // one block of L as three bf16 components
    __shared__ __attribute__((aligned(16))) float Nt[4][64 * NT_STRIDE];
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    const bool live = r < P * S;
    const int p = live ? r / S : 0, s = live ? r - p * S : 0;
    const int j = lane & 15, g = lane >> 4;
    const int Mc = (H + 63) >> 6;                                   // chunks in use (block-uniform)
    float* nt = Nt[wave];
    f32x4 acc[M][4];
#pragma unroll
    for (int hc = 0; hc < M; ++hc)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[hc][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < M; ++kc) {
        if (kc < Mc) {
            // ---- eps[c = j][k = 64 kc + 32 kb + 8 g + e] of this column chunk as the three bf16 operand components of its
            //      two column blocks (mpb_stomp_noise.h; device noise: Philox calls 2 kb, 2 kb + 1 of the chunk)
            StompEps8 sp[2];
            uint32_t carry[2] = {0u, 0u};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float v[8];
                if (eps != nullptr) {
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int k = 64 * kc + 32 * kb + 8 * g + e8;
                        v[e8] = (j < d && k < H) ? eps[(((size_t)s * d + j) * P + p) * H + k] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = 0.f;
                    if (j < d) {
                        if (kb == 0)
                            stomp_normals_lo<STOMP_PRIO_NONE>(particle_offset + (uint32_t)p, (uint32_t)s, (uint32_t)j, (uint32_t)g,
                                                              (uint32_t)(kc << 4), iter, seed_lo, seed_hi, v, carry);
                        else
                            stomp_normals_hi<STOMP_PRIO_NONE>(particle_offset + (uint32_t)p, (uint32_t)s, (uint32_t)j, (uint32_t)g,
                                                              (uint32_t)(kc << 4), iter, seed_lo, seed_hi, carry, v);
                    }
                }
                if (eps != nullptr) stomp_split8<true>(v, sp[kb]);
                else stomp_split8<false>(v, sp[kb]);           // (drawn normals: two components, mpb_stomp_noise.h)
            }
#pragma unroll
            for (int hc = kc; hc < M; ++hc) {
                if (hc < Mc) {
                    // ---- stage block (hc, kc) of L as its MFMA image (zero past H); a diagonal block only has its lower triangle
                    __syncthreads();
                    for (int v4 = threadIdx.x; v4 < 64 * 16; v4 += 256) {
                        const int row = v4 >> 4, col0 = (v4 & 15) << 2;
                        const int gr = 64 * hc + row, gc = 64 * kc + col0;
                        f32x4 lv;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) lv[e4] = (gr < H && gc + e4 < H) ? Lmat[(size_t)gr * H + gc + e4] : 0.f;
                        if (hc == kc) stomp_l_image_store<false>(Limg, row, col0, lv);
                        else stomp_l_image_store<true>(Limg, row, col0, lv);
                    }
                    __syncthreads();
                    if (eps != nullptr) {
                        if (hc == kc) {
                            stomp_noise_product_kb<0, false>(Limg, sp[0], j, g, acc[hc]);
                            stomp_noise_product_kb<1, false>(Limg, sp[1], j, g, acc[hc]);
                        } else {
                            stomp_noise_product_kb<0, true>(Limg, sp[0], j, g, acc[hc]);
                            stomp_noise_product_kb<1, true>(Limg, sp[1], j, g, acc[hc]);
                        }
                    } else if (hc == kc) {
                        stomp_noise_product_kb<0, false, false, false>(Limg, sp[0], j, g, acc[hc]);
                        stomp_noise_product_kb<1, false, false, false>(Limg, sp[1], j, g, acc[hc]);
                    } else {
                        stomp_noise_product_kb<0, true, false, false>(Limg, sp[0], j, g, acc[hc]);
                        stomp_noise_product_kb<1, true, false, false>(Limg, sp[1], j, g, acc[hc]);
                    }
                }
            }
        }
    }
    // ---- per chunk: transpose through the wave's tile, add the mean, store the samples, keep the positions
    float q[M][MPB_MAX_DOF];
    const bool vec_store = ((H * d) & 3) == 0;
#pragma unroll
    for (int hc = 0; hc < M; ++hc) {
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) q[hc][i] = 0.f;
        if (hc < Mc) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) nt[(16 * m + 4 * g + rr) * NT_STRIDE + j] = acc[hc][m][rr];
            __syncthreads();                                       // written by lane = channel, read by lane = waypoint
            const int h = 64 * hc + lane;
            const bool on = h < H;
            const bool edge = (h == 0) || (h == H - 1);
            float x[16];
            {
                const f32x4* row = reinterpret_cast<const f32x4*>(nt + lane * NT_STRIDE);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 t = row[v];
                    x[4 * v + 0] = t[0]; x[4 * v + 1] = t[1]; x[4 * v + 2] = t[2]; x[4 * v + 3] = t[3];
                }
            }
            const float* mrow = means + ((size_t)p * H + (on ? h : 0)) * d;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < d) {
                    x[c] = mrow[c] + (edge ? 0.f : x[c]);
                    if (c < MPB_MAX_DOF) q[hc][c < MPB_MAX_DOF ? c : 0] = x[c];
                }
            }
            float* sbase = samples + (((size_t)p * S + s) * H + 64 * hc) * d;
            const int nfl = (min(H - 64 * hc, 64)) * d;             // floats of this chunk
            if (vec_store) {
                // pack the (rows x d) chunk contiguously in the consumed tile, write 16-byte lanes
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < d && on) nt[lane * d + c] = x[c];
                const f32x4* pk4 = reinterpret_cast<const f32x4*>(nt);
                f32x4* out4 = reinterpret_cast<f32x4*>(sbase);
                for (int idx = lane; 4 * idx < nfl; idx += 64)
                    if (live) out4[idx] = pk4[idx];
            } else if (on && live) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < d) sbase[lane * d + c] = x[c];
            }
            __syncthreads();                                       // tile reads done before the next chunk overwrites it
        }
    }
    if (WITH_COST) {
        float c = 0.f;
        static_assert(STOMP_LIMG_WORDS_FULL >= MPB_GRID_MAX_CELLS, "the grid reuses the L image");
        unsigned* gridw = Limg;                                  // (the L image is dead by now)
        for (const float* gp = geom; gp != nullptr; gp = geom_next(gp)) {
            const GeomView G = geom_view(gp);
            const bool use_grid = grid_usable(G);
            if (use_grid) {
                __syncthreads();
                grid_stage(G, gridw, otab, threadIdx.x, 256);
                __syncthreads();
            }
#pragma unroll
            for (int hc = 0; hc < M; ++hc) {
                const int h = 64 * hc + lane;
                if (hc < Mc && live && h >= 1 && h < H) {
                    float dq[MPB_MAX_DOF];
                    c = fmaf(G.fscale, use_grid ? waypoint_cost_grid(G, gridw, otab, q[hc]) : waypoint_cost<false>(G, q[hc], dq), c);
                }
            }
        }
        const double csum = wave_sum_f64((double)c);
        if (live && lane == 0) costs[r] = weight * (k_sigma * (float)csum);
    }
}

// ------------------------------------------------------------------------------------------------
// STOMP kernel B: one workgroup per particle.
//   w = softmax(-c/T) over S (stomp.py:219-220); delta = sum_s w_s (sample_s - mean);
//   mean += (lr*Sigma) @ delta (stomp.py:207-211; `lr * Sigma @ x` binds as (lr*Sigma) @ x).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stomp_update_kernel(
    float* __restrict__ means, const float* __restrict__ samples, const float* __restrict__ costs,
    float* __restrict__ weights, const float* __restrict__ Sigma, int P, int S, int H, int d, float lr,
    float temperature, int sigma_in_lds) {
    extern __shared__ float lds[];
    const int n = H * d;
    const int SG = (S >= 4) ? 4 : 1;  // sample groups reduced in parallel, combined in fixed order
    float* w_lds = lds;               // S
    float* delta = w_lds + S;         // n
    float* part = delta + n;          // SG * n
    float* sig = part + SG * n;       // H*H when staged
    __shared__ float red[16];
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    if (sigma_in_lds && Sigma != nullptr)
        for (int i = tid; i < H * H; i += blockDim.x) sig[i] = lr * Sigma[i];
    // ---- softmax over S
    float m = -3.0e38f;
    for (int s = tid; s < S; s += blockDim.x) {
        const float x = -costs[(size_t)p * S + s] / temperature;
        w_lds[s] = x;
        m = fmaxf(m, x);
    }
    m = wave_max_f32(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    float z = 0.f;
    for (int s = tid; s < S; s += blockDim.x) {
        const float e = expf(w_lds[s] - m);
        w_lds[s] = e;
        z += e;
    }
    z = wave_sum_f32(z);
    if (lane == 0) red[wave] = z;
    __syncthreads();
    z = 0.f;
    for (int i = 0; i < nw; ++i) z += red[i];
    for (int s = tid; s < S; s += blockDim.x) {
        const float w = w_lds[s] / z;
        w_lds[s] = w;
        weights[(size_t)p * S + s] = w;
    }
    __syncthreads();
    // ---- weighted noise reduce: (element, sample-group) work items, coalesced over the H*d row
    for (int idx = tid; idx < n * SG; idx += blockDim.x) {
        const int sg = idx / n, i = idx - sg * n;
        const int s0 = (sg * S) / SG, s1 = ((sg + 1) * S) / SG;
        const float mu = means[(size_t)p * n + i];
        const float* sp = samples + ((size_t)p * S) * n + i;
        float acc = 0.f;
#pragma unroll 8
        for (int s = s0; s < s1; ++s) acc += w_lds[s] * (sp[(size_t)s * n] - mu);
        part[idx] = acc;
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        float acc = part[i];
        for (int sg = 1; sg < SG; ++sg) acc += part[sg * n + i];
        delta[i] = acc;
    }
    __syncthreads();
    // ---- covariance-weighted step: mean += (lr*Sigma) @ delta
    for (int i = tid; i < n; i += blockDim.x) {
        const int h = i / d, c = i - h * d;
        float acc = 0.f;
        if (Sigma == nullptr) {
            acc = lr * delta[i];      // update without the covariance product (StochGPMP, stoch_gpmp.py:272-275)
        } else if (sigma_in_lds) {
            for (int k = 0; k < H; ++k) acc = fmaf(sig[h * H + k], delta[k * d + c], acc);
        } else {
            for (int k = 0; k < H; ++k) acc = fmaf(lr * Sigma[h * H + k], delta[k * d + c], acc);
        }
        means[(size_t)p * n + i] += acc;
    }
}

// Vectorised variant of kernel B for H*d divisible by 4, H*d <= 1024, H <= 64 and S <= 64, built for a
// short critical path (the kernel is pure latency: ~900 VALU instructions per wave):
//   * every load is issued in the first instructions: 8 float4 sample loads per thread (they do not
//     depend on the weights), the thread's float4 of Sigma (staged in LDS with padded rows) and the costs;
//   * the softmax over S <= 64 is done redundantly by every wave with wave reductions -- no barriers;
//   * two barriers in total (partial sums -> delta -> matvec); delta is kept transposed in LDS so the matvec reads
//     float4s, and runs four independent accumulators with the learning rate applied once (8.2 -> 7.6 us).
#define UPD_LD 68   // padded row of the transposed delta tile (floats): 16-byte aligned, channels spread over the LDS banks
__global__ __launch_bounds__(1024) void stomp_update_v4_kernel(
    float* __restrict__ means, const float* __restrict__ samples, const float* __restrict__ costs,
    float* __restrict__ weights, const float* __restrict__ Sigma, int P, int S, int H, int d, float lr,
    float temperature) {
    extern __shared__ float lds[];
    const int n = H * d, n4 = n >> 2;
    const int SG = min(4, 1024 / n4);                 // sample groups
    const int spg = (S + SG - 1) / SG;                // samples per group
    float* delta = lds;                               // d rows of UPD_LD floats (transposed, padded)
    float4* part = reinterpret_cast<float4*>(delta + d * UPD_LD);  // SG * n4 float4
    float* sig_l = reinterpret_cast<float*>(part + SG * n4);       // H rows of UPD_LD floats
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int sg = tid / n4, i4 = tid - sg * n4;
    const bool worker = sg < SG;
    const int s0 = sg * spg, s1 = min(S, s0 + spg);
    const float4* smp4 = reinterpret_cast<const float4*>(samples) + (size_t)p * S * n4 + i4;
    // ---- issue every load up front.  Rows past the group's last sample are loaded from a clamped (valid) index and
    //      never used: selecting between the load and a register copy of `mu` made hipcc spill `mu` to scratch and
    //      load through a flat pointer (32 B / lane of scratch in a 6 us latency kernel)
    float4 v[8];
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
    if (worker) {
        mu = reinterpret_cast<const float4*>(means)[(size_t)p * n4 + i4];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = smp4[(size_t)min(s0 + k, S - 1) * n4];
    }
    const float cst = (lane < S) ? costs[(size_t)p * S + lane] : 0.f;
    // the matvec output this thread owns (thread tid < n <-> element (h, c)); Sigma (H x H, the same for every block)
    // is staged once in LDS with padded rows: one coalesced float4 per thread (H = 64) instead of a 64-float row per
    // thread through the L1 (229 KB per block: 1.3 us of the kernel, measured by elimination)
    const int hh = (tid < n) ? tid / d : 0, cc = (tid < n) ? tid - hh * d : 0;
    float4 sg4[(64 * 64 / 4 + 1023) / 1024];
#pragma unroll
    for (int u = 0; u < (64 * 64 / 4 + 1023) / 1024; ++u) {
        const int v4 = tid + 1024 * u;
        sg4[u] = (Sigma != nullptr && v4 < H * H / 4) ? reinterpret_cast<const float4*>(Sigma)[v4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- softmax over S (every wave, redundantly): w for sample `lane`
    const float xs = (lane < S) ? -cst / temperature : -3.0e38f;
    const float mx = wave_max_f32(xs);
    const float ex = (lane < S) ? expf(xs - mx) : 0.f;
    const float wl = ex / wave_sum_f32(ex);
    if (tid < S) weights[(size_t)p * S + tid] = wl;
#pragma unroll
    for (int u = 0; u < (64 * 64 / 4 + 1023) / 1024; ++u) {
        const int v4 = tid + 1024 * u;
        if (v4 < H * H / 4) {
            const int row = (4 * v4) / H, col = 4 * v4 - row * H;
            *reinterpret_cast<float4*>(sig_l + row * UPD_LD + col) = sg4[u];
        }
    }
    // ---- weighted noise reduce; the weight of sample s is broadcast from lane s.  The broadcasts run with ALL 64 lanes
    //      active (outside the `worker` branch, uniform trip counts): a wave straddling the worker boundary (H*d not a
    //      multiple of 64) would otherwise shuffle from EXEC-disabled lanes, which return 0 on gfx950
    float wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wk[k] = __shfl(wl, min(s0 + k, 63), 64);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (worker) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (s0 + k < s1) {
                acc.x += wk[k] * (v[k].x - mu.x); acc.y += wk[k] * (v[k].y - mu.y);
                acc.z += wk[k] * (v[k].z - mu.z); acc.w += wk[k] * (v[k].w - mu.w);
            }
        }
    }
    for (int k = 8; k < spg; ++k) {                      // S > 32: the rest of the group (block-uniform trip count)
        const int s = s0 + k;
        const float w = __shfl(wl, min(s, 63), 64);
        if (worker && s < s1) {
            const float4 t = smp4[(size_t)s * n4];
            acc.x += w * (t.x - mu.x); acc.y += w * (t.y - mu.y);
            acc.z += w * (t.z - mu.z); acc.w += w * (t.w - mu.w);
        }
    }
    if (worker) part[sg * n4 + i4] = acc;
    __syncthreads();
    if (tid < n4) {
        float4 a = part[tid];
        for (int g = 1; g < SG; ++g) {
            const float4 b = part[g * n4 + tid];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        // delta goes to LDS TRANSPOSED, channel-major with a padded row (UPD_LD floats): the matvec below then reads
        // four consecutive waypoints of its channel with one ds_read_b128
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = 4 * tid + e, hq = idx / d, cq = idx - hq * d;
            delta[cq * UPD_LD + hq] = av[e];
        }
    }
    __syncthreads();
    // ---- covariance-weighted step: mean += lr * (Sigma @ delta), Sigma row from registers; four independent partial
    //      sums (the kernel is a latency chain: one accumulator would serialise 64 dependent fma)
    if (tid < n && Sigma == nullptr) {
        // update without the covariance product (StochGPMP, stoch_gpmp.py:272-275)
        means[(size_t)p * n + tid] += lr * delta[cc * UPD_LD + hh];
    } else if (tid < n) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float4* dcol = reinterpret_cast<const float4*>(delta + cc * UPD_LD);
        const float4* srow = reinterpret_cast<const float4*>(sig_l + hh * UPD_LD);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (4 * k < H) {
                const float4 dv = dcol[k], sv = srow[k];
                acc[k & 3] = fmaf(sv.x, dv.x, acc[k & 3]);
                acc[k & 3] = fmaf(sv.y, dv.y, acc[k & 3]);
                acc[k & 3] = fmaf(sv.z, dv.z, acc[k & 3]);
                acc[k & 3] = fmaf(sv.w, dv.w, acc[k & 3]);
            }
        }
        means[(size_t)p * n + tid] += lr * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone collision cost (and gradient): one wave per trajectory, lane = waypoint.
// ------------------------------------------------------------------------------------------------
template <bool GRAD, int MODEL = 0>
__global__ __launch_bounds__(256) void collision_cost_kernel(
    const float* __restrict__ trajs, const float* __restrict__ geom, float* __restrict__ out,
    float* __restrict__ per_wp, float* __restrict__ grad, int B, int H, int d, int h_begin, float k_sigma,
    float weight) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int D = reinterpret_cast<const int*>(geom)[3];
    __shared__ unsigned gridw[MPB_GRID_MAX_CELLS];
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const bool dead = b >= B;                                  // such waves still take part in the block barriers
    double csum = 0.0;
    // one pass over the trajectory per chained field (its broad-phase grid is staged in LDS); outputs of the
    // second and later fields are added onto what the first one wrote
    for (const float* gp = geom; gp != nullptr; gp = geom_next(gp)) {
        const GeomView G = geom_view(gp);
        const bool use_grid = GRAD ? grid_usable_grad(G) : grid_usable(G);
        const bool first = gp == geom;
        __syncthreads();
        if (use_grid) grid_stage(G, gridw, otab, threadIdx.x, blockDim.x);
        __syncthreads();
        if (dead) continue;
        for (int h = lane; h < ((H + 63) & ~63); h += 64) {
            float c = 0.f;
            if (h < H) {
                const float* row = trajs + ((size_t)b * H + h) * d;
                float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
                load_row_prefix<MPB_MAX_DOF>(row, D, (d & 1) == 0, q);
#pragma unroll
                for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
                if (h >= h_begin) {
                    if (GRAD && MODEL == PandaModel::ID) {     // compile-time model (launcher: geom_flags); tag re-checked here
                        if (use_grid && G.model == PandaModel::ID) {
                            c = G.fscale * waypoint_cost_grid_grad_model<PandaModel>(G, gridw, otab, q, dq);
                        } else {
                            c = __uint_as_float(0x7FC00000u);
#pragma unroll
                            for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = c;
                        }
                    } else if (GRAD)
                        c = G.fscale * (use_grid ? waypoint_cost_grid_grad(G, gridw, otab, q, dq) : waypoint_cost<true>(G, q, dq));
                    else if (use_grid && G.model == PandaModel::ID)   // compile-time robot model (same bits as the table walk)
                        c = G.fscale * waypoint_cost_grid_model<PandaModel>(G, gridw, otab, q);
                    else
                        c = G.fscale * (use_grid ? waypoint_cost_grid(G, gridw, otab, q) : waypoint_cost<false>(G, q, dq));
                }
                if (per_wp) {
                    float* pw = per_wp + (size_t)b * H + h;
                    *pw = first ? c : *pw + c;
                }
                if (GRAD) {
                    float* grow = grad + ((size_t)b * H + h) * d;
                    const float sc = weight * k_sigma * G.fscale;
                    if ((d & 1) == 0) {        // 8-byte pieces (rows start 8-byte aligned)
                        float2* g2 = reinterpret_cast<float2*>(grow);
#pragma unroll
                        for (int i = 0; i < 2 * MPB_MAX_DOF; i += 2) {      // (rows up to positions + velocities of MPB_MAX_DOF joints)
                            if (i < d) {
                                const float g0 = (i < MPB_MAX_DOF && i < D && h >= h_begin) ? sc * dq[i < MPB_MAX_DOF ? i : 0] : 0.f;
                                const float g1 = (i + 1 < MPB_MAX_DOF && i + 1 < D && h >= h_begin) ? sc * dq[i + 1 < MPB_MAX_DOF ? i + 1 : 0] : 0.f;
                                float2 o = make_float2(g0, g1);
                                if (!first) { const float2 old = g2[i >> 1]; o.x = old.x + g0; o.y = old.y + g1; }
                                g2[i >> 1] = o;
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2 * MPB_MAX_DOF; ++i) {
                            if (i < d) {
                                const float gi = (i < MPB_MAX_DOF && i < D && h >= h_begin) ? sc * dq[i < MPB_MAX_DOF ? i : 0] : 0.f;
                                grow[i] = first ? gi : grow[i] + gi;
                            }
                        }
                    }
                }
            }
            csum += (double)c;
        }
    }
    if (dead) return;
    csum = wave_sum_f64(csum);
    if (lane == 0) out[b] = weight * (k_sigma * (float)csum);
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
// LDS budget of the update kernel: weights + delta + 4 partial tiles (+ lr*Sigma when it fits)
static bool update_lds(int S, int H, int d, size_t& bytes, int& sigma_in_lds) {
    const size_t base = ((size_t)S + 5 * (size_t)H * d) * 4;
    const size_t with_sigma = base + (size_t)H * H * 4;
    sigma_in_lds = with_sigma <= 64 * 1024;
    bytes = sigma_in_lds ? with_sigma : base;
    return bytes <= 150 * 1024;
}

// kernel B launcher: vectorised path when the (H,d) tile is float4-divisible and fits 1024 threads
static bool launch_update(float* means, const float* samples, const float* costs, float* weights, const float* Sigma,
                          int P, int S, int H, int d, float lr, float temperature, hipStream_t st) {
    const int n = H * d;
    if ((n & 3) == 0 && n <= 1024 && H <= 64 && (H & 3) == 0 && S <= 64) {   // Sigma may be NULL (no covariance product)
        const int n4 = n >> 2;
        const int SG = (1024 / n4) < 4 ? (1024 / n4) : 4;
        const size_t lds = (size_t)d * UPD_LD * 4 + (size_t)SG * n4 * 16 + (size_t)H * UPD_LD * 4;
        MPB_LAUNCH(stomp_update_v4_kernel, dim3(P), dim3(1024), lds, st, means, samples, costs, weights, Sigma, P,
                           S, H, d, lr, temperature);
        return true;
    }
    size_t lds;
    int sig_lds;
    if (!update_lds(S, H, d, lds, sig_lds)) return false;
    MPB_LAUNCH(stomp_update_kernel, dim3(P), dim3(1024), lds, st, means, samples, costs, weights, Sigma, P, S, H,
                       d, lr, temperature, sig_lds);
    return true;
}

// kernel A launcher: H = 64 takes the MFMA fast path for the channel counts of the reference's robots
template <bool WITH_COST>
static void launch_sample(const float* means, const float* eps, float* samples, float* costs, const float* L,
                          const float* geom, int geom_flags, int P, int S, int H, int d, float k_sigma, float weight,
                          uint64_t seed, uint32_t iter, uint32_t particle_offset, hipStream_t st) {
    const int B = P * S;
    const dim3 grid((B + 3) / 4), block(256);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
#define MPB_A_CASE(DCH, MODEL)                                                                                   \
    case DCH:                                                                                                    \
        MPB_LAUNCH((stomp_sample_cost_h64_kernel<DCH, WITH_COST, MODEL>), dim3((B + MPB_A_WPB - 1) / MPB_A_WPB),  \
                           dim3(64 * MPB_A_WPB), 0, st, means, eps, samples, costs, L, geom, P, S, k_sigma, weight, \
                           lo, hi, iter, particle_offset);                                                       \
        return;
    // (bit 8: every field COMPACT-grid-backed -- since round 6 the model byte is also set for list-grid scenes, which this kernel
    // serves through the exhaustive walk)
    if (H == 64 && WITH_COST && (geom_flags & 0xFF) == PandaModel::ID && (geom_flags & 0x100)) {   // the Panda's channel counts (pos_only / not)
        switch (d) {
            MPB_A_CASE(7, (WITH_COST ? PandaModel::ID : 0)) MPB_A_CASE(14, (WITH_COST ? PandaModel::ID : 0))
            default: break;
        }
    }
    if (H == 64) {
        switch (d) {
            MPB_A_CASE(2, 0) MPB_A_CASE(3, 0) MPB_A_CASE(4, 0) MPB_A_CASE(6, 0) MPB_A_CASE(7, 0) MPB_A_CASE(14, 0)
            default: break;
        }
    }
#undef MPB_A_CASE
    // any other horizon / channel count (H <= MPB_MAX_H = 256, d <= MPB_MAX_D = 16): chunked MFMA kernel
    {
#define MPB_HX(M)                                                                                                   \
    MPB_LAUNCH((stomp_sample_cost_hx_kernel<WITH_COST, M>), grid, block, 0, st, means, eps, samples, costs, L, \
                       geom, P, S, H, d, k_sigma, weight, lo, hi, iter, particle_offset)
        if (H <= 64) MPB_HX(1);
        else if (H <= 128) MPB_HX(2);
        else MPB_HX(4);
#undef MPB_HX
    }
}

static bool shape_ok(int H, int d, int D) {
    return H >= 3 && H <= MPB_MAX_H && D >= 1 && D <= MPB_MAX_DOF && (d == D || d == 2 * D);
}

extern "C" int mpb_cost_collision_eval(const float* trajs, const float* geom, float* out, float* per_waypoint,
                                       int B, int H, int d, int h_begin, float k_sigma, float weight, void* stream) {
    if (B == 0) return MPB_OK;   // empty batch: nothing to do (and torch hands out null pointers for it)
    if (!trajs || !geom || !out) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if (B < 0 || H < 1 || d < 1 || d > 2 * MPB_MAX_DOF || h_begin < 0) return fail(MPB_E_INVALID, "%s: bad shape", __func__);   // (rows of any width up to positions + velocities of 12 joints; the first n_dof channels are read)
    if (B == 0) return MPB_OK;
    hipLaunchKernelGGL((collision_cost_kernel<false, 0>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, trajs, geom,
                       out, per_waypoint, (float*)nullptr, B, H, d, h_begin, k_sigma, weight);
    return check_launch(__func__);
}

extern "C" int mpb_cost_collision_grad(const float* trajs, const float* geom, int geom_flags, float* out, float* grad, int B,
                                       int H, int d, int h_begin, float k_sigma, float weight, void* stream) {
    if (B == 0) return MPB_OK;
    if (!trajs || !geom || !out || !grad) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if (B < 0 || H < 1 || d < 1 || d > 2 * MPB_MAX_DOF || h_begin < 0) return fail(MPB_E_INVALID, "%s: bad shape", __func__);   // (rows of any width up to positions + velocities of 12 joints; the first n_dof channels are read)
    if (B == 0) return MPB_OK;
    if ((geom_flags & 0xFF) == PandaModel::ID && (geom_flags & 0x100))
        hipLaunchKernelGGL((collision_cost_kernel<true, PandaModel::ID>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, trajs,
                           geom, out, (float*)nullptr, grad, B, H, d, h_begin, k_sigma, weight);
    else
        hipLaunchKernelGGL((collision_cost_kernel<true, 0>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, trajs, geom,
                           out, (float*)nullptr, grad, B, H, d, h_begin, k_sigma, weight);
    return check_launch(__func__);
}

extern "C" int mpb_stomp_sample(const float* means, const float* eps, float* samples, const float* L,
                                const float* geom, int geom_flags, float* costs, int P, int S, int H, int d, float k_sigma,
                                float weight, uint64_t seed, uint32_t iter, uint32_t particle_offset, void* stream) {
    if (P == 0) return MPB_OK;
    if (!means || !samples || !L) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if ((geom == nullptr) != (costs == nullptr)) return fail(MPB_E_INVALID, "%s: geom and costs must be given together", __func__);
    if (P < 0 || S < 1 || H < 3 || H > MPB_MAX_H || d < 1 || d > MPB_MAX_D) return fail(MPB_E_INVALID, "%s: bad shape", __func__);
    if (mpb_misaligned16(means, eps, samples, L, geom)) return fail(MPB_E_INVALID, "%s: means / eps / samples / L / geom must be 16-byte aligned", __func__);
    if (P == 0) return MPB_OK;
    if (geom)
        launch_sample<true>(means, eps, samples, costs, L, geom, geom_flags, P, S, H, d, k_sigma, weight, seed, iter,
                            particle_offset, (hipStream_t)stream);
    else
        launch_sample<false>(means, eps, samples, nullptr, L, nullptr, 0, P, S, H, d, 0.f, 0.f, seed, iter,
                             particle_offset, (hipStream_t)stream);
    return check_launch(__func__);
}

extern "C" int mpb_stomp_update(float* means, const float* samples, const float* costs, float* weights,
                                const float* Sigma, int P, int S, int H, int d, float lr, float temperature,
                                void* stream) {
    if (P == 0) return MPB_OK;
    if (!means || !samples || !costs || !weights) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if (P < 0 || S < 1 || H < 3 || H > MPB_MAX_H || d < 1 || d > MPB_MAX_D) return fail(MPB_E_INVALID, "%s: bad shape", __func__);
    if (!(temperature > 0.f)) return fail(MPB_E_INVALID, "%s: temperature must be > 0", __func__);
    if (mpb_misaligned16(means, samples, Sigma)) return fail(MPB_E_INVALID, "%s: means / samples / Sigma must be 16-byte aligned", __func__);
    if (P == 0) return MPB_OK;
    if (!launch_update(means, samples, costs, weights, Sigma, P, S, H, d, lr, temperature, (hipStream_t)stream))
        return fail(MPB_E_UNSUPPORTED, "%s: S + H*d too large for LDS", __func__);
    return check_launch(__func__);
}

extern "C" int mpb_stomp_step(float* means, const float* eps, float* samples, float* costs, float* weights,
                              const float* L, const float* Sigma, const float* geom, int geom_flags, int P, int S, int H, int d, int D,
                              float k_sigma, float weight, float lr, float temperature, int n_iters, uint64_t seed,
                              uint32_t iter0, uint32_t particle_offset, void* stream) {
    if (P == 0) return MPB_OK;
    if (!means || !samples || !costs || !weights || !L || !Sigma || !geom) return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if (P < 0 || S < 1 || !shape_ok(H, d, D) || n_iters < 0) return fail(MPB_E_INVALID, "%s: bad shape", __func__);
    if (!(temperature > 0.f)) return fail(MPB_E_INVALID, "%s: temperature must be > 0", __func__);
    if (mpb_misaligned16(means, eps, samples, L, Sigma, geom))
        return fail(MPB_E_INVALID, "%s: means / eps / samples / L / Sigma / geom must be 16-byte aligned", __func__);
    size_t lds_b;
    int sig_lds;
    if (!update_lds(S, H, d, lds_b, sig_lds)) return fail(MPB_E_UNSUPPORTED, "%s: S + H*d too large for LDS", __func__);
    if (P == 0) return MPB_OK;
    const size_t eps_stride = (size_t)S * d * P * H;
    // Launch-queue throttle: long runs keep at most two chunks of MPB_CHUNK iterations queued ahead of the
    // GPU (before queueing chunk k+2 the host waits on an event recorded after chunk k), so the host never
    // sits on thousands of pending launches.  Short calls (<= 2 chunks) and calls made while the stream is
    // being captured into a graph never wait.
    constexpr int MPB_CHUNK = 128;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool throttle = n_iters > 2 * MPB_CHUNK &&
                          hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    for (int it = 0; it < n_iters; ++it) {
        if (throttle && it % MPB_CHUNK == 0) {
            const int k = (it / MPB_CHUNK) & 1;
            if (ev[k]) (void)hipEventSynchronize(ev[k]);                 // chunk it/MPB_CHUNK - 2 has finished
            else (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        }
        launch_sample<true>(means, eps ? eps + (size_t)it * eps_stride : nullptr, samples, costs, L, geom, geom_flags, P, S, H, d,
                            k_sigma, weight, seed, iter0 + (uint32_t)it, particle_offset, (hipStream_t)stream);
        launch_update(means, samples, costs, weights, Sigma, P, S, H, d, lr, temperature, (hipStream_t)stream);
        if (throttle && it % MPB_CHUNK == MPB_CHUNK - 1) (void)hipEventRecord(ev[(it / MPB_CHUNK) & 1], (hipStream_t)stream);
    }
    for (int k = 0; k < 2; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    return check_launch(__func__);
}

extern "C" int mpb_stomp_step_profile(float* means, float* samples, float* costs, float* weights, const float* L,
                                      const float* Sigma, const float* geom, int geom_flags, int P, int S, int H, int d, int D,
                                      float k_sigma, float weight, float lr, float temperature, int n_iters, uint64_t seed,
                                      uint32_t iter0, uint32_t particle_offset, void* stream, float* sample_kernel_ms,
                                      float* update_kernel_ms) {
    if (!means || !samples || !costs || !weights || !L || !Sigma || !geom || !sample_kernel_ms || !update_kernel_ms)
        return fail(MPB_E_INVALID, "%s: null pointer", __func__);
    if (P < 1 || S < 1 || !shape_ok(H, d, D) || n_iters < 1 || n_iters > 1024) return fail(MPB_E_INVALID, "%s: bad shape", __func__);
    if (!(temperature > 0.f)) return fail(MPB_E_INVALID, "%s: temperature must be > 0", __func__);
    size_t lds_b;
    int sig_lds;
    if (!update_lds(S, H, d, lds_b, sig_lds)) return fail(MPB_E_UNSUPPORTED, "%s: S + H*d too large for LDS", __func__);
    hipEvent_t* ev = new hipEvent_t[4 * (size_t)n_iters];
    for (int i = 0; i < 4 * n_iters; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(ev[k]);
            delete[] ev;
            return fail(MPB_E_HIP, "%s: hipEventCreate failed", __func__);
        }
    for (int it = 0; it < n_iters; ++it) {
        t_ev0 = ev[4 * it + 0]; t_ev1 = ev[4 * it + 1];
        launch_sample<true>(means, nullptr, samples, costs, L, geom, geom_flags, P, S, H, d, k_sigma, weight, seed, iter0 + (uint32_t)it,
                            particle_offset, (hipStream_t)stream);
        t_ev0 = ev[4 * it + 2]; t_ev1 = ev[4 * it + 3];
        launch_update(means, samples, costs, weights, Sigma, P, S, H, d, lr, temperature, (hipStream_t)stream);
    }
    t_ev0 = t_ev1 = nullptr;
    int rc = check_launch(__func__);
    if (rc == MPB_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = fail(MPB_E_HIP, "%s: synchronize failed", __func__);
    double sa = 0.0, sb = 0.0;
    for (int it = 0; it < n_iters && rc == MPB_OK; ++it) {
        float ma = 0.f, mb = 0.f;
        if (hipEventElapsedTime(&ma, ev[4 * it + 0], ev[4 * it + 1]) != hipSuccess ||
            hipEventElapsedTime(&mb, ev[4 * it + 2], ev[4 * it + 3]) != hipSuccess)
            rc = fail(MPB_E_HIP, "%s: hipEventElapsedTime failed", __func__);
        sa += ma;
        sb += mb;
    }
    for (int i = 0; i < 4 * n_iters; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    if (rc != MPB_OK) return rc;
    *sample_kernel_ms = (float)(sa / n_iters);
    *update_kernel_ms = (float)(sb / n_iters);
    return MPB_OK;
}

