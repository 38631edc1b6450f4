// mpb_geom.h -- device-side geometry: packed buffer view, forward kinematics of the collision spheres,
// sphere / box signed distance, hinge cost and its analytic gradient.
//
// Build-defined back-end standing in for torch_robotics (reference call sites
// cost_functions.py:50-52 robot.fk_map_collision, field_factor.py:39 field.compute_cost); the same
// definitions are evaluated on the CPU by oracle/geometry_ref.py for checking.
//
// Mapping to the hardware: one LANE evaluates one waypoint; every geometry word (joint transforms,
// sphere offsets, obstacle constants) is addressed with wave-uniform indices through a const
// __restrict__ kernel-argument pointer, so hipcc turns those reads into scalar loads (s_load_dwordx8/16)
// and the values ride in SGPRs as VALU operands -- no LDS traffic and no VGPRs for constants.
//
// Obstacle loop: the collision spheres of one kinematic frame (<= 8 at a time) are held in VGPRs and
// tested against blocks of 4 obstacle spheres held in SGPRs.  The per-pair test is conservative and
// cheap (3 fma + 1 compare on |x|^2 - 2 x.c < rhs_o, table built on the host in fp64); the exact
// distance (3 sub, 3 fma, v_sqrt, ...) is evaluated only when some lane of the wave passes it, so the
// result is bit-identical to evaluating every pair while the common far-away pair costs 4 VALU slots
// instead of ~12 (v_sqrt_f32 alone is 4: quarter rate, measured in scripts/microbench_valu.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

#include "mpb_model_panda.h"

#define MPB_GEOM_MAGIC 0x4D504247
#define MPB_GEOM_VERSION 6
#define MPB_GEOM_VERSION_LIST 7  // a field whose grid section is a LIST grid (round 6: scenes beyond the compact grid's 63 spheres; geometry.py build_list_grid)
#define MPB_LIST_MAX_SPH 255     // the list evaluators' sphere table (LDS): 255 + the far dummy
#define MPB_LIST_MAX_BOX 127     // ... and box table: 127 + the far dummy
#define MPB_LIST_MAX_CAND 16384  // bytes of candidate indices (LDS)
#define MPB_LIST_CELL_MAX_SPH 126
#define MPB_LIST_CELL_MAX_BOX 62
#define MPB_MAX_FIELDS 4   // collision fields chained in one buffer (header word 27 = words to the next one)
#define MPB_GEOM_HEADER_WORDS 32
#define MPB_GRID_MAX_CELLS 4096
#define MPB_GRID_PAD 1024        // the grid section of a buffer is padded to a multiple of this many words
#define MPB_GRID_MAX_SPH 63      // obstacle table in LDS: 63 spheres + one far-away dummy
#define MPB_GRID_OVERFLOW 0xFFFFFFFEu   // a cell word packs four 8-bit obstacle indices; an unused slot holds n_sph (the far dummy)
#define MPB_KIND_POINT 0
#define MPB_KIND_CHAIN 1
#define MPB_MAX_DOF 12
#define MPB_MAX_TF (MPB_MAX_DOF + 1)
// collision spheres processed together (VGPR resident) by the exhaustive obstacle loop: 4.  (The cost-only path ran 8 at
// a time in round 1; that evaluator is now the cold path -- fields without a usable broad-phase grid -- and at 8 it alone
// pushed STOMP kernel A, which also holds the grid evaluator, over its 128 registers: 64 B / lane of scratch.)
#ifndef MPB_LCH_COST
#define MPB_LCH_COST 4
#endif
#define MPB_LCH_OF(GRAD) ((GRAD) ? 4 : MPB_LCH_COST)
// conservative-test policy: 0 = adaptive (default), 1 = never test, 2 = always test (tuning builds only)
#ifndef MPB_CULL_MODE
#define MPB_CULL_MODE 0
#endif

struct GeomView {
    int kind, n_dof, n_tf, n_links, n_sph, n_box;
    float margin;
    const float* tf;     // n_tf x 12
    const float* links;  // n_links x 8: frame(int), ox, oy, oz, r, 0,0,0
    const float* sph;    // n_sph x 4
    const float* box;    // n_box x 8: cx,cy,cz,0,hx,hy,hz,0
    const float* cull;   // ceil4(n_sph) x 8: -2cx,-2cy,-2cz,rhs, cx,cy,cz,r
    const int* fstart;   // links [fstart[j], fstart[j+1]) ride on frame j+1
    // broad-phase grid over the inflated obstacle spheres (n_cells == 0: none)
    const unsigned* grid;
    int gnx, gny, gnz, n_cells;
    float glx, gly, glz, gix, giy, giz;  // origin, 1 / cell size (version 6: on a lattice through 0 -- lo = (K - 1/2) h per axis)
    int k_lin;                           // linear index of the lattice point the grid starts at: cell = round(x / h) - K per axis
    float fscale;                        // s_f: this field's share in  sum_f s_f * cost_f
    int next;                            // words from this header to the next chained field (0: last)
    int version;                         // MPB_GEOM_VERSION (compact grid) or MPB_GEOM_VERSION_LIST
    const unsigned char* cand;           // (list grid) candidate indices, behind the padded cell words
    int n_cand;                          // ... bytes of them
    int model;                           // compile-time robot model the tables equal bit for bit (0: none), mpb_model_*.h
    unsigned keep_mask;                  // bit l: the model's collision sphere l is in the link table (static pruning)
};

// next field of the chain (the reference sums one CostCollision per field), nullptr after the last one
__device__ __forceinline__ const float* geom_next(const float* __restrict__ g) {
    const int off = reinterpret_cast<const int*>(g)[27];
    return off ? g + off : nullptr;
}

__device__ __forceinline__ GeomView geom_view(const float* __restrict__ g) {
    const int* gi = reinterpret_cast<const int*>(g);
    GeomView v;
    v.kind = gi[2];
    v.n_dof = gi[3];
    v.n_tf = gi[4];
    v.n_links = gi[5];
    v.n_sph = gi[6];
    v.n_box = gi[7];
    v.margin = g[8];
    v.tf = g + gi[9];
    v.links = g + gi[10];
    v.sph = g + gi[11];
    v.box = g + gi[12];
    v.cull = g + gi[14];
    v.fstart = gi + gi[15];
    v.grid = reinterpret_cast<const unsigned*>(g) + gi[16];
    v.gnx = gi[17]; v.gny = gi[18]; v.gnz = gi[19];
    v.glx = g[20]; v.gly = g[21]; v.glz = g[22];
    v.gix = g[23]; v.giy = g[24]; v.giz = g[25];
    v.n_cells = gi[26];
    v.version = gi[1];
    {
        const int off_cand = gi[16] + (gi[26] + MPB_GRID_PAD - 1) / MPB_GRID_PAD * MPB_GRID_PAD;
        v.cand = reinterpret_cast<const unsigned char*>(g + off_cand);
        v.n_cand = 4 * (gi[13] - off_cand);
    }
    v.k_lin = gi[31];
    v.fscale = g[28];
    v.next = gi[27];
    v.model = gi[29];
    v.keep_mask = (unsigned)gi[30];
    return v;
}

// sin and cos together, |x| up to a few thousand.  Round 4: reduction by PI (three-term Cody-Waite, fma) to r in
// [-pi/2, pi/2] and minimax polynomials on that interval (fitted for this file: odd degree 11 / even degree 8 in r,
// scripts/sincos_fit.py), so that the quadrant logic is ONE sign, (-1)^k on both values, applied as an xor with the low
// bit of k -- which the round-to-nearest of the reduction delivers for free (magic-number add: the integer sits in the low
// mantissa bits).  21 VALU instructions for both values, none of them a compare / select (the pi/2 form with the cephes
// polynomials on [-pi/4, pi/4] took 29, eight of them half-rate compares, selects and converts: 7 x 8 per waypoint).
// Max abs error against fp64 over |x| <= 20: sin 1.16e-7, cos 1.28e-7 (rms 2.1e-8 / 3.2e-8; the pi/2 form: 9.1e-8 / 9.2e-8,
// rms 2.1e-8 -- one ulp either way; emulated in scripts/sincos_fit.py, measured on the device by scripts/sincos_accuracy.hip).
// ocml sinf + cosf cost ~4x that because of their huge-argument path.  Joint angles are bounded by the limits plus STOMP noise.
#ifdef MPB_SINCOS_PI2   // the former pi/2 form, kept for A/B measurements (scripts/ab_k20.sh)
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.6366197466850281f);
    float r = fmaf(-k, 1.5707963705062866f, x);
    r = fmaf(-k, -4.371138828673793e-08f, r);
    r = fmaf(-k, -1.7763568394002505e-15f, r);
    const float r2 = r * r;
    const float s = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float c = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                         fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float a = (q & 1) ? c : s;
    const float b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
#else
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cs) {
    const float t = fmaf(x, 0.3183098861837907f, 12582912.0f);        // 1.5 * 2^23 + rint(x / pi): valid for |x| < 1e6
    const float k = t - 12582912.0f;
    float r = fmaf(-k, 3.1415927410125732f, x);
    r = fmaf(-k, -8.742277657347586e-08f, r);
    r = fmaf(-k, -3.552713678800501e-15f, r);
    const float u = r * r;
    const float s = fmaf(r * u, fmaf(u, fmaf(u, fmaf(u, fmaf(u, -2.4080563321e-08f, 2.7536482321e-06f), -1.9841086760e-04f),
                                             8.3333328366e-03f), -1.6666667163e-01f), r);
    const float c = fmaf(u * u, fmaf(u, fmaf(u, fmaf(u, -2.6546715048e-07f, 2.4786108042e-05f), -1.3888812391e-03f), 4.1666667908e-02f),
                         fmaf(-0.5f, u, 1.0f));
    const unsigned sg = __float_as_uint(t) << 31;                      // parity of k
    sn = __uint_as_float(__float_as_uint(s) ^ sg);
    cs = __uint_as_float(__float_as_uint(c) ^ sg);
}
#endif

// v_sqrt_f32: 1 ulp, one quarter-rate instruction (the IEEE-correct expansion hipcc emits for sqrtf is
// ~20 VALU instructions).  Arguments are squared distances, far inside the range that needs no scaling.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// State of N collision spheres (VGPR resident) while the obstacle set streams past them in SGPRs.
template <bool GRAD>
struct LinkChunk {
    static constexpr int N = MPB_LCH_OF(GRAD);
    float x[N], y[N], z[N], xx[N], best[N];
    float vx[GRAD ? N : 1], vy[GRAD ? N : 1], vz[GRAD ? N : 1], vn[GRAD ? N : 1];
};

// exact signed distance of sphere obstacle (cx,cy,cz,r) at slot I; keeps the running minimum (and for GRAD
// the un-normalised direction v and its norm vn with grad sdf = v / vn)
template <bool GRAD, int I>
__device__ __forceinline__ void exact_sphere(LinkChunk<GRAD>& C, float cx, float cy, float cz, float r) {
    const float dx = C.x[I] - cx, dy = C.y[I] - cy, dz = C.z[I] - cz;
    float d2 = dx * dx + dy * dy + dz * dz;
    if (GRAD) d2 = fmaxf(d2, 1e-30f);  // keeps 1/dist finite; the oracle clamps the same way
    const float dist = fast_sqrt(d2);
    const float sd = dist - r;
    if (GRAD) {
        const bool better = sd < C.best[I];
        C.vx[I] = better ? dx : C.vx[I];
        C.vy[I] = better ? dy : C.vy[I];
        C.vz[I] = better ? dz : C.vz[I];
        C.vn[I] = better ? dist : C.vn[I];
    }
    C.best[I] = fminf(C.best[I], sd);
}

template <bool GRAD, int I>
__device__ __forceinline__ void exact_box(LinkChunk<GRAD>& C, const float4 c, const float4 h) {
    const float px = C.x[I] - c.x, py = C.y[I] - c.y, pz = C.z[I] - c.z;
    const float ax = fabsf(px) - h.x, ay = fabsf(py) - h.y, az = fabsf(pz) - h.z;
    const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
    float o2 = qx * qx + qy * qy + qz * qz;
    if (GRAD) o2 = fmaxf(o2, 1e-30f);
    const float outside = fast_sqrt(o2);
    const float mx = fmaxf(ax, fmaxf(ay, az));
    const float sd = outside + fminf(mx, 0.f);
    if (GRAD) {
        const bool better = sd < C.best[I];
        const bool out = mx > 0.f;
        // outside: direction of the clamped offset; inside: unit axis of the largest component
        // (first on ties, as torch.max)
        const bool ix = (ax >= ay) && (ax >= az);
        const bool iy = !ix && (ay >= az);
        const float nx = out ? copysignf(qx, px) : (ix ? copysignf(1.f, px) : 0.f);
        const float ny = out ? copysignf(qy, py) : (iy ? copysignf(1.f, py) : 0.f);
        const float nz = out ? copysignf(qz, pz) : ((!ix && !iy) ? copysignf(1.f, pz) : 0.f);
        C.vx[I] = better ? nx : C.vx[I];
        C.vy[I] = better ? ny : C.vy[I];
        C.vz[I] = better ? nz : C.vz[I];
        C.vn[I] = better ? (out ? outside : 1.f) : C.vn[I];
    }
    C.best[I] = fminf(C.best[I], sd);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// conservative test of all N slots against a block of 4 obstacles (k[2o] = (-2cx,-2cy,-2cz,rhs)):
// true when some lane of the wave may be within the hinge threshold of one of them.  Straight-line,
// 4 N independent fma chains; the 64-bit lane masks are OR-ed on the scalar unit.
template <bool GRAD>
__device__ __forceinline__ bool block_may_touch(const LinkChunk<GRAD>& C, const float4 (&k)[8]) {
    unsigned long long m = 0ull;
    static_for<0, LinkChunk<GRAD>::N>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float t = fmaf(C.z[I], k[2 * o].z, fmaf(C.y[I], k[2 * o].y, fmaf(C.x[I], k[2 * o].x, C.xx[I])));
            m |= __ballot(t < k[2 * o].w);
        }
    });
    return m != 0ull;
}

// exact distances of all N slots to the 4 obstacles of a block (k[2o+1] = (cx,cy,cz,r)); straight-line
template <bool GRAD>
__device__ __forceinline__ void block_exact(LinkChunk<GRAD>& C, const float4 (&k)[8]) {
    static_for<0, LinkChunk<GRAD>::N>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        exact_sphere<GRAD, I>(C, k[1].x, k[1].y, k[1].z, k[1].w);
        exact_sphere<GRAD, I>(C, k[3].x, k[3].y, k[3].z, k[3].w);
        exact_sphere<GRAD, I>(C, k[5].x, k[5].y, k[5].z, k[5].w);
        exact_sphere<GRAD, I>(C, k[7].x, k[7].y, k[7].z, k[7].w);
    });
}

// Wave-uniform statistics of the conservative test: when it passes most of the time (wildly spread
// trajectories, e.g. the first STOMP iterations with a large noise scale) it is pure overhead and is
// switched off for the rest of the waypoint; the result is bit-identical either way.
struct CullStats {
    int tested, hit;
    bool on;
};

// min signed distance of the N slots in C to every obstacle
template <bool GRAD>
__device__ __forceinline__ void chunk_vs_obstacles(const GeomView& G, LinkChunk<GRAD>& C, CullStats& cs) {
    const float4* cu = reinterpret_cast<const float4*>(G.cull);
    for (int ob = 0; ob < G.n_sph; ob += 4) {
        float4 k[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = cu[2 * ob + i];
        if (MPB_CULL_MODE != 1 && cs.on) {
            const bool any = block_may_touch<GRAD>(C, k);
            cs.tested += 1;
            cs.hit += any ? 1 : 0;
            if (MPB_CULL_MODE == 0) cs.on = (cs.tested < 8) || (2 * cs.hit <= cs.tested);
            if (!any) continue;
        }
        block_exact<GRAD>(C, k);
    }
    const float4* bp = reinterpret_cast<const float4*>(G.box);
    for (int o = 0; o < G.n_box; ++o) {
        const float4 c = bp[2 * o], h = bp[2 * o + 1];
        static_for<0, LinkChunk<GRAD>::N>([&](auto ic) { exact_box<GRAD, decltype(ic)::value>(C, c, h); });
    }
}

// The chain arithmetic is written with explicit fma sequences (no `a * b + c` left for the compiler to contract as it
// pleases): the generic table-driven walk and the compile-time robot models (below) then execute the SAME operation
// sequence -- a model merely drops the terms whose constant factor is an exact 0 and the products by an exact 1, which
// changes no bit -- so both return identical results.
//   mad3: t + a x + b y + c z (three fma);  dot3: a x + b y + c z (mul, two fma)
__device__ __forceinline__ float mad3(float a, float x, float b, float y, float c, float z, float t) {
    return fmaf(c, z, fmaf(b, y, fmaf(a, x, t)));
}
__device__ __forceinline__ float dot3(float a, float x, float b, float y, float c, float z) {
    return fmaf(c, z, fmaf(b, y, a * x));
}

// d cost / d q_j contribution of one collision sphere at x with hinge force f: f . (z_j x (x - p_j)), written out as
// explicit fma sequences so that the table-driven gradient walk and the compile-time models round alike
__device__ __forceinline__ float joint_term(float zx, float zy, float zz, float px, float py, float pz, float x, float y,
                                            float z, float fx, float fy, float fz) {
    const float ex = x - px, ey = y - py, ez = z - pz;
    const float cx = fmaf(zy, ez, -(zz * ey));
    const float cy = fmaf(zz, ex, -(zx * ez));
    const float cz = fmaf(zx, ey, -(zy * ex));
    return fmaf(fz, cz, fmaf(fy, cy, fx * cx));
}

// forward-kinematics state: current frame transform (+ joint axes / origins for the gradient)
template <bool GRAD>
struct FKState {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22, tx, ty, tz;
    int frame;  // number of transforms applied so far
    float zx[GRAD ? MPB_MAX_DOF : 1], zy[GRAD ? MPB_MAX_DOF : 1], zz[GRAD ? MPB_MAX_DOF : 1];
    float px[GRAD ? MPB_MAX_DOF : 1], py[GRAD ? MPB_MAX_DOF : 1], pz[GRAD ? MPB_MAX_DOF : 1];
};

// frame_{j+1} = frame_j * P_j * Rz(q_j); q_j and the joint-array slot are picked with selects, never by a
// runtime register index
template <bool GRAD>
__device__ __forceinline__ void fk_advance(const GeomView& G, FKState<GRAD>& F, const float (&q)[MPB_MAX_DOF]) {
    const int j = F.frame;
    const float4* P = reinterpret_cast<const float4*>(G.tf + 12 * j);
    const float4 p0 = P[0], p1 = P[1], p2 = P[2];  // rows of the 3x4 constant transform
    const float ntx = mad3(F.r00, p0.w, F.r01, p1.w, F.r02, p2.w, F.tx);
    const float nty = mad3(F.r10, p0.w, F.r11, p1.w, F.r12, p2.w, F.ty);
    const float ntz = mad3(F.r20, p0.w, F.r21, p1.w, F.r22, p2.w, F.tz);
    F.tx = ntx; F.ty = nty; F.tz = ntz;
    float a00 = dot3(F.r00, p0.x, F.r01, p1.x, F.r02, p2.x), a01 = dot3(F.r00, p0.y, F.r01, p1.y, F.r02, p2.y),
          a02 = dot3(F.r00, p0.z, F.r01, p1.z, F.r02, p2.z);
    float a10 = dot3(F.r10, p0.x, F.r11, p1.x, F.r12, p2.x), a11 = dot3(F.r10, p0.y, F.r11, p1.y, F.r12, p2.y),
          a12 = dot3(F.r10, p0.z, F.r11, p1.z, F.r12, p2.z);
    float a20 = dot3(F.r20, p0.x, F.r21, p1.x, F.r22, p2.x), a21 = dot3(F.r20, p0.y, F.r21, p1.y, F.r22, p2.y),
          a22 = dot3(F.r20, p0.z, F.r21, p1.z, F.r22, p2.z);
    if (j < G.n_dof) {
        // select chain on the wave-uniform j (7 v_cndmask); the empty asm keeps hipcc from turning it into a
        // scratch-memory array lookup
        float qj = q[0];
#pragma unroll
        for (int i = 1; i < MPB_MAX_DOF; ++i) {
            qj = (j == i) ? q[i] : qj;
            asm volatile("" : "+v"(qj));
        }
        float sn, cs;
        fast_sincos(qj, sn, cs);
        const float n00 = fmaf(a01, sn, a00 * cs), n01 = fmaf(-a00, sn, a01 * cs);
        const float n10 = fmaf(a11, sn, a10 * cs), n11 = fmaf(-a10, sn, a11 * cs);
        const float n20 = fmaf(a21, sn, a20 * cs), n21 = fmaf(-a20, sn, a21 * cs);
        a00 = n00; a01 = n01; a10 = n10; a11 = n11; a20 = n20; a21 = n21;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) {
                const bool me = (i == j);
                F.zx[i] = me ? a02 : F.zx[i]; F.zy[i] = me ? a12 : F.zy[i]; F.zz[i] = me ? a22 : F.zz[i];
                F.px[i] = me ? F.tx : F.px[i]; F.py[i] = me ? F.ty : F.py[i]; F.pz[i] = me ? F.tz : F.pz[i];
            }
        }
    }
    F.r00 = a00; F.r01 = a01; F.r02 = a02; F.r10 = a10; F.r11 = a11; F.r12 = a12;
    F.r20 = a20; F.r21 = a21; F.r22 = a22;
    F.frame = j + 1;
}

// Point robot: one collision sphere at q -- straight loops over the obstacle spheres (4 per trip, SGPR
// operands) and boxes; no chunk state, so the CHOMP / MPPI kernels of the 2-D examples carry no dead slots.
template <bool GRAD>
__device__ __forceinline__ float point_cost(const GeomView& G, float x, float y, float z, float& gx, float& gy,
                                            float& gz) {
    float best = 3.0e38f, vx = 0.f, vy = 0.f, vz = 0.f, vn = 1.f;
    const float4* sp = reinterpret_cast<const float4*>(G.sph);
#pragma unroll 4
    for (int o = 0; o < G.n_sph; ++o) {
        const float4 s = sp[o];
        const float dx = x - s.x, dy = y - s.y, dz = z - s.z;
        float d2 = dx * dx + dy * dy + dz * dz;
        if (GRAD) d2 = fmaxf(d2, 1e-30f);
        const float dist = fast_sqrt(d2);
        const float sd = dist - s.w;
        if (GRAD) {
            const bool better = sd < best;
            vx = better ? dx : vx; vy = better ? dy : vy; vz = better ? dz : vz; vn = better ? dist : vn;
        }
        best = fminf(best, sd);
    }
    const float4* bp = reinterpret_cast<const float4*>(G.box);
#pragma unroll 2
    for (int o = 0; o < G.n_box; ++o) {
        const float4 c = bp[2 * o], h = bp[2 * o + 1];
        const float px = x - c.x, py = y - c.y, pz = z - c.z;
        const float ax = fabsf(px) - h.x, ay = fabsf(py) - h.y, az = fabsf(pz) - h.z;
        const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
        float o2 = qx * qx + qy * qy + qz * qz;
        if (GRAD) o2 = fmaxf(o2, 1e-30f);
        const float outside = fast_sqrt(o2);
        const float mx = fmaxf(ax, fmaxf(ay, az));
        const float sd = outside + fminf(mx, 0.f);
        if (GRAD) {
            const bool better = sd < best;
            const bool out = mx > 0.f;
            const bool ix = (ax >= ay) && (ax >= az);
            const bool iy = !ix && (ay >= az);
            const float nx = out ? copysignf(qx, px) : (ix ? copysignf(1.f, px) : 0.f);
            const float ny = out ? copysignf(qy, py) : (iy ? copysignf(1.f, py) : 0.f);
            const float nz = out ? copysignf(qz, pz) : ((!ix && !iy) ? copysignf(1.f, pz) : 0.f);
            vx = better ? nx : vx; vy = better ? ny : vy; vz = better ? nz : vz;
            vn = better ? (out ? outside : 1.f) : vn;
        }
        best = fminf(best, sd);
    }
    const float h = fmaxf(G.margin + G.links[4] - best, 0.f);
    if (GRAD) {
        const float sc = (h > 0.f) ? -1.0f / vn : 0.f;
        gx = vx * sc; gy = vy * sc; gz = vz * sc;
    }
    return h;
}

// Collision cost of one waypoint q[0..D) (sum over the robot's collision spheres); for GRAD
// dq[i] = d cost / d q_i (i < D).  q / dq are register arrays indexed only with compile-time indices.
template <bool GRAD>
__device__ __forceinline__ float waypoint_cost(const GeomView& G, const float (&q)[MPB_MAX_DOF],
                                               float (&dq)[MPB_MAX_DOF]) {
    constexpr int N = LinkChunk<GRAD>::N;
    constexpr float FAR = 1.0e9f;  // parked slot: farther than any obstacle, hinge 0
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
    }
    if (G.kind == MPB_KIND_POINT) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        const float h = point_cost<GRAD>(G, q[0], q[1], (G.n_dof > 2) ? q[2] : 0.f, gx, gy, gz);
        if (GRAD) { dq[0] = gx; dq[1] = gy; if (G.n_dof > 2) dq[2] = gz; }
        return h;
    }
    LinkChunk<GRAD> C;
    CullStats cs = {0, 0, true};
    FKState<GRAD> F;
    F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
    F.tx = F.ty = F.tz = 0.f;
    F.frame = 0;
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) { F.zx[i] = F.zy[i] = F.zz[i] = F.px[i] = F.py[i] = F.pz[i] = 0.f; }
    }
    const bool point = (G.kind == MPB_KIND_POINT);
    float cost = 0.f;
    // chunks of N consecutive collision spheres; the kinematic chain advances inside the slot loop so
    // that every chunk is full whatever the spheres-per-frame distribution is
    for (int l0 = 0; l0 < G.n_links; l0 += N) {
        const int nl = min(N, G.n_links - l0);
        float rl[N];
        int fr[GRAD ? N : 1];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            C.best[i] = 3.0e38f;
            if (GRAD) { C.vx[i] = C.vy[i] = C.vz[i] = 0.f; C.vn[i] = 1.f; }
            if (i < nl) {
                const int li = l0 + i;
                const float4 lk = *reinterpret_cast<const float4*>(G.links + 8 * li);  // frame, ox, oy, oz
                rl[i] = G.links[8 * li + 4];
                if (point) {
                    C.x[i] = q[0]; C.y[i] = q[1]; C.z[i] = (G.n_dof > 2) ? q[2] : 0.f;
                    if (GRAD) fr[i] = 0;
                } else {
                    const int f = __float_as_int(lk.x);
                    while (F.frame < f) fk_advance<GRAD>(G, F, q);
                    C.x[i] = mad3(F.r00, lk.y, F.r01, lk.z, F.r02, lk.w, F.tx);
                    C.y[i] = mad3(F.r10, lk.y, F.r11, lk.z, F.r12, lk.w, F.ty);
                    C.z[i] = mad3(F.r20, lk.y, F.r21, lk.z, F.r22, lk.w, F.tz);
                    if (GRAD) fr[i] = f;
                }
            } else {
                rl[i] = 0.f;
                C.x[i] = C.y[i] = C.z[i] = FAR;
                if (GRAD) fr[i] = 0;
            }
            C.xx[i] = C.x[i] * C.x[i] + C.y[i] * C.y[i] + C.z[i] * C.z[i];
        }
        chunk_vs_obstacles<GRAD>(G, C, cs);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float h = fmaxf(G.margin + rl[i] - C.best[i], 0.f);  // parked slots: best = 3e38 -> 0
            cost += h;
            if (GRAD) {
                if (__any(h > 0.f)) {
                    const float s = (h > 0.f) ? -1.0f / C.vn[i] : 0.f;
                    const float fx = C.vx[i] * s, fy = C.vy[i] * s, fz = C.vz[i] * s;
                    if (point) {
                        dq[0] += fx; dq[1] += fy;
                        if (G.n_dof > 2) dq[2] += fz;
                    } else {
#pragma unroll
                        for (int ii = 0; ii < MPB_MAX_DOF; ++ii) {
                            if (ii < fr[i] && ii < G.n_dof) {
                                // d x / d q_ii = z_ii x (x - p_ii) for every joint upstream of the sphere's frame
                                dq[ii] += joint_term(F.zx[ii], F.zy[ii], F.zz[ii], F.px[ii], F.py[ii], F.pz[ii], C.x[i], C.y[i],
                                                     C.z[i], fx, fy, fz);
                            }
                        }
                    }
                }
            }
        }
    }
    return cost;
}


// ------------------------------------------------------------------------------------------------
// Broad-phase variant of the cost-only path.  The wave-level conservative test above cannot skip work
// when the 64 waypoints of a trajectory are spread over the workspace (wide STOMP noise): some lane is
// always near every obstacle.  A uniform grid culls PER LANE: each collision sphere looks up the cell
// that contains it (one LDS word: up to four obstacle indices, host-built in fp64, conservative) and
// evaluates exact distances only to those candidates; the wave iterates max-over-lanes(candidates)
// times (typically 1-3 instead of n_sph).  The minimum over the candidates equals the minimum over
// all obstacles whenever the hinge is active, so the cost is bit-identical to the exhaustive loop.
//   gridw : n_cells words in LDS;  otab : (cx,cy,cz,r) of the n_sph spheres + a far dummy at n_sph, in LDS.
// ------------------------------------------------------------------------------------------------
// sum_f s_f * cost_f(q) over the chained fields (exhaustive / gradient evaluator); dq accumulates the gradient
template <bool GRAD>
__device__ __forceinline__ float waypoint_cost_chain(const float* __restrict__ geom, const float (&q)[MPB_MAX_DOF],
                                                    float (&dq)[MPB_MAX_DOF]) {
    float c = 0.f;
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
    for (const float* gp = geom; gp != nullptr; gp = geom_next(gp)) {
        const GeomView G = geom_view(gp);
        float dqf[MPB_MAX_DOF];
        c = fmaf(G.fscale, waypoint_cost<GRAD>(G, q, dqf), c);
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = fmaf(G.fscale, dqf[i], dq[i]);
        }
    }
    return c;
}

// issue priority of a wave with r groups of the chain walk still to come after the current one (see model_group_positions)
#ifndef MPB_COST_PRIO
#define MPB_COST_PRIO(r) ((r) < 3 ? (r) : 3)
#endif
__device__ __forceinline__ bool grid_usable(const GeomView& G) {
    return G.version == MPB_GEOM_VERSION && G.n_cells > 0 && G.n_cells <= MPB_GRID_MAX_CELLS && G.n_sph <= MPB_GRID_MAX_SPH;
}

// clamped cell of a point, all in fp32: 3 fma + 3 floor + 3 med3 + 2 fma + 1 cvt (the integer formulation needs
// floor-convert + min + max per axis and two integer mads, one of them quarter rate).  Cell coordinates and the
// cell count stay below 2^24, so the float index arithmetic is exact; fma(x, inv, -lo * inv) differs from the
// host's (x - lo) * inv by ~1e-6 cells, inside the 1e-5 m the host adds to the candidate radius for exactly this
// (geometry.py build_grid).  NaN / inf coordinates land in some valid cell.
// FP32 = false keeps the integer formulation: the gradient evaluators are register-bound, and the eight uniform float
// constants of the fp32 form cost them more than the shorter index arithmetic saves (cost+grad +5 %, measured).
template <bool FP32>
__device__ __forceinline__ unsigned grid_cell(const GeomView& G, float x, float y, float z) {
    if (!FP32) {
        const float fx = (x - G.glx) * G.gix, fy = (y - G.gly) * G.giy, fz = (z - G.glz) * G.giz;
#ifdef MPB_GRID_CELL_FP32
        const int ix = min(max((int)floorf(fx), 0), G.gnx - 1), iy = min(max((int)floorf(fy), 0), G.gny - 1),
                  iz = min(max((int)floorf(fz), 0), G.gnz - 1);
        return (unsigned)(__mul24(__mul24(iz, G.gny) + iy, G.gnx) + ix);
#else
        // (one clamp of the linear index instead of two per axis: see below -- a point outside the box has hinge 0 and
        // therefore gradient 0 whatever candidates it is given)
        int ix, iy, iz;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(fx));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(fy));
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iz) : "v"(fz));
        unsigned t, idx;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(iz), "s"(G.gny), "v"(iy));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(idx) : "v"(t), "s"(G.gnx), "v"(ix));
        return min(idx, (unsigned)(G.n_cells - 1));
#endif
    }
#ifdef MPB_GRID_CELL_FP32
    const float fx = __builtin_amdgcn_fmed3f(floorf(fmaf(x, G.gix, -G.glx * G.gix)), 0.f, (float)(G.gnx - 1));
    const float fy = __builtin_amdgcn_fmed3f(floorf(fmaf(y, G.giy, -G.gly * G.giy)), 0.f, (float)(G.gny - 1));
    const float fz = __builtin_amdgcn_fmed3f(floorf(fmaf(z, G.giz, -G.glz * G.giz)), 0.f, (float)(G.gnz - 1));
    return (unsigned)fmaf(fmaf(fz, (float)G.gny, fy), (float)G.gnx, fx);
#else
    // nine instructions instead of twelve: fma + v_cvt_flr_i32_f32 per axis, two v_mad_u32_u24, ONE clamp of the linear
    // index.  No per-axis clamp: a point inside the box gets the cell it always got; a point outside it (or on its upper
    // faces) lands in SOME valid cell, and whatever that cell lists gives hinge 0 exactly (the box bounds the inflated
    // obstacles: see spheres_hinge_grid).  Saturated converts (parked slots at 1e9, inf) and NaN (-> 0) included.
    int ix, iy, iz;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(fmaf(x, G.gix, -G.glx * G.gix)));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(fmaf(y, G.giy, -G.gly * G.giy)));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iz) : "v"(fmaf(z, G.giz, -G.glz * G.giz)));
    unsigned t, idx;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(iz), "s"(G.gny), "v"(iy));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(idx) : "v"(t), "s"(G.gnx), "v"(ix));
    return min(idx, (unsigned)(G.n_cells - 1));
#endif
}

// BYTE OFFSET of a point's grid word (persistent STOMP kernels, MPPI; geometry version 6), clamped to the grid.  The cells
// sit on a lattice through the world origin (geometry.py build_grid): cell = round-to-nearest(x / h) - K per axis,
// and fma(x, 1/h, 1.5 * 2^23) leaves that integer in the low mantissa bits of a float in [2^23, 2^24) -- ONE instruction with
// two register sources and a literal (v_fmaak_f32: full rate) where floor((x - lo) / h) takes a three-source fma and a
// convert (half rate each).  The linear index is combined on the BIT PATTERNS, modulo 2^32: two v_mad_u32_u24 (the low 24 bits of
// a pattern are 2^22 + the integer; the constant parts are folded into c_rel), the byte offset is one v_lshl_add_u32, one
// v_min_u32 clamps it: 7 instructions where grid_cell<true> + the index shift are 10 at ~42 cycles
// (profiles/r05_isa_cost_hist_before.md: 12 % of the vector pipe at C3; the first form of this round combined in float -- two
// subtractions of the constant, two v_fmac: 9 instructions -- and measured 3.5 % slower at C3 than the integer form).  A point
// outside the box, a parked slot at 1e9, inf or NaN give SOME offset inside the grid, as before (whatever that cell lists yields
// hinge 0: the box bounds the inflated obstacles); a point on a cell face goes to either neighbour (ties to even), which the
// host's 1e-5 m of slack on the candidate radius covers like the fp32 rounding of the old form.
struct GridAddr {
    float gix_v, giy_v, giz_v;   // 1 / h per axis in VECTOR registers (v_fmaak takes the literal plus two registers; a scalar would make it a VOP3 fma)
    unsigned gnx_u, gnxy_u;      // row / slab lengths in cells (scalar operands of the two v_mad_u32_u24)
    unsigned c_rel;      // -4 * (bits(1.5 * 2^23) + 2^22 (gnx + gnx gny) + k_lin): what the combined bit patterns carry besides the linear index
    unsigned max_rel;    // 4 * (n_cells - 1)
};
__device__ __forceinline__ GridAddr grid_addr(const GeomView& G) {
    GridAddr A;
    A.gix_v = G.gix; A.giy_v = G.giy; A.giz_v = G.giz;
    asm volatile("" : "+v"(A.gix_v), "+v"(A.giy_v), "+v"(A.giz_v));
    A.gnx_u = (unsigned)G.gnx;
    A.gnxy_u = (unsigned)G.gnx * (unsigned)G.gny;
    A.c_rel = 0u - 4u * (0x4B400000u + 0x400000u * (A.gnx_u + A.gnxy_u) + (unsigned)G.k_lin);
    A.max_rel = 4u * (unsigned)(G.n_cells - 1);
    return A;
}
__device__ __forceinline__ unsigned grid_cell_rel(const GridAddr& A, float x, float y, float z) {
    constexpr float MAGIC = 12582912.0f;                 // 1.5 * 2^23
    const float tx = fmaf(x, A.gix_v, MAGIC), ty = fmaf(y, A.giy_v, MAGIC), tz = fmaf(z, A.giz_v, MAGIC);
    // bits(t?) = 0x4B400000 + r?: the low 24 bits are 2^22 + r? (|r?| < 2^21: mpb_geom_check), which is what v_mad_u32_u24 multiplies
    unsigned t, v;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(ty), "s"(A.gnx_u), "v"(tx));     // bits(tx) + (2^22 + ry) gnx
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(v) : "v"(tz), "s"(A.gnxy_u), "v"(t));     // ... + (2^22 + rz) gnx gny
    const unsigned rel = (v << 2) + A.c_rel;
    return min(rel, A.max_rel);
}

// cooperative staging by `nthreads` threads (caller synchronises before and after)
__device__ __forceinline__ void grid_stage(const GeomView& G, unsigned* gridw, float4* otab, int tid, int nthreads) {
    for (int i = tid; i < G.n_cells; i += nthreads) gridw[i] = G.grid[i];
    const float4* sp = reinterpret_cast<const float4*>(G.sph);
    for (int i = tid; i <= G.n_sph; i += nthreads)
        otab[i] = (i < G.n_sph) ? sp[i] : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
}

// OFFSET WORDS (persistent STOMP kernels; round 4): the LDS copy of the grid re-encoded while it is staged, so that the
// look-up of the walk needs no index arithmetic.  The obstacle table has at most 64 entries of 16 B (63 spheres + the far
// dummy): a byte offset into it fits ten bits.  Word: bits 0-9 slot 0 as a byte offset (an empty cell: the dummy's),
// bits 10-19 / 20-29 slots 1 / 2 as byte offsets, 0 = unused (an unused lane of a later trip then reads obstacle 0: any
// obstacle that is not a candidate of the cell is beyond margin + r of every point in it and cannot change a hinge -- the
// property the overflow path rests on); bit 31: the cell lists more than three obstacles -> exhaustive loop.  Obstacle
// 0 can only sit in slot 0 (the host lists candidates in ascending order; any order is handled).  Against the byte
// form per group of four spheres: slot 0's address is one v_and (was and + shift), "does any lane need another trip"
// one compare of the four words or-ed together (was a field extract and a compare per sphere), and the overflow test
// rides on it (was a compare per sphere): 104 -> 91 vector instructions per group on the common path.
__device__ __forceinline__ unsigned grid_offset_word(unsigned w, unsigned n_sph) {
    if (w == MPB_GRID_OVERFLOW) return 0x80000000u | (n_sph << 4);
    unsigned b[4] = {0u, 0u, 0u, 0u}, n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned v = (w >> (8 * k)) & 0xFFu;
        if (v < n_sph) {                                            // used slots, in the order found
#pragma unroll
            for (int t = 0; t < 4; ++t) if ((unsigned)t == n) b[t] = v;
            ++n;
        }
    }
    if (n > 3) return 0x80000000u | (n_sph << 4);
    // obstacle 0 to the front (its offset, 0, means "unused" in the later slots)
    if (n > 1 && b[1] == 0u) { b[1] = b[0]; b[0] = 0u; }
    if (n > 2 && b[2] == 0u) { b[2] = b[0]; b[0] = 0u; }
    unsigned r = ((n > 0 ? b[0] : n_sph) << 4);
    if (n > 1) r |= b[1] << 14;
    if (n > 2) r |= b[2] << 24;
    return r;
}
__device__ __forceinline__ void grid_stage_offsets(const GeomView& G, unsigned* gridw, float4* otab, int tid, int nthreads) {
    for (int i = tid; i < G.n_cells; i += nthreads) gridw[i] = grid_offset_word(G.grid[i], (unsigned)G.n_sph);
    const float4* sp = reinterpret_cast<const float4*>(G.sph);
    for (int i = tid; i <= G.n_sph; i += nthreads)
        otab[i] = (i < G.n_sph) ? sp[i] : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
}

// N collision spheres at once: the N grid words are fetched together and every trip of the candidate
// loop issues N obstacle-table reads before the N distance evaluations, so the LDS latency is paid
// once per group instead of once per sphere.  Adds the N hinges to `cost` one by one, in sphere order
// (the same association as the exhaustive path, so both paths agree bit for bit).
// UNIT (compile-time robot models only): the caller guarantees every hinge margin + r_l - sdf is below 1 (pack_geometry tags a
// buffer with a model only when margin + max r_l + max obstacle radius < 1 m, mpb_geom_check verifies it), so relu is the
// [0, 1] clamp the VOP3 encoding applies for free on the subtraction -- same bits as v_max_f32(x, 0), which issues at half rate.
// RLM (compile-time robot models only): rl[] already holds margin + r_l (the model's arms add the margin to their literal radii,
// one v_add with a literal in place of the v_mov that materialised the radius): the same (margin + r_l) - sdf, one instruction
// per sphere less in the shared tail.
template <int N, bool OFFS = false, bool UNIT = false, bool RLM = false>
__device__ __forceinline__ void spheres_hinge_grid(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                   const float (&x)[N], const float (&y)[N], const float (&z)[N],
                                                   const float (&rl)[N], float& cost, const GridAddr& GA = GridAddr{}) {
    unsigned w[N];
    float best[N];
    if constexpr (OFFS) {
        // the grid as offset words (grid_offset_word): slot 0 of every sphere unconditionally, the rest behind ONE test
        const char* ob = reinterpret_cast<const char*>(otab);
        float4 s0[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            w[i] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(gridw) + grid_cell_rel(GA, x[i], y[i], z[i]));
#pragma unroll
        for (int i = 0; i < N; ++i) s0[i] = *reinterpret_cast<const float4*>(ob + (w[i] & 0x3FFu));
        unsigned comb = w[0];
#pragma unroll
        for (int i = 1; i < N; ++i) comb |= w[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float dx = x[i] - s0[i].x, dy = y[i] - s0[i].y, dz = z[i] - s0[i].z;
            best[i] = fast_sqrt(dx * dx + dy * dy + dz * dz) - s0[i].w;
        }
        if (__ballot(comb > 0x3FFu) != 0ull) {
            if (__builtin_expect(__ballot((int)comb < 0) != 0ull, 0)) {
                // some lane sits in a crowded cell: exhaustive exact loop for this group (rare)
                for (int o = 0; o < G.n_sph; ++o) {
                    const float4 s = otab[o];
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float dx = x[i] - s.x, dy = y[i] - s.y, dz = z[i] - s.z;
                        best[i] = fminf(best[i], fast_sqrt(dx * dx + dy * dy + dz * dz) - s.w);
                    }
                }
            } else {
                // (measured and dropped, twice: the later slots sphere by sphere, each behind its own wave-uniform test -- +1.5 %)
#pragma unroll
                for (int k = 1; k < 3; ++k) {
                    if (k > 1 && __ballot(comb > 0xFFFFFu) == 0ull) break;
                    float4 s[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) s[i] = *reinterpret_cast<const float4*>(ob + ((w[i] >> (10 * k)) & 0x3FFu));
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float dx = x[i] - s[i].x, dy = y[i] - s[i].y, dz = z[i] - s[i].z;
                        best[i] = fminf(best[i], fast_sqrt(dx * dx + dy * dy + dz * dz) - s[i].w);
                    }
                }
            }
        }
    } else {
    unsigned long long over = 0ull;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // a point outside the grid box is farther than margin + r_l from every obstacle (the box is the bounding box
        // of the inflated obstacles), so whatever candidates its CLAMPED cell lists all give hinge 0 exactly: no
        // in-bounds test (parked slots at 1e9 clamp to the last cell)
        w[i] = gridw[grid_cell<true>(G, x[i], y[i], z[i])];
        best[i] = 3.0e38f;
        over |= __ballot(w[i] == MPB_GRID_OVERFLOW);
    }
    if (__builtin_expect(over != 0ull, 0)) {
        // some lane sits in a crowded cell: exhaustive exact loop for this group (rare)
        for (int o = 0; o < G.n_sph; ++o) {
            const float4 s = otab[o];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float dx = x[i] - s.x, dy = y[i] - s.y, dz = z[i] - s.z;
                best[i] = fminf(best[i], fast_sqrt(dx * dx + dy * dy + dz * dz) - s.w);
            }
        }
    } else {
        // candidate slot k of every sphere of the group at once (an unused slot holds n_sph, the far dummy of the
        // table: no index clamp); slot 0 is evaluated unconditionally (some lane of the wave always has a candidate),
        // the later ones only while some lane still has one.  (Measured and rejected: the rare later slots sphere by
        // sphere behind wave-uniform tests, +3 %; "slot in use" as one compare of w against a scalar threshold, +2 %.)
        const unsigned none = (unsigned)G.n_sph;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k > 0) {
                unsigned long long any = 0ull;
#pragma unroll
                for (int i = 0; i < N; ++i) any |= __ballot(((w[i] >> (8 * k)) & 0xFFu) != none);
                if (any == 0ull) break;
            }
            float4 s[N];
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = otab[(w[i] >> (8 * k)) & 0xFFu];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float dx = x[i] - s[i].x, dy = y[i] - s[i].y, dz = z[i] - s[i].z;
                const float dist = fast_sqrt(dx * dx + dy * dy + dz * dz) - s[i].w;
                // (slot 0 is always evaluated: min(3e38, dist) is dist -- v_min_f32 issues at half the rate of v_sub_f32 on
                // gfx950, profiles/r03_microbench_rates.txt)
                best[i] = (k == 0) ? dist : fminf(best[i], dist);
            }
        }
    }
    }   // byte form
    // boxes are few: exhaustive
    const float4* bp = reinterpret_cast<const float4*>(G.box);
    for (int o = 0; o < G.n_box; ++o) {
        const float4 c = bp[2 * o], h = bp[2 * o + 1];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float px = x[i] - c.x, py = y[i] - c.y, pz = z[i] - c.z;
            const float ax = fabsf(px) - h.x, ay = fabsf(py) - h.y, az = fabsf(pz) - h.z;
            const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
            const float sd = fast_sqrt(qx * qx + qy * qy + qz * qz) + fminf(fmaxf(ax, fmaxf(ay, az)), 0.f);
            best[i] = fminf(best[i], sd);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                 // parked slots: best = 3e38 / 1e9 -> +0
        const float hinge = (RLM ? rl[i] : G.margin + rl[i]) - best[i];
        cost += UNIT ? fminf(fmaxf(hinge, 0.f), 1.f) : fmaxf(hinge, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// LIST grid (geometry version 7, round 6): scenes beyond the compact grid's 63 spheres / its three candidates per cell, boxes
// culled like spheres.  A cell word names a RANGE of a byte array of candidate indices -- bits 0-14 start, 15-21 sphere count, 22-27
// box count (the boxes follow the spheres), bit 31: the cell overflows a count field -- and the evaluator walks the longest range
// among the wave's lanes, a lane whose range has ended reading the far dummy at the end of the table (no index clamp, no
// branch).  Everything it reads sits in LDS (list_stage): cell words, candidate bytes, the sphere table (up to 255 + dummy) and the
// box table (up to 127 + dummy).  The candidate sets are conservative (host, fp64), the distance expressions are the exhaustive
// evaluator's and min is exact: the hinges are the bits of the exhaustive path's.
// ------------------------------------------------------------------------------------------------
struct ListView {
    const unsigned* cellw;          // n_cells words
    const unsigned char* cand;      // candidate bytes
    const float4* stab;             // n_sph + 1 spheres (the last one the far dummy)
    const float4* btab;             // 2 (n_box + 1) float4: centre, half extents (the last box the far dummy)
};
__device__ __forceinline__ bool list_usable(const GeomView& G) {
    return G.version == MPB_GEOM_VERSION_LIST && G.n_cells > 0 && G.n_cells <= MPB_GRID_MAX_CELLS && G.n_sph <= MPB_LIST_MAX_SPH &&
           G.n_box <= MPB_LIST_MAX_BOX && G.n_cand <= MPB_LIST_MAX_CAND + 16;
}
// cooperative staging by `nthreads` threads (caller synchronises before and after); cand_l: MPB_LIST_MAX_CAND + 16 bytes
__device__ __forceinline__ void list_stage(const GeomView& G, unsigned* cellw_l, unsigned char* cand_l, float4* stab_l, float4* btab_l,
                                           int tid, int nthreads) {
    for (int i = tid; i < G.n_cells; i += nthreads) cellw_l[i] = G.grid[i];
    const uint4* c4 = reinterpret_cast<const uint4*>(G.cand);
    for (int i = tid; i < (G.n_cand >> 4); i += nthreads) reinterpret_cast<uint4*>(cand_l)[i] = c4[i];
    const float4* sp = reinterpret_cast<const float4*>(G.sph);
    for (int i = tid; i <= G.n_sph; i += nthreads) stab_l[i] = (i < G.n_sph) ? sp[i] : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f);
    const float4* bp = reinterpret_cast<const float4*>(G.box);
    for (int i = tid; i < 2 * (G.n_box + 1); i += nthreads)
        btab_l[i] = (i < 2 * G.n_box) ? bp[i] : ((i & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(-1.0e9f, -1.0e9f, -1.0e9f, 0.f));
}

template <int N, bool UNIT = false, bool RLM = false>
__device__ __forceinline__ void spheres_hinge_list(const GeomView& G, const ListView& L, const float (&x)[N], const float (&y)[N],
                                                   const float (&z)[N], const float (&rl)[N], float& cost, const GridAddr& GA) {
    unsigned w[N];
    float best[N];
    unsigned comb = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        w[i] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(L.cellw) + grid_cell_rel(GA, x[i], y[i], z[i]));
        best[i] = 3.0e38f;
        comb |= w[i];
    }
    auto sphere = [&](int i, const float4 s) {
        const float dx = x[i] - s.x, dy = y[i] - s.y, dz = z[i] - s.z;
        best[i] = fminf(best[i], fast_sqrt(dx * dx + dy * dy + dz * dz) - s.w);
    };
    auto box = [&](int i, const float4 c, const float4 h) {
        const float px = x[i] - c.x, py = y[i] - c.y, pz = z[i] - c.z;
        const float ax = fabsf(px) - h.x, ay = fabsf(py) - h.y, az = fabsf(pz) - h.z;
        const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
        const float sd = fast_sqrt(qx * qx + qy * qy + qz * qz) + fminf(fmaxf(ax, fmaxf(ay, az)), 0.f);
        best[i] = fminf(best[i], sd);
    };
    if (__builtin_expect(__ballot((int)comb < 0) != 0ull, 0)) {
        // some lane sits in a cell that overflows a count field: every obstacle for this group (rare)
        for (int o = 0; o < G.n_sph; ++o) {
            const float4 s = L.stab[o];
#pragma unroll
            for (int i = 0; i < N; ++i) sphere(i, s);
        }
        for (int o = 0; o < G.n_box; ++o) {
            const float4 c = L.btab[2 * o], h = L.btab[2 * o + 1];
#pragma unroll
            for (int i = 0; i < N; ++i) box(i, c, h);
        }
    } else {
        unsigned st[N], ns[N], nb[N];
        unsigned mns = 0u, mnb = 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            st[i] = w[i] & 0x7FFFu; ns[i] = (w[i] >> 15) & 0x7Fu; nb[i] = (w[i] >> 22) & 0x3Fu;
            mns = max(mns, ns[i]); mnb = max(mnb, nb[i]);
        }
        // as many trips as the longest sphere / box range among the wave's ACTIVE lanes (a ballot per trip: the callers run this
        // under a lane mask -- waypoint 0, lanes past the horizon --, which rules the DPP reductions out)
        const unsigned none_s = (unsigned)G.n_sph, none_b = (unsigned)G.n_box;
        for (unsigned k = 0; __ballot(k < mns) != 0ull; ++k) {
            float4 s[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const unsigned idx = (k < ns[i]) ? (unsigned)L.cand[st[i] + k] : none_s;
                s[i] = L.stab[idx];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) sphere(i, s[i]);
        }
        for (unsigned k = 0; __ballot(k < mnb) != 0ull; ++k) {
            float4 c[N], h[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const unsigned idx = (k < nb[i]) ? (unsigned)L.cand[st[i] + ns[i] + k] : none_b;
                c[i] = L.btab[2 * idx];
                h[i] = L.btab[2 * idx + 1];
            }
#pragma unroll
            for (int i = 0; i < N; ++i) box(i, c[i], h[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                 // parked slots: best = 1e9 -> +0
        const float hinge = (RLM ? rl[i] : G.margin + rl[i]) - best[i];
        cost += UNIT ? fminf(fmaxf(hinge, 0.f), 1.f) : fmaxf(hinge, 0.f);
    }
}

// LIST: the field carries a list grid (version 7) staged in LDS as *LV; gridw / otab are not read
template <bool OFFS = false, bool LIST = false>
__device__ __forceinline__ float waypoint_cost_grid(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                    const float (&q)[MPB_MAX_DOF], const ListView* LV = nullptr) {
    if (G.kind == MPB_KIND_POINT) {
        const float x[1] = {q[0]}, y[1] = {q[1]}, z[1] = {(G.n_dof > 2) ? q[2] : 0.f}, rl[1] = {G.links[4]};
        float c = 0.f;
        if constexpr (LIST) spheres_hinge_list<1>(G, *LV, x, y, z, rl, c, grid_addr(G));
        else spheres_hinge_grid<1, OFFS>(G, gridw, otab, x, y, z, rl, c, OFFS ? grid_addr(G) : GridAddr{});
        return c;
    }
#ifndef MPB_GRID_N
#define MPB_GRID_N 4
#endif
    constexpr int N = MPB_GRID_N;
    constexpr float FAR = 1.0e9f;  // parked slot: outside the grid, no candidates
    FKState<false> F;
    F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
    F.tx = F.ty = F.tz = 0.f;
    F.frame = 0;
    float cost = 0.f;
    const GridAddr GA = (OFFS || LIST) ? grid_addr(G) : GridAddr{};
    for (int l0 = 0; l0 < G.n_links; l0 += N) {
        const int nl = min(N, G.n_links - l0);
#ifndef MPB_NO_COST_PRIO
        // issue priority by the groups still to come (wave-uniform; see model_group_positions)
        switch ((G.n_links - l0 - 1) / N) {
            case 0: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(0)); break;
            case 1: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(1)); break;
            case 2: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(2)); break;
            default: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(3)); break;
        }
#endif
        float x[N], y[N], z[N], rl[N];
        // the N link records of the group are fetched together, so the scalar-load latency is paid once per group
        // instead of once per sphere.  Records past n_links (last group) are read but never used: the words behind
        // the link table are the sphere / cull tables of the same buffer, which exist whenever the grid path runs
        float4 lkv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lkv[i] = *reinterpret_cast<const float4*>(G.links + 8 * (l0 + i));
            rl[i] = G.links[8 * (l0 + i) + 4];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < nl) {
                const float4 lk = lkv[i];                                              // frame, ox, oy, oz
                const int f = __float_as_int(lk.x);
                while (F.frame < f) fk_advance<false>(G, F, q);
                x[i] = mad3(F.r00, lk.y, F.r01, lk.z, F.r02, lk.w, F.tx);
                y[i] = mad3(F.r10, lk.y, F.r11, lk.z, F.r12, lk.w, F.ty);
                z[i] = mad3(F.r20, lk.y, F.r21, lk.z, F.r22, lk.w, F.tz);
            } else {
                rl[i] = 0.f;
                x[i] = y[i] = z[i] = FAR;
            }
        }
        if constexpr (LIST) spheres_hinge_list<N>(G, *LV, x, y, z, rl, cost, GA);
        else spheres_hinge_grid<N, OFFS>(G, gridw, otab, x, y, z, rl, cost, GA);
    }
#ifndef MPB_NO_COST_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return cost;
}

// ------------------------------------------------------------------------------------------------
// Compile-time robot model (mpb_model_*.h, generated from geometry.py): the same chain walk and the same grid lookups
// as waypoint_cost_grid, with the joint transforms and the collision-sphere table as constexpr data -- the chain is
// unrolled, the exact 0 / +-1 entries of the modified-DH transforms and the zero components of the sphere offsets fold
// away, and none of it is fetched through scalar loads.  Expression by expression the arithmetic is that of fk_advance /
// waypoint_cost_grid (a product with an exact 0 or 1 is exact), so both evaluators return the same bits; the hinges are
// added in sphere order like there.  Only taken when the geometry buffer carries the model's id, which pack_geometry
// sets -- and mpb_geom_check verifies against these very constants -- when the robot's tables equal the model's.
//   keep_mask: spheres riding on frame 1 that static pruning dropped are parked (their hinge is exactly 0); a group of
//   frame-1 spheres with no survivor is skipped.
// ------------------------------------------------------------------------------------------------
struct ModelFK {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22, tx, ty, tz;
};

template <class M, int J>
__device__ __forceinline__ void model_fk_advance(ModelFK& F, const float (&q)[MPB_MAX_DOF]) {
    constexpr float p0x = M::TF[J][0], p0y = M::TF[J][1], p0z = M::TF[J][2], p0w = M::TF[J][3];
    constexpr float p1x = M::TF[J][4], p1y = M::TF[J][5], p1z = M::TF[J][6], p1w = M::TF[J][7];
    constexpr float p2x = M::TF[J][8], p2y = M::TF[J][9], p2z = M::TF[J][10], p2w = M::TF[J][11];
    const float ntx = mad3(F.r00, p0w, F.r01, p1w, F.r02, p2w, F.tx);
    const float nty = mad3(F.r10, p0w, F.r11, p1w, F.r12, p2w, F.ty);
    const float ntz = mad3(F.r20, p0w, F.r21, p1w, F.r22, p2w, F.tz);
    F.tx = ntx; F.ty = nty; F.tz = ntz;
    float a00 = dot3(F.r00, p0x, F.r01, p1x, F.r02, p2x), a01 = dot3(F.r00, p0y, F.r01, p1y, F.r02, p2y),
          a02 = dot3(F.r00, p0z, F.r01, p1z, F.r02, p2z);
    float a10 = dot3(F.r10, p0x, F.r11, p1x, F.r12, p2x), a11 = dot3(F.r10, p0y, F.r11, p1y, F.r12, p2y),
          a12 = dot3(F.r10, p0z, F.r11, p1z, F.r12, p2z);
    float a20 = dot3(F.r20, p0x, F.r21, p1x, F.r22, p2x), a21 = dot3(F.r20, p0y, F.r21, p1y, F.r22, p2y),
          a22 = dot3(F.r20, p0z, F.r21, p1z, F.r22, p2z);
    if constexpr (J < M::N_DOF) {
        float sn, cs;
        fast_sincos(q[J], sn, cs);
        const float n00 = fmaf(a01, sn, a00 * cs), n01 = fmaf(-a00, sn, a01 * cs);
        const float n10 = fmaf(a11, sn, a10 * cs), n11 = fmaf(-a10, sn, a11 * cs);
        const float n20 = fmaf(a21, sn, a20 * cs), n21 = fmaf(-a20, sn, a21 * cs);
        a00 = n00; a01 = n01; a10 = n10; a11 = n11; a20 = n20; a21 = n21;
    }
    F.r00 = a00; F.r01 = a01; F.r02 = a02; F.r10 = a10; F.r11 = a11; F.r12 = a12;
    F.r20 = a20; F.r21 = a21; F.r22 = a22;
}

// positions of the (up to four) collision spheres of group GRP, advancing the chain as far as they need.  Groups:
// the spheres on frame 1 first (the only ones static pruning can drop), then the rest, four at a time.
template <class M, int GRP>
__device__ __forceinline__ bool model_group_positions(ModelFK& F, const float (&q)[MPB_MAX_DOF], unsigned keep,
                                                      float (&x)[4], float (&y)[4], float (&z)[4], float (&rl)[4], float mrg) {
    constexpr float FAR = 1.0e9f;   // parked slot: outside the grid, no candidates
    constexpr int G1 = (M::N_FRAME1 + 3) / 4;
    constexpr bool first = GRP < G1;
    constexpr int lo = first ? 4 * GRP : M::N_FRAME1 + 4 * (GRP - G1);
    constexpr int part_end = first ? M::N_FRAME1 : M::N_LINKS;
    constexpr int hi = (lo + 4 < part_end) ? lo + 4 : part_end;
#ifndef MPB_NO_COST_PRIO
    // issue priority by progress: the SIMD arbiter serves its oldest wave first, so without this the waves of a SIMD
    // finish one after the other and the last one runs alone, latency-bound, at a fraction of the issue rate.  A wave
    // that is behind (earlier group) outranks the ones ahead of it, which keeps all of them in flight to the end.
    {
        constexpr int NG_ = (M::N_FRAME1 + 3) / 4 + (M::N_LINKS - M::N_FRAME1 + 3) / 4;
        __builtin_amdgcn_s_setprio(MPB_COST_PRIO(NG_ - 1 - GRP));
    }
#endif
    static_for<lo, hi>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int f = M::LINK_FRAME[l];
        constexpr int fprev = (l == 0) ? 0 : M::LINK_FRAME[l > 0 ? l - 1 : 0];
        static_for<fprev, f>([&](auto jc) { model_fk_advance<M, decltype(jc)::value>(F, q); });
        constexpr int slot = l - lo;
        constexpr float ox = M::LINK[l][0], oy = M::LINK[l][1], oz = M::LINK[l][2], rad = M::LINK[l][3];
        const float px = mad3(F.r00, ox, F.r01, oy, F.r02, oz, F.tx);
        const float py = mad3(F.r10, ox, F.r11, oy, F.r12, oz, F.ty);
        const float pz = mad3(F.r20, ox, F.r21, oy, F.r22, oz, F.tz);
        if constexpr (first) {
            const bool on = (keep >> l) & 1u;                // wave-uniform
            x[slot] = on ? px : FAR; y[slot] = on ? py : FAR; z[slot] = on ? pz : FAR; rl[slot] = on ? mrg + rad : 0.f;
        } else {
            x[slot] = px; y[slot] = py; z[slot] = pz; rl[slot] = mrg + rad;     // (margin + r_l: spheres_hinge_grid, RLM)
        }
    });
#pragma unroll
    for (int i = hi - lo; i < 4; ++i) { x[i] = y[i] = z[i] = FAR; rl[i] = 0.f; }   // (parked: 0 - 1e9 clamps to 0 like margin - 1e9)
    if constexpr (first) {
        constexpr unsigned gmask = ((hi >= 32) ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        return (keep & gmask) != 0u;                         // a frame-1 group with no survivor is skipped
    }
    return true;
}

template <class M, int... GRPS>
__device__ __forceinline__ bool model_group_dispatch(int grp, ModelFK& F, const float (&q)[MPB_MAX_DOF], unsigned keep,
                                                     float (&x)[4], float (&y)[4], float (&z)[4], float (&rl)[4], float mrg,
                                                     std::integer_sequence<int, GRPS...>) {
    bool run = false;
    // one arm per group, selected by the wave-uniform group counter (scalar compares / branches)
    // (a switch over the group index -- a jump table or a compare tree instead of the compare chain -- measured +1 %, round 5)
    ((grp == GRPS ? (void)(run = model_group_positions<M, GRPS>(F, q, keep, x, y, z, rl, mrg)) : (void)0), ...);
    return run;
}

template <class M, bool OFFS = false, bool LIST = false>
__device__ __forceinline__ float waypoint_cost_grid_model(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                          const float (&q)[MPB_MAX_DOF], const ListView* LV = nullptr) {
    constexpr int NG = (M::N_FRAME1 + 3) / 4 + (M::N_LINKS - M::N_FRAME1 + 3) / 4;
    ModelFK F;
    F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
    F.tx = F.ty = F.tz = 0.f;
    float cost = 0.f;
    const unsigned keep = G.keep_mask;
    const GridAddr GA = (OFFS || LIST) ? grid_addr(G) : GridAddr{};
    float mrg = G.margin;                    // in a VECTOR register: v_add takes the literal radius plus one register
    asm volatile("" : "+v"(mrg));
    // a real loop over the groups with ONE instance of the grid look-up (unrolling it per group is 60 KB of code)
#pragma nounroll
    for (int grp = 0; grp < NG; ++grp) {
        float x[4], y[4], z[4], rl[4];
        const bool run = model_group_dispatch<M>(grp, F, q, keep, x, y, z, rl, mrg, std::make_integer_sequence<int, NG>{});
        if constexpr (LIST) {
            if (run) spheres_hinge_list<4, true, true>(G, *LV, x, y, z, rl, cost, GA);
        } else {
            if (run) spheres_hinge_grid<4, OFFS, true, true>(G, gridw, otab, x, y, z, rl, cost, GA);
        }
    }
#ifndef MPB_NO_COST_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return cost;
}

// ------------------------------------------------------------------------------------------------
// Broad-phase variant of the GRADIENT path: the same per-lane candidate lists, additionally tracking the
// direction to the nearest obstacle (v, |v|) of every collision sphere; the hinge force -v/|v| of the spheres in
// contact is pulled back through the kinematic chain (J^T f over the joints upstream of the sphere's frame).
// The nearest obstacle among the candidates is the nearest overall whenever the hinge is active (first minimum in
// obstacle-index order, like the exhaustive loop), so cost and gradient equal the exhaustive evaluator's.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void spheres_nearest_grid(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                     const float (&x)[N], const float (&y)[N], const float (&z)[N],
                                                     float (&best)[N], float (&vx)[N], float (&vy)[N], float (&vz)[N],
                                                     float (&vn)[N]) {
    unsigned w[N];
    unsigned long long over = 0ull;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        w[i] = gridw[grid_cell<false>(G, x[i], y[i], z[i])];   // clamped cell index (exact: see spheres_hinge_grid)
        best[i] = 3.0e38f;
        vx[i] = vy[i] = vz[i] = 0.f;
        vn[i] = 1.f;
        over |= __ballot(w[i] == MPB_GRID_OVERFLOW);
    }
    auto visit = [&](int i, const float4 s) {
        const float dx = x[i] - s.x, dy = y[i] - s.y, dz = z[i] - s.z;
        const float d2 = fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f);   // keeps 1/dist finite; the oracle clamps the same way
        const float dist = fast_sqrt(d2);
        const float sd = dist - s.w;
        const bool better = sd < best[i];
        vx[i] = better ? dx : vx[i];
        vy[i] = better ? dy : vy[i];
        vz[i] = better ? dz : vz[i];
        vn[i] = better ? dist : vn[i];
        best[i] = fminf(best[i], sd);
    };
    if (__builtin_expect(over != 0ull, 0)) {
        for (int o = 0; o < G.n_sph; ++o) {
            const float4 s = otab[o];
#pragma unroll
            for (int i = 0; i < N; ++i) visit(i, s);
        }
    } else {
        const unsigned none = (unsigned)G.n_sph;      // unused candidate slot: the far dummy of the table
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k > 0) {
                unsigned long long any = 0ull;
#pragma unroll
                for (int i = 0; i < N; ++i) any |= __ballot(((w[i] >> (8 * k)) & 0xFFu) != none);
                if (any == 0ull) break;
            }
            float4 s[N];
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = otab[(w[i] >> (8 * k)) & 0xFFu];
#pragma unroll
            for (int i = 0; i < N; ++i) visit(i, s[i]);
        }
    }
    // boxes are few: exhaustive (same direction rules as exact_box)
    const float4* bp = reinterpret_cast<const float4*>(G.box);
    for (int o = 0; o < G.n_box; ++o) {
        const float4 c = bp[2 * o], h = bp[2 * o + 1];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float px = x[i] - c.x, py = y[i] - c.y, pz = z[i] - c.z;
            const float ax = fabsf(px) - h.x, ay = fabsf(py) - h.y, az = fabsf(pz) - h.z;
            const float qx = fmaxf(ax, 0.f), qy = fmaxf(ay, 0.f), qz = fmaxf(az, 0.f);
            const float outside = fast_sqrt(fmaxf(qx * qx + qy * qy + qz * qz, 1e-30f));
            const float mx = fmaxf(ax, fmaxf(ay, az));
            const float sd = outside + fminf(mx, 0.f);
            const bool better = sd < best[i];
            const bool out = mx > 0.f;
            const bool ixm = (ax >= ay) && (ax >= az);
            const bool iym = !ixm && (ay >= az);
            const float nx = out ? copysignf(qx, px) : (ixm ? copysignf(1.f, px) : 0.f);
            const float ny = out ? copysignf(qy, py) : (iym ? copysignf(1.f, py) : 0.f);
            const float nz = out ? copysignf(qz, pz) : ((!ixm && !iym) ? copysignf(1.f, pz) : 0.f);
            vx[i] = better ? nx : vx[i];
            vy[i] = better ? ny : vy[i];
            vz[i] = better ? nz : vz[i];
            vn[i] = better ? (out ? outside : 1.f) : vn[i];
            best[i] = fminf(best[i], sd);
        }
    }
}

// does the grid pay for the GRADIENT evaluators?  A point robot has ONE collision sphere per waypoint: against a
// few dozen obstacles the straight SGPR-operand loop of point_cost is cheaper than a grid lookup (C2: 3.5 vs 4.3 us
// per CHOMP iteration); articulated robots (tens of spheres per waypoint) always gain.
__device__ __forceinline__ bool grid_usable_grad(const GeomView& G) {
    return grid_usable(G) && (G.kind != MPB_KIND_POINT || G.n_sph > 48);
}

// cost and d cost / d q of one waypoint through the broad-phase grid (dq[i] for i < n_dof)
__device__ __forceinline__ float waypoint_cost_grid_grad(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                         const float (&q)[MPB_MAX_DOF], float (&dq)[MPB_MAX_DOF]) {
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
    if (G.kind == MPB_KIND_POINT) {
        const float x[1] = {q[0]}, y[1] = {q[1]}, z[1] = {(G.n_dof > 2) ? q[2] : 0.f};
        float best[1], vx[1], vy[1], vz[1], vn[1];
        spheres_nearest_grid<1>(G, gridw, otab, x, y, z, best, vx, vy, vz, vn);
        const float h = fmaxf(G.margin + G.links[4] - best[0], 0.f);
        const float sc = (h > 0.f) ? -1.0f / vn[0] : 0.f;
        dq[0] = vx[0] * sc;
        dq[1] = vy[0] * sc;
        if (G.n_dof > 2) dq[2] = vz[0] * sc;
        return h;
    }
    constexpr int N = 4;
    constexpr float FAR = 1.0e9f;
    FKState<true> F;
    F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
    F.tx = F.ty = F.tz = 0.f;
    F.frame = 0;
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) { F.zx[i] = F.zy[i] = F.zz[i] = F.px[i] = F.py[i] = F.pz[i] = 0.f; }
    float cost = 0.f;
    for (int l0 = 0; l0 < G.n_links; l0 += N) {
        const int nl = min(N, G.n_links - l0);
#ifndef MPB_NO_COST_PRIO
        // issue priority by the groups still to come (wave-uniform; see model_group_positions)
        switch ((G.n_links - l0 - 1) / N) {
            case 0: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(0)); break;
            case 1: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(1)); break;
            case 2: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(2)); break;
            default: __builtin_amdgcn_s_setprio(MPB_COST_PRIO(3)); break;
        }
#endif
        float x[N], y[N], z[N], rl[N];
        int fr[N];
        float4 lkv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lkv[i] = *reinterpret_cast<const float4*>(G.links + 8 * (l0 + i));
            rl[i] = G.links[8 * (l0 + i) + 4];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i < nl) {
                const float4 lk = lkv[i];
                const int f = __float_as_int(lk.x);
                while (F.frame < f) fk_advance<true>(G, F, q);
                x[i] = mad3(F.r00, lk.y, F.r01, lk.z, F.r02, lk.w, F.tx);
                y[i] = mad3(F.r10, lk.y, F.r11, lk.z, F.r12, lk.w, F.ty);
                z[i] = mad3(F.r20, lk.y, F.r21, lk.z, F.r22, lk.w, F.tz);
                fr[i] = f;
            } else {
                rl[i] = 0.f;
                x[i] = y[i] = z[i] = FAR;
                fr[i] = 0;
            }
        }
        float best[N], vx[N], vy[N], vz[N], vn[N];
        spheres_nearest_grid<N>(G, gridw, otab, x, y, z, best, vx, vy, vz, vn);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float h = fmaxf(G.margin + rl[i] - best[i], 0.f);   // parked slots: best = 3e38 -> 0
            cost += h;
            if (__any(h > 0.f)) {
                const float sc = (h > 0.f) ? -1.0f / vn[i] : 0.f;
                const float fx = vx[i] * sc, fy = vy[i] * sc, fz = vz[i] * sc;
#pragma unroll
                for (int ii = 0; ii < MPB_MAX_DOF; ++ii) {
                    if (ii < fr[i] && ii < G.n_dof) {
                        // d x / d q_ii = z_ii x (x - p_ii) for every joint upstream of the sphere's frame
                        dq[ii] += joint_term(F.zx[ii], F.zy[ii], F.zz[ii], F.px[ii], F.py[ii], F.pz[ii], x[i], y[i], z[i], fx, fy, fz);
                    }
                }
            }
        }
    }
#ifndef MPB_NO_COST_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return cost;
}


// ------------------------------------------------------------------------------------------------
// Compile-time robot model, GRADIENT path: waypoint_cost_grid_grad with the chain unrolled and the constant transforms /
// sphere offsets folded (see waypoint_cost_grid_model).  The arithmetic is that of fk_advance<true> /
// waypoint_cost_grid_grad expression by expression (products with exact 0 / 1 dropped), the hinges and the joint terms are
// added in the same order -- both walks return the same bits (tests/test_gpu_parity_ops.py).  What the model saves
// besides the folded arithmetic: no scalar loads on the chain, no `joint < frame` predicates (a sphere's frame is a
// constant: its joint loop is exactly as long as its chain), no select chains for q_j and for the joint tables.
// ------------------------------------------------------------------------------------------------
template <class M>
struct ModelFKGrad : ModelFK {
    float zx[M::N_DOF], zy[M::N_DOF], zz[M::N_DOF], px[M::N_DOF], py[M::N_DOF], pz[M::N_DOF];
};

template <class M, int J>
__device__ __forceinline__ void model_fk_advance_grad(ModelFKGrad<M>& F, const float (&q)[MPB_MAX_DOF]) {
    constexpr float p0x = M::TF[J][0], p0y = M::TF[J][1], p0z = M::TF[J][2], p0w = M::TF[J][3];
    constexpr float p1x = M::TF[J][4], p1y = M::TF[J][5], p1z = M::TF[J][6], p1w = M::TF[J][7];
    constexpr float p2x = M::TF[J][8], p2y = M::TF[J][9], p2z = M::TF[J][10], p2w = M::TF[J][11];
    const float ntx = mad3(F.r00, p0w, F.r01, p1w, F.r02, p2w, F.tx);
    const float nty = mad3(F.r10, p0w, F.r11, p1w, F.r12, p2w, F.ty);
    const float ntz = mad3(F.r20, p0w, F.r21, p1w, F.r22, p2w, F.tz);
    F.tx = ntx; F.ty = nty; F.tz = ntz;
    float a00 = dot3(F.r00, p0x, F.r01, p1x, F.r02, p2x), a01 = dot3(F.r00, p0y, F.r01, p1y, F.r02, p2y),
          a02 = dot3(F.r00, p0z, F.r01, p1z, F.r02, p2z);
    float a10 = dot3(F.r10, p0x, F.r11, p1x, F.r12, p2x), a11 = dot3(F.r10, p0y, F.r11, p1y, F.r12, p2y),
          a12 = dot3(F.r10, p0z, F.r11, p1z, F.r12, p2z);
    float a20 = dot3(F.r20, p0x, F.r21, p1x, F.r22, p2x), a21 = dot3(F.r20, p0y, F.r21, p1y, F.r22, p2y),
          a22 = dot3(F.r20, p0z, F.r21, p1z, F.r22, p2z);
    if constexpr (J < M::N_DOF) {
        float sn, cs;
        fast_sincos(q[J], sn, cs);
        const float n00 = fmaf(a01, sn, a00 * cs), n01 = fmaf(-a00, sn, a01 * cs);
        const float n10 = fmaf(a11, sn, a10 * cs), n11 = fmaf(-a10, sn, a11 * cs);
        const float n20 = fmaf(a21, sn, a20 * cs), n21 = fmaf(-a20, sn, a21 * cs);
        a00 = n00; a01 = n01; a10 = n10; a11 = n11; a20 = n20; a21 = n21;
        F.zx[J] = a02; F.zy[J] = a12; F.zz[J] = a22;          // joint axis and origin in the base frame
        F.px[J] = F.tx; F.py[J] = F.ty; F.pz[J] = F.tz;
    }
    F.r00 = a00; F.r01 = a01; F.r02 = a02; F.r10 = a10; F.r11 = a11; F.r12 = a12;
    F.r20 = a20; F.r21 = a21; F.r22 = a22;
}

// positions of the collision spheres of group GRP (model_group_positions with the gradient state)
template <class M, int GRP>
__device__ __forceinline__ bool model_group_positions_grad(ModelFKGrad<M>& F, const float (&q)[MPB_MAX_DOF], unsigned keep,
                                                           float (&x)[4], float (&y)[4], float (&z)[4], float (&rl)[4]) {
    constexpr float FAR = 1.0e9f;
    constexpr int G1 = (M::N_FRAME1 + 3) / 4;
    constexpr bool first = GRP < G1;
    constexpr int lo = first ? 4 * GRP : M::N_FRAME1 + 4 * (GRP - G1);
    constexpr int part_end = first ? M::N_FRAME1 : M::N_LINKS;
    constexpr int hi = (lo + 4 < part_end) ? lo + 4 : part_end;
#ifndef MPB_NO_COST_PRIO
    {
        constexpr int NG_ = (M::N_FRAME1 + 3) / 4 + (M::N_LINKS - M::N_FRAME1 + 3) / 4;
        __builtin_amdgcn_s_setprio(MPB_COST_PRIO(NG_ - 1 - GRP));
    }
#endif
    static_for<lo, hi>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int f = M::LINK_FRAME[l];
        constexpr int fprev = (l == 0) ? 0 : M::LINK_FRAME[l > 0 ? l - 1 : 0];
        static_for<fprev, f>([&](auto jc) { model_fk_advance_grad<M, decltype(jc)::value>(F, q); });
        constexpr int slot = l - lo;
        constexpr float ox = M::LINK[l][0], oy = M::LINK[l][1], oz = M::LINK[l][2], rad = M::LINK[l][3];
        const float px = mad3(F.r00, ox, F.r01, oy, F.r02, oz, F.tx);
        const float py = mad3(F.r10, ox, F.r11, oy, F.r12, oz, F.ty);
        const float pz = mad3(F.r20, ox, F.r21, oy, F.r22, oz, F.tz);
        if constexpr (first) {
            const bool on = (keep >> l) & 1u;                // wave-uniform
            x[slot] = on ? px : FAR; y[slot] = on ? py : FAR; z[slot] = on ? pz : FAR; rl[slot] = on ? rad : 0.f;
        } else {
            x[slot] = px; y[slot] = py; z[slot] = pz; rl[slot] = rad;
        }
    });
#pragma unroll
    for (int i = hi - lo; i < 4; ++i) { x[i] = y[i] = z[i] = FAR; rl[i] = 0.f; }
    if constexpr (first) {
        constexpr unsigned gmask = ((hi >= 32) ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        return (keep & gmask) != 0u;
    }
    return true;
}

// J^T f of the spheres of group GRP: slot i (sphere lo + i, on frame LINK_FRAME) pulls on the joints 0 .. frame - 1
template <class M, int GRP>
__device__ __forceinline__ void model_group_jtf(const ModelFKGrad<M>& F, const GeomView& G, const float (&x)[4],
                                                const float (&y)[4], const float (&z)[4], const float (&rl)[4],
                                                const float (&best)[4], const float (&vx)[4], const float (&vy)[4],
                                                const float (&vz)[4], const float (&vn)[4], float& cost,
                                                float (&dq)[MPB_MAX_DOF]) {
    constexpr int G1 = (M::N_FRAME1 + 3) / 4;
    constexpr bool first = GRP < G1;
    constexpr int lo = first ? 4 * GRP : M::N_FRAME1 + 4 * (GRP - G1);
    constexpr int part_end = first ? M::N_FRAME1 : M::N_LINKS;
    constexpr int hi = (lo + 4 < part_end) ? lo + 4 : part_end;
    static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const float h = fmaxf(G.margin + rl[i] - best[i], 0.f);   // parked slots: best = 3e38 -> 0
        cost += h;
        if constexpr (lo + i < hi) {
            constexpr int fr = M::LINK_FRAME[lo + i];
            constexpr int nj = fr < M::N_DOF ? fr : M::N_DOF;     // joints upstream of the sphere's frame
            if (__any(h > 0.f)) {
                const float sc = (h > 0.f) ? -1.0f / vn[i] : 0.f;
                const float fx = vx[i] * sc, fy = vy[i] * sc, fz = vz[i] * sc;
                static_for<0, nj>([&](auto jc) {
                    constexpr int jj = decltype(jc)::value;
                    dq[jj] += joint_term(F.zx[jj], F.zy[jj], F.zz[jj], F.px[jj], F.py[jj], F.pz[jj], x[i], y[i], z[i], fx, fy, fz);
                });
            }
        }
    });
}

template <class M, int... GRPS>
__device__ __forceinline__ bool model_group_dispatch_grad(int grp, ModelFKGrad<M>& F, const float (&q)[MPB_MAX_DOF], unsigned keep,
                                                          float (&x)[4], float (&y)[4], float (&z)[4], float (&rl)[4],
                                                          std::integer_sequence<int, GRPS...>) {
    bool run = false;
    ((grp == GRPS ? (void)(run = model_group_positions_grad<M, GRPS>(F, q, keep, x, y, z, rl)) : (void)0), ...);
    return run;
}
template <class M, int... GRPS>
__device__ __forceinline__ void model_group_dispatch_jtf(int grp, const ModelFKGrad<M>& F, const GeomView& G, const float (&x)[4],
                                                         const float (&y)[4], const float (&z)[4], const float (&rl)[4],
                                                         const float (&best)[4], const float (&vx)[4], const float (&vy)[4],
                                                         const float (&vz)[4], const float (&vn)[4], float& cost,
                                                         float (&dq)[MPB_MAX_DOF], std::integer_sequence<int, GRPS...>) {
    ((grp == GRPS ? model_group_jtf<M, GRPS>(F, G, x, y, z, rl, best, vx, vy, vz, vn, cost, dq) : (void)0), ...);
}

template <class M>
__device__ __forceinline__ float waypoint_cost_grid_grad_model(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                               const float (&q)[MPB_MAX_DOF], float (&dq)[MPB_MAX_DOF]) {
    constexpr int NG = (M::N_FRAME1 + 3) / 4 + (M::N_LINKS - M::N_FRAME1 + 3) / 4;
#pragma unroll
    for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
    ModelFKGrad<M> F;
    F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
    F.tx = F.ty = F.tz = 0.f;
#pragma unroll
    for (int i = 0; i < M::N_DOF; ++i) { F.zx[i] = F.zy[i] = F.zz[i] = F.px[i] = F.py[i] = F.pz[i] = 0.f; }
    float cost = 0.f;
    const unsigned keep = G.keep_mask;
#pragma nounroll
    for (int grp = 0; grp < NG; ++grp) {
        float x[4], y[4], z[4], rl[4];
        const bool run = model_group_dispatch_grad<M>(grp, F, q, keep, x, y, z, rl, std::make_integer_sequence<int, NG>{});
        if (run) {
            float best[4], vx[4], vy[4], vz[4], vn[4];
            spheres_nearest_grid<4>(G, gridw, otab, x, y, z, best, vx, vy, vz, vn);
            model_group_dispatch_jtf<M>(grp, F, G, x, y, z, rl, best, vx, vy, vz, vn, cost, dq, std::make_integer_sequence<int, NG>{});
        }
    }
#ifndef MPB_NO_COST_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    return cost;
}

// the gradient evaluator a kernel should call for this field: the compile-time model when the buffer carries its id
__device__ __forceinline__ float waypoint_cost_grid_grad_any(const GeomView& G, const unsigned* gridw, const float4* otab,
                                                             const float (&q)[MPB_MAX_DOF], float (&dq)[MPB_MAX_DOF]) {
    if (G.model == PandaModel::ID) return waypoint_cost_grid_grad_model<PandaModel>(G, gridw, otab, q, dq);   // wave-uniform
    return waypoint_cost_grid_grad(G, gridw, otab, q, dq);
}
