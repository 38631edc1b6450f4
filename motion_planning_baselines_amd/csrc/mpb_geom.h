// mpb_geom.h -- device-side geometry: packed buffer view, forward kinematics of the collision spheres,
// sphere / box signed distance, hinge cost and its analytic gradient.
//
// Build-defined back-end standing in for torch_robotics (reference call sites
// cost_functions.py:50-52 robot.fk_map_collision, field_factor.py:39 field.compute_cost); the same
// definitions are evaluated on the CPU by oracle/geometry_ref.py for checking.
//
// Mapping to the hardware: one LANE evaluates one waypoint; every geometry word (joint transforms,
// sphere offsets, obstacle centres) is addressed with wave-uniform indices through a const
// __restrict__ kernel-argument pointer, so hipcc turns those reads into scalar loads (s_load_dwordx4)
// and the values ride in SGPRs as VALU operands -- no LDS traffic and no VGPRs for constants.
#pragma once
#include <hip/hip_runtime.h>

#define MPB_GEOM_MAGIC 0x4D504247
#define MPB_GEOM_VERSION 1
#define MPB_GEOM_HEADER_WORDS 16
#define MPB_KIND_POINT 0
#define MPB_KIND_CHAIN 1
#define MPB_MAX_DOF 8
#define MPB_MAX_TF (MPB_MAX_DOF + 1)

struct GeomView {
    int kind, n_dof, n_tf, n_links, n_sph, n_box;
    float margin;
    const float* tf;     // n_tf x 12
    const float* links;  // n_links x 8: frame(int), ox, oy, oz, r, 0,0,0
    const float* sph;    // n_sph x 4
    const float* box;    // n_box x 8: cx,cy,cz,0,hx,hy,hz,0
};

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
    return v;
}

// sin and cos together, |x| up to a few hundred: Cody-Waite reduction by pi/2 in three fma steps, then
// the cephes single-precision minimax polynomials on [-pi/4, pi/4] (max abs error 9e-8, measured against
// fp64).  ~25 VALU instructions for both values; ocml sinf + cosf cost ~4x that because of their
// huge-argument path.  Joint angles are bounded by the joint limits plus STOMP noise.
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

// v_sqrt_f32: 1 ulp, one quarter-rate instruction (the IEEE-correct expansion hipcc emits for sqrtf is
// ~20 VALU instructions and dominated the obstacle loop).  Arguments here are squared distances in
// [0, ~10]; well inside the range where the raw instruction needs no scaling.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// min over obstacles of the signed distance at x; for GRAD also the un-normalised direction (vx,vy,vz)
// and its norm vn such that grad sdf = v / vn (v = 0 at degenerate points -> zero sub-gradient, as the
// oracle's clamped norm).
template <bool GRAD>
__device__ __forceinline__ void sphere_sd(const float4 s, float x, float y, float z, float& best, float& vx,
                                          float& vy, float& vz, float& vn) {
    const float dx = x - s.x, dy = y - s.y, dz = z - s.z;
    float d2 = dx * dx + dy * dy + dz * dz;
    if (GRAD) d2 = fmaxf(d2, 1e-30f);  // keeps 1/dist finite; the oracle clamps the same way
    const float dist = fast_sqrt(d2);
    const float sd = dist - s.w;
    if (GRAD) {
        const bool better = sd < best;
        vx = better ? dx : vx; vy = better ? dy : vy; vz = better ? dz : vz; vn = better ? dist : vn;
    }
    best = fminf(best, sd);
}

template <bool GRAD>
__device__ __forceinline__ float min_signed_distance(const GeomView& G, float x, float y, float z, float& vx,
                                                     float& vy, float& vz, float& vn) {
    float best = 3.0e38f;
    if (GRAD) { vx = vy = vz = 0.f; vn = 1.f; }
    const float4* sp = reinterpret_cast<const float4*>(G.sph);
    // 8 obstacles per trip: two s_load_dwordx16 feed 8 x ~7 VALU instructions, all operands in SGPRs
#pragma unroll 8
    for (int o = 0; o < G.n_sph; ++o) sphere_sd<GRAD>(sp[o], x, y, z, best, vx, vy, vz, vn);
    const float4* bp = reinterpret_cast<const float4*>(G.box);
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
            if (better) {
                if (mx > 0.f) {  // outside: direction of the clamped offset
                    vx = copysignf(qx, px); vy = copysignf(qy, py); vz = copysignf(qz, pz); vn = outside;
                } else {         // inside: unit axis of the largest component (first on ties, as torch.max)
                    const bool ix = (ax >= ay) && (ax >= az);
                    const bool iy = !ix && (ay >= az);
                    vx = ix ? copysignf(1.f, px) : 0.f;
                    vy = iy ? copysignf(1.f, py) : 0.f;
                    vz = (!ix && !iy) ? copysignf(1.f, pz) : 0.f;
                    vn = 1.f;
                }
            }
        }
        best = fminf(best, sd);
    }
    return best;
}

// hinge cost of one collision sphere; for GRAD (fx,fy,fz) = d hinge / d x.
template <bool GRAD>
__device__ __forceinline__ float sphere_hinge(const GeomView& G, float x, float y, float z, float rl,
                                              float& fx, float& fy, float& fz) {
    float vx, vy, vz, vn;
    const float sd = min_signed_distance<GRAD>(G, x, y, z, vx, vy, vz, vn);
    const float h = fmaxf(G.margin + rl - sd, 0.f);
    if (GRAD) {
        const float s = (h > 0.f) ? -1.0f / vn : 0.f;
        fx = vx * s; fy = vy * s; fz = vz * s;
    }
    return h;
}

// Collision cost of one waypoint q[0..D) (sum over the robot's collision spheres); for GRAD
// dq[i] = d cost / d q_i (i < D).  q / dq are register arrays indexed only with compile-time indices.
template <bool GRAD>
__device__ __forceinline__ float waypoint_cost(const GeomView& G, const float (&q)[MPB_MAX_DOF],
                                               float (&dq)[MPB_MAX_DOF]) {
    if (GRAD) {
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) dq[i] = 0.f;
    }
    if (G.kind == MPB_KIND_POINT) {
        float fx, fy, fz;
        const float rl = G.links[4];
        const float z = (G.n_dof > 2) ? q[2] : 0.f;
        const float h = sphere_hinge<GRAD>(G, q[0], q[1], z, rl, fx, fy, fz);
        if (GRAD) { dq[0] = fx; dq[1] = fy; if (G.n_dof > 2) dq[2] = fz; }
        return h;
    }
    // serial revolute chain: frame_{j+1} = frame_j * P_j * Rz(q_j)
    float r00 = 1.f, r01 = 0.f, r02 = 0.f, r10 = 0.f, r11 = 1.f, r12 = 0.f, r20 = 0.f, r21 = 0.f, r22 = 1.f;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    float zx[MPB_MAX_DOF], zy[MPB_MAX_DOF], zz[MPB_MAX_DOF], px[MPB_MAX_DOF], py[MPB_MAX_DOF], pz[MPB_MAX_DOF];
    float cost = 0.f;
    int l = 0;
#pragma unroll
    for (int j = 0; j < MPB_MAX_TF; ++j) {
        if (j < G.n_tf) {
            const float4* P = reinterpret_cast<const float4*>(G.tf + 12 * j);
            const float4 p0 = P[0], p1 = P[1], p2 = P[2];  // rows of the 3x4 constant transform
            // t += R * P[:,3]
            const float ntx = tx + (r00 * p0.w + r01 * p1.w + r02 * p2.w);
            const float nty = ty + (r10 * p0.w + r11 * p1.w + r12 * p2.w);
            const float ntz = tz + (r20 * p0.w + r21 * p1.w + r22 * p2.w);
            tx = ntx; ty = nty; tz = ntz;
            // R = R * P[:, :3]
            float a00 = r00 * p0.x + r01 * p1.x + r02 * p2.x, a01 = r00 * p0.y + r01 * p1.y + r02 * p2.y,
                  a02 = r00 * p0.z + r01 * p1.z + r02 * p2.z;
            float a10 = r10 * p0.x + r11 * p1.x + r12 * p2.x, a11 = r10 * p0.y + r11 * p1.y + r12 * p2.y,
                  a12 = r10 * p0.z + r11 * p1.z + r12 * p2.z;
            float a20 = r20 * p0.x + r21 * p1.x + r22 * p2.x, a21 = r20 * p0.y + r21 * p1.y + r22 * p2.y,
                  a22 = r20 * p0.z + r21 * p1.z + r22 * p2.z;
            if (j < MPB_MAX_DOF && j < G.n_dof) {
                const int jj = j < MPB_MAX_DOF ? j : 0;  // compile-time constant after unrolling
                float sn, cs;
                fast_sincos(q[jj], sn, cs);
                const float n00 = a00 * cs + a01 * sn, n01 = a01 * cs - a00 * sn;
                const float n10 = a10 * cs + a11 * sn, n11 = a11 * cs - a10 * sn;
                const float n20 = a20 * cs + a21 * sn, n21 = a21 * cs - a20 * sn;
                a00 = n00; a01 = n01; a10 = n10; a11 = n11; a20 = n20; a21 = n21;
                if (GRAD) {
                    zx[jj] = a02; zy[jj] = a12; zz[jj] = a22;
                    px[jj] = tx; py[jj] = ty; pz[jj] = tz;
                }
            }
            r00 = a00; r01 = a01; r02 = a02; r10 = a10; r11 = a11; r12 = a12; r20 = a20; r21 = a21; r22 = a22;
            // collision spheres rigidly attached to frame j+1 (sorted by frame on the host)
            while (l < G.n_links && __float_as_int(G.links[8 * l]) == j + 1) {
                const float4 lk = *reinterpret_cast<const float4*>(G.links + 8 * l);  // frame, ox, oy, oz
                const float rl = G.links[8 * l + 4];
                const float x = tx + (r00 * lk.y + r01 * lk.z + r02 * lk.w);
                const float y = ty + (r10 * lk.y + r11 * lk.z + r12 * lk.w);
                const float z = tz + (r20 * lk.y + r21 * lk.z + r22 * lk.w);
                float fx, fy, fz;
                const float h = sphere_hinge<GRAD>(G, x, y, z, rl, fx, fy, fz);
                cost += h;
                if (GRAD) {
                    if (__any(h > 0.f)) {
#pragma unroll
                        for (int i = 0; i < MPB_MAX_DOF; ++i) {
                            if (i <= j && i < G.n_dof) {
                                // d x / d q_i = z_i x (x - p_i);  dq_i += f . (z_i x (x - p_i))
                                const float ex = x - px[i], ey = y - py[i], ez = z - pz[i];
                                const float cx = zy[i] * ez - zz[i] * ey;
                                const float cy = zz[i] * ex - zx[i] * ez;
                                const float cz = zx[i] * ey - zy[i] * ex;
                                dq[i] += fx * cx + fy * cy + fz * cz;
                            }
                        }
                    }
                }
                ++l;
            }
        }
    }
    return cost;
}
