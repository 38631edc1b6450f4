// mpb_points.hip -- the robot / field API the reference's cost layer CONSUMES, as separate device ops:
//   robot.fk_map_collision(q_pos)            (call site cost_functions.py:52)  -> positions of the collision spheres
//   field.compute_cost(q_pos, link_pos, ...) (call site field_factor.py:39,52) -> hinge cost per waypoint
// and their vector-Jacobian products, so that torch.autograd can differentiate through them the way the reference
// differentiates through torch_robotics (field_factor.py:54).  The planners of this package never call these: they
// use the fused FK+SDF evaluators of mpb_geom.h.  These ops exist so that the reference's UNMODIFIED cost classes
// (CostCollision / FieldFactor) run against this package's robot and field objects on GPU tensors.
// Mapping: one wave per trajectory, one lane per waypoint, like every cost kernel here.
#include "mpb_common.h"
#include "mpb_geom.h"

// ---- forward kinematics of the collision spheres: q (B,H,d) -> pts (B,H,L,3) ----------------------------
__global__ __launch_bounds__(256) void fk_points_kernel(const float* __restrict__ q_in, const float* __restrict__ geom,
                                                        float* __restrict__ pts, int B, int H, int d) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const GeomView G = geom_view(geom);
    const int D = G.n_dof, Ln = G.n_links;
    for (int h = lane; h < H; h += 64) {
        const float* row = q_in + ((size_t)b * H + h) * d;
        float q[MPB_MAX_DOF];
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) q[i] = (i < D) ? row[i] : 0.f;
        float* o = pts + ((size_t)b * H + h) * Ln * 3;
        if (G.kind == MPB_KIND_POINT) {
            o[0] = q[0]; o[1] = q[1]; o[2] = (D > 2) ? q[2] : 0.f;
            continue;
        }
        FKState<false> F;
        F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
        F.tx = F.ty = F.tz = 0.f;
        F.frame = 0;
        for (int l = 0; l < Ln; ++l) {
            const float4 lk = *reinterpret_cast<const float4*>(G.links + 8 * l);   // frame, ox, oy, oz
            const int f = __float_as_int(lk.x);
            while (F.frame < f) fk_advance<false>(G, F, q);
            o[3 * l + 0] = mad3(F.r00, lk.y, F.r01, lk.z, F.r02, lk.w, F.tx);
            o[3 * l + 1] = mad3(F.r10, lk.y, F.r11, lk.z, F.r12, lk.w, F.ty);
            o[3 * l + 2] = mad3(F.r20, lk.y, F.r21, lk.z, F.r22, lk.w, F.tz);
        }
    }
}

// ---- its vector-Jacobian product: gq (B,H,D) = J^T gpts, d x_l / d q_i = z_i x (x_l - p_i) for joints upstream --------
__global__ __launch_bounds__(256) void fk_points_vjp_kernel(const float* __restrict__ q_in, const float* __restrict__ geom,
                                                            const float* __restrict__ gpts, float* __restrict__ gq, int B,
                                                            int H, int d) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const GeomView G = geom_view(geom);
    const int D = G.n_dof, Ln = G.n_links;
    for (int h = lane; h < H; h += 64) {
        const float* row = q_in + ((size_t)b * H + h) * d;
        const float* g = gpts + ((size_t)b * H + h) * Ln * 3;
        float q[MPB_MAX_DOF], dq[MPB_MAX_DOF];
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i) {
            q[i] = (i < D) ? row[i] : 0.f;
            dq[i] = 0.f;
        }
        if (G.kind == MPB_KIND_POINT) {
            dq[0] = g[0]; dq[1] = g[1];
            if (D > 2) dq[2] = g[2];
        } else {
            FKState<true> F;
            F.r00 = 1.f; F.r01 = 0.f; F.r02 = 0.f; F.r10 = 0.f; F.r11 = 1.f; F.r12 = 0.f; F.r20 = 0.f; F.r21 = 0.f; F.r22 = 1.f;
            F.tx = F.ty = F.tz = 0.f;
            F.frame = 0;
#pragma unroll
            for (int i = 0; i < MPB_MAX_DOF; ++i) { F.zx[i] = F.zy[i] = F.zz[i] = F.px[i] = F.py[i] = F.pz[i] = 0.f; }
            for (int l = 0; l < Ln; ++l) {
                const float4 lk = *reinterpret_cast<const float4*>(G.links + 8 * l);
                const int f = __float_as_int(lk.x);
                while (F.frame < f) fk_advance<true>(G, F, q);
                const float x = mad3(F.r00, lk.y, F.r01, lk.z, F.r02, lk.w, F.tx);
                const float y = mad3(F.r10, lk.y, F.r11, lk.z, F.r12, lk.w, F.ty);
                const float z = mad3(F.r20, lk.y, F.r21, lk.z, F.r22, lk.w, F.tz);
                const float fx = g[3 * l], fy = g[3 * l + 1], fz = g[3 * l + 2];
#pragma unroll
                for (int ii = 0; ii < MPB_MAX_DOF; ++ii) {
                    if (ii < f && ii < D) {
                        const float ex = x - F.px[ii], ey = y - F.py[ii], ez = z - F.pz[ii];
                        const float cx = F.zy[ii] * ez - F.zz[ii] * ey;
                        const float cy = F.zz[ii] * ex - F.zx[ii] * ez;
                        const float cz = F.zx[ii] * ey - F.zy[ii] * ex;
                        dq[ii] += fx * cx + fy * cy + fz * cz;
                    }
                }
            }
        }
        float* o = gq + ((size_t)b * H + h) * D;
#pragma unroll
        for (int i = 0; i < MPB_MAX_DOF; ++i)
            if (i < D) o[i] = dq[i];
    }
}

// ---- field cost of given collision-sphere positions: pts (B,H,L,3) -> cost (B,H) [and d cost / d pts] -----------------
template <bool GRAD>
__global__ __launch_bounds__(256) void points_cost_kernel(const float* __restrict__ pts, const float* __restrict__ geom,
                                                          const float* __restrict__ gout, float* __restrict__ cost,
                                                          float* __restrict__ gpts, int B, int H) {
    __shared__ unsigned gridw[MPB_GRID_MAX_CELLS];
    __shared__ float4 otab[MPB_GRID_MAX_SPH + 1];
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const GeomView G = geom_view(geom);
    const bool ug = grid_usable(G);
    if (ug) grid_stage(G, gridw, otab, threadIdx.x, blockDim.x);
    __syncthreads();
    if (b >= B) return;
    const int Ln = G.n_links;
    for (int h = lane; h < H; h += 64) {
        const float* p = pts + ((size_t)b * H + h) * Ln * 3;
        float c = 0.f;
        const float go = GRAD ? gout[(size_t)b * H + h] : 0.f;
        for (int l = 0; l < Ln; ++l) {
            const float x[1] = {p[3 * l]}, y[1] = {p[3 * l + 1]}, z[1] = {p[3 * l + 2]};
            float best[1], vx[1], vy[1], vz[1], vn[1];
            if (ug) {
                spheres_nearest_grid<1>(G, gridw, otab, x, y, z, best, vx, vy, vz, vn);
            } else {
                float gx, gy, gz;
                // exhaustive evaluator of one sphere (point_cost keeps margin / radius inside: undo to get `best`)
                LinkChunk<true> C;
#pragma unroll
                for (int i = 0; i < LinkChunk<true>::N; ++i) {
                    C.x[i] = (i == 0) ? x[0] : 1.0e9f; C.y[i] = (i == 0) ? y[0] : 1.0e9f; C.z[i] = (i == 0) ? z[0] : 1.0e9f;
                    C.xx[i] = C.x[i] * C.x[i] + C.y[i] * C.y[i] + C.z[i] * C.z[i];
                    C.best[i] = 3.0e38f; C.vx[i] = C.vy[i] = C.vz[i] = 0.f; C.vn[i] = 1.f;
                }
                CullStats cs = {0, 0, false};
                chunk_vs_obstacles<true>(G, C, cs);
                best[0] = C.best[0]; vx[0] = C.vx[0]; vy[0] = C.vy[0]; vz[0] = C.vz[0]; vn[0] = C.vn[0];
                (void)gx; (void)gy; (void)gz;
            }
            const float hng = fmaxf(G.margin + G.links[8 * l + 4] - best[0], 0.f);
            c += hng;
            if (GRAD) {
                const float sc = (hng > 0.f) ? -go / vn[0] : 0.f;
                float* o = gpts + ((size_t)b * H + h) * Ln * 3 + 3 * l;
                o[0] = vx[0] * sc; o[1] = vy[0] * sc; o[2] = vz[0] * sc;
            }
        }
        if (!GRAD) cost[(size_t)b * H + h] = c;
    }
}

extern "C" int mpb_fk_collision_points(const float* q, const float* geom, float* pts, int B, int H, int d, void* stream) {
    if (B < 0 || H < 1 || d < 1 || d > 2 * MPB_MAX_DOF) return mpb_fail(MPB_E_INVALID, "mpb_fk_collision_points: bad shape");
    if (B == 0) return MPB_OK;
    if (!q || !geom || !pts) return mpb_fail(MPB_E_INVALID, "mpb_fk_collision_points: null pointer");
    hipLaunchKernelGGL(fk_points_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, geom, pts, B, H, d);
    return mpb_check_launch("mpb_fk_collision_points");
}

extern "C" int mpb_fk_collision_points_vjp(const float* q, const float* geom, const float* grad_pts, float* grad_q, int B,
                                           int H, int d, void* stream) {
    if (B < 0 || H < 1 || d < 1 || d > 2 * MPB_MAX_DOF) return mpb_fail(MPB_E_INVALID, "mpb_fk_collision_points_vjp: bad shape");
    if (B == 0) return MPB_OK;
    if (!q || !geom || !grad_pts || !grad_q) return mpb_fail(MPB_E_INVALID, "mpb_fk_collision_points_vjp: null pointer");
    hipLaunchKernelGGL(fk_points_vjp_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, geom, grad_pts, grad_q, B,
                       H, d);
    return mpb_check_launch("mpb_fk_collision_points_vjp");
}

extern "C" int mpb_field_cost_points(const float* pts, const float* geom, float* cost, int B, int H, void* stream) {
    if (B < 0 || H < 1) return mpb_fail(MPB_E_INVALID, "mpb_field_cost_points: bad shape");
    if (B == 0) return MPB_OK;
    if (!pts || !geom || !cost) return mpb_fail(MPB_E_INVALID, "mpb_field_cost_points: null pointer");
    hipLaunchKernelGGL(points_cost_kernel<false>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, pts, geom, nullptr, cost,
                       nullptr, B, H);
    return mpb_check_launch("mpb_field_cost_points");
}

extern "C" int mpb_field_cost_points_vjp(const float* pts, const float* geom, const float* grad_cost, float* grad_pts, int B,
                                         int H, void* stream) {
    if (B < 0 || H < 1) return mpb_fail(MPB_E_INVALID, "mpb_field_cost_points_vjp: bad shape");
    if (B == 0) return MPB_OK;
    if (!pts || !geom || !grad_cost || !grad_pts) return mpb_fail(MPB_E_INVALID, "mpb_field_cost_points_vjp: null pointer");
    hipLaunchKernelGGL(points_cost_kernel<true>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, pts, geom, grad_cost,
                       nullptr, grad_pts, B, H);
    return mpb_check_launch("mpb_field_cost_points_vjp");
}
