/*
 * mpb.h -- C-ABI of libmpb_hip.so: MI355X (gfx950) kernels for the batched trajectory-optimisation
 * inner loops of anindex/motion_planning_baselines (mp_baselines/planners).
 *
 * The reference has NO FFI of its own (it is pure Python/PyTorch): the seam is the
 * planners' Python methods.  Each entry point below replaces the body of the reference method(s)
 * cited next to it; the classes in motion_planning_baselines_amd/planners keep the reference's class / ctor /
 * optimize() / reset() surface and call these through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated otherwise
 *     (obtained from tensor.data_ptr() of a PyTorch-ROCm tensor); the library never allocates,
 *     frees or retains caller memory and keeps no state between calls (thread-safe);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream; NULL = default);
 *     all work is enqueued asynchronously on it, nothing synchronises the device;
 *   - return value 0 = ok; non-zero = MPB_E_*; mpb_last_error() gives the message of the calling
 *     thread's last failure;
 *   - shapes: P particles, S samples per particle, B = P*S rollouts, H support points (horizon),
 *     D degrees of freedom, d = optimised state width (D if pos_only else 2D).
 *   - `geom` is the packed geometry word buffer (motion_planning_baselines_amd/geometry.py
 *     pack_geometry; layout in csrc/mpb_geom.h): robot kinematics + collision spheres + obstacle
 *     spheres / boxes + hinge margin.  It stands in for the reference's external robot / field objects
 *     (cost_functions.py:50-52 robot.fk_map_collision, field_factor.py:39 field.compute_cost).
 */
#ifndef MPB_H
#define MPB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPB_OK 0
#define MPB_E_INVALID 1     /* bad argument (null pointer, shape out of range, bad geometry header) */
#define MPB_E_UNSUPPORTED 2 /* valid request this build cannot serve (e.g. H too large for LDS) */
#define MPB_E_HIP 3         /* a HIP runtime call failed; see mpb_last_error() */

#define MPB_MAX_H 256
#define MPB_MAX_DOF 12

/* ABI version in the low 16 bits (MPB_ABI_VERSION: bumped whenever a signature of this header changes positionally or a call that
 * used to be accepted is now refused; a binding must refuse a library that reports another number); bit 30 set = a tuning build
 * (compiled with wrong-result timing switches: never a product library).
 * Changes:  4 (round 4)  workspace header of the persistent STOMP kernels 64 -> 1 280 bytes; device-noise streams of STOMP / MPPI
 *                        changed (Philox4x32-7, 23-bit Box-Muller): not reproducible across 3 -> 4;
 *           5 (round 5)  geometry buffers are version 6 (broad-phase grid on a lattice through the origin, header word 31; a
 *                        version-5 buffer is refused by mpb_geom_check); the STOMP entry points refuse pointers that are not
 *                        16-byte aligned; a model-tagged geometry must keep margin + radii below 1 m; mpb_gpmp2_solve applies the
 *                        collision factors by Sherman-Morrison beyond a precision ratio of 1e7 (more accurate results there);
 *           6 (round 6)  mpb_gpmp2_workspace_bytes and mpb_stomp_workspace_bytes return MORE (the low-rank form of the GPMP2 solve keeps
 *                        its tables, right-hand sides and steps in the workspace; the STOMP workspace also fits the generalised persistent
 *                        kernel at H = 64, which serves list-grid scenes): a workspace sized by an older library is too small;
 *                        geometry buffers of version 7 (LIST broad-phase grid, per field) are accepted next to version 6;
 *                        MPB_MAX_DOF 8 -> 12; mpb_gpmp2_solve takes the low-rank form wherever n_fields (H - 1) <= 127 (same
 *                        results to the solver's fp64 rounding). */
#define MPB_ABI_VERSION 6
#define MPB_VERSION_TUNING_BUILD 0x40000000
int mpb_version(void);
const char *mpb_last_error(void);

/* Validate a packed geometry buffer held in HOST memory (n_words 32-bit words). */
int mpb_geom_check(const float *geom_host, int n_words);
/* Properties of a (valid) packed geometry buffer, read from its HOST copy, that let a launcher pick a kernel
 * instantiation without touching device memory.  *flags: bits 0-7 = id of the compile-time robot model
 * (csrc/mpb_model_*.h) every chained field is tagged with AND whose cost-only kernels can run (every field has a
 * usable broad-phase grid); 0 = generic table-driven kernels; bit 8 = every chained field has a usable broad-phase
 * grid (what the persistent STOMP kernel needs); bit 9 = point robot with ONE field of at most 32 spheres and 8 boxes
 * (CHOMP's four-lanes-per-waypoint kernel keeps such an obstacle set in registers); bit 10 = point robot; bit 12 = ONE field (no chain: MPPI's LDS grid, the persistent STOMP kernels' one-field instantiations); bits 16-28 = cells of the largest
 * broad-phase grid of the chain when bit 8 or bit 13 is set (what a kernel that stages the grid in LDS has to reserve); bit 13 (round 6) = every
 * chained field carries a LIST grid (geometry version 7: a field with more than 63 obstacle spheres -- up to 255 spheres + 127 boxes, any number
 * of candidates per cell, boxes culled like spheres; bit 8 is then clear): the generalised persistent STOMP kernel takes such a geometry up to
 * H = 64 (csrc/mpb_stomp_fused_hx.hip, LIST), every other kernel walks such a field exhaustively; the model byte is also set for it, and
 * consumers other than that kernel require bit 8 beside it.  Entry points that take `geom_flags` expect the value
 * computed from the host copy of the very buffer `geom` points to (0 is always valid); a kernel re-checks the tag
 * and the one-field promise against the device header and writes NaN costs if they disagree. */
int mpb_geom_flags(const float *geom_host, int n_words, int *flags);

/* ---------------------------------------------------------------------------------------------
 * Collision cost  -- replaces CostCollision.eval (costs/cost_functions.py:171-189) over
 * get_q_pos_vel_and_fk_map (:41-53) + FieldFactor.get_error (costs/factors/field_factor.py:17-39):
 *   out[b] = weight * (k_sigma * sum_{h >= h_begin} field_cost(q[b,h,:D]))       (h_begin = 1: Q5)
 * trajs (B,H,d); out (B).
 * mpb_cost_collision_grad additionally writes grad (B,H,d) = d out[b] / d trajs[b] (the quantity the
 * reference obtains by autograd: chomp.py:139, field_factor.py:54); velocity channels get 0.
 * per_waypoint (B,H) optional (may be NULL): un-scaled field cost of every waypoint (0 for h < h_begin).
 * geom_flags = mpb_geom_flags(host copy of geom) lets the gradient evaluators pick the compile-time robot model's kernel
 * (same bits as the table-driven walk, fewer instructions and registers); 0 is always valid (table-driven walk).
 * ------------------------------------------------------------------------------------------- */
int mpb_cost_collision_eval(const float *trajs, const float *geom, float *out, float *per_waypoint,
                            int B, int H, int d, int h_begin, float k_sigma, float weight, void *stream);
int mpb_cost_collision_grad(const float *trajs, const float *geom, int geom_flags, float *out, float *grad,
                            int B, int H, int d, int h_begin, float k_sigma, float weight, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Trajectory-only cost terms -- replaces the eval() of CostGP (costs/cost_functions.py:271-289),
 * CostGPTrajectory (:344-354), CostGPTrajectoryPositionOnlyWrapper (:365-368), CostSmoothnessCHOMP
 * (:384-387), CostJointLimits (:406-426) and CostGoalPrior (:523-536); any subset in one pass over the
 * batch (what CostComposite.eval, :70-87, sums term by term):
 *   out[b] (+)= k_gp    * sum_t e_t^T ([[12/dt^3,-6/dt^2],[-6/dt^2,4/dt]] (x) I) e_t, e_t = x_{t+1} - Phi x_t
 *            + k_start * |start_state - x_0|^2            (UnaryFactor, K = I/sigma^2 folded into k_*)
 *            + k_goal  * |goal_states[b / trajs_per_goal] - x_{H-1}|^2
 *            + k_smooth* sum_cols x^T R x, R = CHOMP._get_R_mat(dt, H) (chomp.py:81-101)
 *   *jl_total   = k_jlim * sum over the WHOLE batch of squared joint-limit violations beyond
 *                 q_min + jl_eps / q_max - jl_eps (the reference's `.sum(-1)` of a 1-D gather is a scalar);
 *                 broadcast_jlim != 0 also adds it to every out[b] (what the composite's `+=` does).
 * k_* = composite weight / sigma^2 of the term.  trajs (B,H,d) with d == 2*n_dof, or d == n_dof with
 * MPB_TERM_VEL_FD (velocities = central differences, zero end rows) for the GP term / any d for
 * SMOOTH / d >= n_dof for JLIM.  start_state (2*n_dof), goal_states (G, 2*n_dof), q_min/q_max (n_dof),
 * jl_total: one fp64 word, all device memory; pointers of disabled terms may be NULL.
 * accumulate != 0 adds onto the existing out[b] (e.g. the collision costs mpb_stomp_sample wrote).
 * ------------------------------------------------------------------------------------------- */
#define MPB_TERM_GP 1u
#define MPB_TERM_START 2u
#define MPB_TERM_GOAL 4u
#define MPB_TERM_SMOOTH 8u
#define MPB_TERM_JLIM 16u
#define MPB_TERM_VEL_FD 32u
#define MPB_TERM_ALL 63u
int mpb_cost_terms_eval(const float *trajs, float *out, double *jl_total, const float *start_state,
                        const float *goal_states, const float *q_min, const float *q_max,
                        int B, int H, int d, int n_dof, int trajs_per_goal, uint32_t flags, float dt,
                        float k_gp, float k_start, float k_goal, float k_smooth, float k_jlim, float jl_eps,
                        int accumulate, int broadcast_jlim, void *stream);
/* Analytic gradient of the same terms -- what the reference obtains by autograd through CostComposite.eval when CHOMP
 * differentiates costs.sum() (chomp.py:135-139) -- and, with apply != 0, CHOMP's update in the same pass (:141-147):
 *   g = grad_in (may be NULL; e.g. mpb_cost_collision_grad's output) + d/dx [enabled terms, k_* as in
 *       mpb_cost_terms_eval; the joint-limit term times jl_scale: its batch-global scalar sits in EVERY trajectory's
 *       cost, so the gradient of costs.sum() carries the global batch size] + prior_bw * (R + R^T) x (CHOMP's
 *       smoothness prior, prior_bw = B_global * weight_prior_cost: quirk Q3; R (H,H), only its tridiagonal band is read)
 *   apply == 0: grad_out (B,H,d) = g;  apply != 0: trajs -= lr * mask(clamp(g, -grad_clip, grad_clip)) in place,
 *   the mask zeroing rows 0 and H-1.  grad_in / grad_out may alias. */
int mpb_cost_terms_grad(float *trajs, const float *grad_in, float *grad_out, const float *R,
                        const float *start_state, const float *goal_states, const float *q_min, const float *q_max,
                        int B, int H, int d, int n_dof, int trajs_per_goal, uint32_t flags, float dt,
                        float k_gp, float k_start, float k_goal, float k_smooth, float k_jlim, float jl_eps,
                        float jl_scale, float prior_bw, float lr, float grad_clip, int apply, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Trajectory utilities either side of the loop (all external to the reference: torch_robotics; build-defined).
 *   mpb_traj_interpolate       -- interpolate_points_v1 (call site cost_functions.py:118): n_interp evenly
 *                                 spaced points inserted between consecutive waypoints, linear in joint
 *                                 space: trajs (B,H,d) -> out (B,(H-1)*(n_interp+1)+1,d).
 *   mpb_traj_finite_difference -- finite_difference_vector(method='central') + the concatenation of
 *                                 OptimizationPlanner._get_traj (base.py:204-213): pos (B,H,D) ->
 *                                 out (B,H,2D) = [pos, (pos_{t+1}-pos_{t-1})/(2 dt)], zero end velocities.
 *   mpb_traj_resample          -- the warm start of HybridPlanner.optimize (hybrid_planner.py:42-66, over the
 *                                 external smoothen_trajectory / tensor_linspace_v1): N polylines of
 *                                 lengths[n] <= Lmax waypoints (paths (N,Lmax,D), rows beyond the length
 *                                 ignored) -> out (N,H,2D): H points uniform in arc length, linear in
 *                                 between, velocity = (last - first)/((H-1) dt) on interior points, 0 at ends.
 * ------------------------------------------------------------------------------------------- */
int mpb_traj_resample(const float *paths, const int *lengths, float *out, int N, int Lmax, int H, int D, float dt,
                      void *stream);
int mpb_traj_interpolate(const float *trajs, float *out, int B, int H, int d, int n_interp, void *stream);
/* GPFactor.get_error (costs/factors/gp_factor.py:52-56): err[b,t] = x[b,t+1] - Phi x[b,t]; x (B,H,2D) -> out (B,H-1,2D) */
int mpb_gp_factor_error(const float *x, float *out, int B, int H, int D, float dt, void *stream);
int mpb_traj_finite_difference(const float *pos, float *out, int B, int H, int D, float dt, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The robot / field API the reference's cost layer consumes (provider in the reference: torch_robotics), as separate
 * device ops with vector-Jacobian products, so that the UNMODIFIED reference cost classes can run on this package's
 * robot / field objects and torch.autograd can differentiate through them (field_factor.py:54):
 *   mpb_fk_collision_points      robot.fk_map_collision (call site cost_functions.py:52): q (B,H,d) -> pts (B,H,L,3)
 *   mpb_fk_collision_points_vjp  grad_q (B,H,n_dof) = J^T grad_pts
 *   mpb_field_cost_points        field.compute_cost (call sites field_factor.py:39,52): pts (B,H,L,3) -> cost (B,H)
 *                                = sum_l relu(margin + r_l - min_o sdf_o(pts_l))   (first field of `geom`)
 *   mpb_field_cost_points_vjp    grad_pts (B,H,L,3) = grad_cost (B,H) * d cost / d pts
 * The planners never call these (they use the fused evaluators); L = number of collision spheres of the robot.
 * ------------------------------------------------------------------------------------------- */
int mpb_fk_collision_points(const float *q, const float *geom, float *pts, int B, int H, int d, void *stream);
int mpb_fk_collision_points_vjp(const float *q, const float *geom, const float *grad_pts, float *grad_q,
                                int B, int H, int d, void *stream);
int mpb_field_cost_points(const float *pts, const float *geom, float *cost, int B, int H, void *stream);
int mpb_field_cost_points_vjp(const float *pts, const float *geom, const float *grad_cost, float *grad_pts,
                              int B, int H, void *stream);

/* ---------------------------------------------------------------------------------------------
 * STOMP -- replaces STOMP._run_optimization's loop body (stomp.py:150-160):
 *   sample (stomp.py:97-108 + MultivariateNormal.rsample), _get_costs (base.py:218-223) with the
 *   collision cost above, _calc_sample_weights (stomp.py:219-220), _update_distribution (:199-211).
 *
 * means (P,H,d) in/out, updated in place n_iters times.
 * eps: NULL -> standard normals are generated on the device (Philox4x32, 7 rounds, keyed by `seed`,
 *      counter = (particle_offset + p, s, (channel, waypoint group, call), iter0 + i); Box-Muller on 23-bit uniforms, a
 *      drawn normal carries 16 significant bits -- csrc/mpb_stomp_noise.h; result independent of sharding and of the
 *      kernel that serves the call);
 *      else (n_iters, S, d, P, H): pre-drawn standard normals in the reference's draw order
 *      (one MultivariateNormal.sample((S,d)) of batch shape (P,) and event shape (H,) per iteration).
 * samples (P,S,H,d), costs (P,S), weights (P,S): outputs of the LAST iteration (all required).
 * Alignment: means, eps, samples, L, Sigma, geom (and the workspace / means_copy of mpb_stomp_run*) must be 16-byte aligned --
 *      the kernels move them as 16-byte vectors; a pointer that is not is refused with MPB_E_INVALID (allocations of hipMalloc /
 *      PyTorch-ROCm are 256-byte aligned; a VIEW at an odd element offset is what trips this).
 * L (H,H): scale_tril of the noise distribution, Sigma (H,H) = inverse(R): constants the host
 *      computes exactly as the reference does (stomp.py:63-64, :88-95) -- SURVEY.md H2.
 * mpb_stomp_step runs the fused path (sample+cost kernel, update kernel per iteration).  Everything is
 * enqueued asynchronously; only for n_iters > 256 (and never under stream capture) the call blocks on its
 * own events so that at most 256 iterations are queued ahead of the GPU.
 * mpb_stomp_update accepts Sigma == NULL (update without the covariance product, used by StochGPMP).
 * mpb_stomp_sample / mpb_stomp_update expose the two halves (the two kernels of one iteration) so that
 * a caller-supplied cost callable (any Python cost on device tensors) can sit between them; with
 * geom + costs given, mpb_stomp_sample is exactly the first kernel of mpb_stomp_step.
 * ------------------------------------------------------------------------------------------- */
int mpb_stomp_step(float *means, const float *eps, float *samples, float *costs, float *weights,
                   const float *L, const float *Sigma, const float *geom, int geom_flags,
                   int P, int S, int H, int d, int D,
                   float k_sigma, float weight, float lr, float temperature,
                   int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset, void *stream);
/* Measurement aid for bench.py: the n_iters (1..1024) iterations of mpb_stomp_step with device noise, every kernel
 * launch carrying its own pair of HIP events on the dispatch (hipExtLaunchKernelGGL), then ONE stream synchronise.
 * Writes the average duration in ms of the sample+cost kernel and of the update kernel, measured in the loop they
 * run in (the per-kernel figure rocprofv3 --kernel-trace --stats reports for the same loop).  Host pointers. */
int mpb_stomp_step_profile(float *means, float *samples, float *costs, float *weights,
                           const float *L, const float *Sigma, const float *geom, int geom_flags,
                           int P, int S, int H, int d, int D,
                           float k_sigma, float weight, float lr, float temperature,
                           int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset, void *stream,
                           float *sample_kernel_ms, float *update_kernel_ms);
#define MPB_STOMP_WS_HEADER_BYTES 1280
/* The same loop as ONE persistent launch (csrc/mpb_stomp_fused.hip): a workgroup of 16 waves owns (particle, chunk of 16
 * samples) for all n_iters iterations -- constants, means and the iteration's samples stay in LDS, the partners of a
 * particle (S > 16) exchange their 3.6 KB partial sums through `workspace`; no per-iteration launch ramps or dispatch
 * gaps.  Same arguments and outputs as mpb_stomp_step plus the caller-allocated workspace (mpb_stomp_workspace_bytes;
 * its header, the first MPB_STOMP_WS_HEADER_BYTES, must be ZERO before the first call -- mpb_stomp_workspace_init -- and is
 * maintained by the library afterwards (ABI 4: 1 280 bytes, the counters the workgroups hit at start-up one cache line each; 64 before); the rest need not be initialised; one workspace serves one call at a time).  Served for H = 64, S <= 64,
 * grid-backed fields (geom_flags bit 8); any other call -- or workspace == NULL -- runs mpb_stomp_step
 * (mpb_stomp_run_path tells which, without launching).  The softmax is evaluated as exp(x - m) / z over per-chunk
 * partials, the same weights up to rounding.  When the particles are at least as many as the CUs (and 16 < S <= 32) one
 * workgroup per particle runs the samples as two batches of 16 instead -- no exchange, same bits (environment
 * MPB_STOMP_BATCHES = 1 / 2 forces a layout; a test aid).
 *
 * Failure contract.  The workgroups of a particle wait for each other's partial sums every iteration.  They are paired
 * by a ticket drawn when a workgroup STARTS (not by block index), so partners are always workgroups that started next to
 * each other and no dispatch order or co-residency is assumed; every wait is nevertheless bounded (2 s + 100 us per
 * iteration; MPB_STOMP_TIMEOUT_US overrides, a test aid).  A workgroup whose partner does not arrive in time -- the
 * device stopped starting this grid's workgroups for that long, e.g. another context holds every CU -- marks the call
 * LOST: every workgroup leaves, the means of the affected particles are not written, samples / costs / weights are
 * undefined.  The call still returns MPB_OK (it is asynchronous); the loss is reported
 *   - by mpb_stomp_run_status (reads the workspace header; synchronises the stream): 0 fine, 1 lost, 2 header not zeroed;
 *   - without synchronising, by mpb_stomp_run_checked's `status`: 8 words of pinned, device-mapped HOST memory
 *     (hipHostMalloc), zero before the first call: [0] = tag of the last call that has completed, [1] = tag of the last
 *     call that was lost (0: none), [2] = why (1 partner timed out, 2 header not zeroed), [4..5] / [6..7] = the device's
 *     100 MHz real-time counter when the launch's first unit started / when its last workgroup left (the launch's
 *     duration as the device saw it: a measurement aid that costs nothing); `tag_out` receives the tag
 *     of this call (0 when it ran the two-kernel loop, which cannot be lost).  The caller compares [1] with the tags it
 *     has issued whenever convenient -- planners/stomp.py raises at the next planner call.  Word [0] says the tag was
 *     PUBLISHED -- every workgroup's result stores are ordered before it (each wave drains its stores, a block barrier,
 *     then the release) -- not that the kernel has retired: a consumer on ANOTHER stream must still order itself behind
 *     the launch stream (event / stream wait); work on the launch stream and host reads after a stream / device
 *     synchronise are ordered as always.
 * means_copy (P,H,d) or NULL: a second destination for the final means, written by the same launch (the reference's
 * optimize() returns a CLONE of its means, base.py:204-213: this saves the dependent copy kernel).
 * Not capturable in a HIP graph (the per-call tag is drawn on the host). */
#define MPB_STOMP_PATH_TWO_KERNEL 0
#define MPB_STOMP_PATH_PERSISTENT_EXCHANGE 1   /* one workgroup per (particle, chunk of 16 samples), partials exchanged */
#define MPB_STOMP_PATH_PERSISTENT 2            /* one workgroup per particle (S <= 16, or two batches of 16), no exchange */
size_t mpb_stomp_workspace_bytes(int P, int S, int H, int d);
int mpb_stomp_workspace_init(float *workspace, size_t workspace_bytes, void *stream);
int mpb_stomp_run_path(int geom_flags, size_t workspace_bytes, int P, int S, int H, int d);
int mpb_stomp_run(float *means, const float *eps, float *samples, float *costs, float *weights,
                  const float *L, const float *Sigma, const float *geom, int geom_flags,
                  float *workspace, size_t workspace_bytes,
                  int P, int S, int H, int d, int D,
                  float k_sigma, float weight, float lr, float temperature,
                  int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset, void *stream);
int mpb_stomp_run_checked(float *means, const float *eps, float *samples, float *costs, float *weights,
                          const float *L, const float *Sigma, const float *geom, int geom_flags,
                          float *workspace, size_t workspace_bytes,
                          int P, int S, int H, int d, int D,
                          float k_sigma, float weight, float lr, float temperature,
                          int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset,
                          uint32_t *status, uint32_t *tag_out, float *means_copy, void *stream);
int mpb_stomp_run_status(const float *workspace, void *stream, int *timed_out);
/* Measurement aid (bench.py): mpb_stomp_run_checked with the persistent kernel's begin / end timestamps recorded on the
 * dispatch itself (what rocprofv3 --kernel-trace reports); synchronises the stream; *kernel_ms = the kernel's duration in
 * milliseconds (0 when the call ran the two-kernel loop). */
int mpb_stomp_run_timed(float *means, const float *eps, float *samples, float *costs, float *weights,
                        const float *L, const float *Sigma, const float *geom, int geom_flags,
                        float *workspace, size_t workspace_bytes,
                        int P, int S, int H, int d, int D,
                        float k_sigma, float weight, float lr, float temperature,
                        int n_iters, uint64_t seed, uint32_t iter0, uint32_t particle_offset,
                        uint32_t *status, uint32_t *tag_out, float *means_copy, void *stream, float *kernel_ms);
/* A persistent-launch call with everything but (n_iters, iter0, means_copy, stream) fixed, kept on the library's side (round 4):
 * for a binding whose foreign-function marshalling is not free (ctypes: ~3 us for the 28 arguments of mpb_stomp_run_checked,
 * a quarter of the host side of a call).  mpb_stomp_plan_create validates and copies the arguments (host memory only: one small
 * malloc, the one place the library allocates; device noise only, eps = NULL); mpb_stomp_plan_launch == mpb_stomp_run_checked on
 * them (same failure contract, same tag); mpb_stomp_plan_destroy frees the record.  The caller keeps every buffer alive. */
typedef struct mpb_stomp_plan_s mpb_stomp_plan;
int mpb_stomp_plan_create(mpb_stomp_plan **plan, float *means, float *samples, float *costs, float *weights,
                          const float *L, const float *Sigma, const float *geom, int geom_flags,
                          float *workspace, size_t workspace_bytes, int P, int S, int H, int d, int D,
                          float k_sigma, float weight, float lr, float temperature, uint64_t seed, uint32_t particle_offset,
                          uint32_t *status);
int mpb_stomp_plan_launch(mpb_stomp_plan *plan, int n_iters, uint32_t iter0, float *means_copy, void *stream, uint32_t *tag_out);
int mpb_stomp_plan_destroy(mpb_stomp_plan *plan);

int mpb_stomp_sample(const float *means, const float *eps, float *samples, const float *L,
                     const float *geom, int geom_flags, float *costs, /* geom, costs both NULL: sample only; both set: fused cost */
                     int P, int S, int H, int d, float k_sigma, float weight,
                     uint64_t seed, uint32_t iter, uint32_t particle_offset, void *stream);
int mpb_stomp_update(float *means, const float *samples, const float *costs, float *weights,
                     const float *Sigma, int P, int S, int H, int d, float lr, float temperature,
                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * CHOMP -- replaces CHOMP._run_optimization's loop body (chomp.py:127-151) and _eval (:153-169):
 *   g = d/dx [ collision cost + w_prior * B_global * sum_c x_c^T R x_c ]   (quirk Q3: the batch-total
 *   smoothness scalar is added to every particle's cost and costs.sum() is differentiated, so the
 *   smoothness gradient carries the GLOBAL batch size), clamp to +-grad_clip, zero the two endpoint
 *   rows, x -= lr * g;  repeated n_iters times inside one launch.
 * means (B_local,H,d) in/out.  R (H,H): CHOMP._get_R_mat (chomp.py:81-101), computed by the host;
 * only its tridiagonal band is read.  costs_out (B_local) optional: collision cost (scaled) of the
 * iterate BEFORE the last update, without the smoothness scalar.
 * ------------------------------------------------------------------------------------------- */
int mpb_chomp_step(float *means, const float *R, const float *geom, int geom_flags, float *costs_out,
                   int B_local, int B_global, int H, int d, int D,
                   float k_sigma, float weight, float w_prior, float lr, float grad_clip,
                   int n_iters, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GPMP2 -- replaces GPMP2._step (gpmp2.py:308-342): CostComposite.get_linear_system
 * (cost_functions.py:107-144 over CostGP :291-314, CostGoalPrior :538-554, CostCollision :191-231),
 * _get_grad_terms dense branch (gpmp2.py:355-368), get_torch_solve('cholesky') (:451-452) and the
 * update (:339), WITHOUT materialising the dense (A,b,K): the normal equations A^T K A are
 * block-tridiagonal with 2D x 2D blocks and are assembled and solved per particle by block Cholesky.
 * Internal arithmetic is fp64 (weights reach 1/sigma^2 = 1e10); x is stored in fp32.
 *
 * x (B,H,2D) in/out; start (B,2D), goal (B,2D) per-particle states (zero velocities appended by the
 * host; the reference's single start / goal is broadcast by the caller).
 * workspace: caller-allocated device scratch of mpb_gpmp2_workspace_bytes(B,H,D) bytes, 256-byte
 * aligned (collision Jacobians, the diagonal sums and the per-step elimination factors).
 *
 * One iteration = linearize -> [diag] -> solve:
 *   mpb_gpmp2_linearize : collision cost c_t and Jacobian h_t = -dc_t/dq of every waypoint (FieldFactor
 *                         get_error(calc_jacobian=True), field_factor.py:41-57) into the workspace.
 *                         n_interp > 0 (CostComposite.get_linear_system(n_interpolated_points),
 *                         cost_functions.py:115-119): h_t = -d/dq_t of the summed cost of the trajectory
 *                         with n_interp points inserted per segment (see mpb_traj_interpolate); c_t unchanged;
 *   mpb_gpmp2_diag      : LOCAL SUM over the B particles of diag(A^T K A) (H*2D fp64) -- quirk Q9: the
 *                         trust-region damping uses the BATCH MEAN of that diagonal (gpmp2.py:361-367).
 *                         diag_sum_out NULL: kept in the workspace.  When sharded, the host all-reduces
 *                         the sums and passes diag_mean = sum / B_global to mpb_gpmp2_solve;
 *   mpb_gpmp2_solve     : assemble + solve + x += step_size * dtheta.  Two forms of the same fp64 solve: the LOW-RANK form
 *                         (round 6, csrc/mpb_gpmp2_lr.hip: priors + GP blocks + damping are shared by all particles and decouple
 *                         over the joints -- factored once per call --, the collision factors enter through a dense SPD system
 *                         over each particle's ACTIVE rows; as accurate as dense fp64 Cholesky at every precision ratio) wherever
 *                         n_fields * (H - 1) <= 127, and the block-tridiagonal elimination (csrc/mpb_gpmp2.hip) otherwise (D <= 8
 *                         only: a 2D x 2D block must fit its 16 x 16 tile).  MPB_GPMP2_FORM = lr / block (environment, read per
 *                         call) forces one; MPB_GPMP2_SM set means the block form.
 *                         trust_region == 0: damping delta * I (diag_mean ignored);
 *                         else delta * diag_mean (diag_mean NULL: the LOCAL mean mpb_gpmp2_diag leaves in the workspace
 *                         when it is called without diag_sum_out, as mpb_gpmp2_step does).
 *                         costs_out (B) optional: b^T K b of the iterate BEFORE the update (gpmp2.py:493-495).
 *   mpb_gpmp2_step      : n_iters full iterations on one GPU (mean over the local B).
 * Accuracy: the elimination forms W_t = S_t^-1 explicitly (blocked Gauss-Jordan, fp64).  With the rank-1 collision term
 * ASSEMBLED into S_t that resolves its stiff direction to kappa^2 u where a Cholesky factorisation gives kappa u: step error
 * against the dense fp64 solution refined in long double 5e-8 at a ratio (sigma_gp / sigma_coll)^2 of the collision to the GP
 * precision of 1e6 (the reference's defaults), <= 2.3e-6 at 1e8, 8e-3 at 1e10 (ABI <= 4 as released in round 4).  Since round 5
 * mpb_gpmp2_solve applies the collision factors by SHERMAN-MORRISON on the inverse of the well-conditioned rest whenever
 * (sigma_gp / sigma_coll)^2 * n_fields > 1e7 (csrc/mpb_gpmp2.hip, template flag SM; +7.5 % on the solve): measured 3e-8 .. 7e-8 at
 * 1e8 and 1e10 (one / two fields, trust region on / off, interpolated Jacobian), 1.2e-5 at 1e12 where the dense fp64 Cholesky
 * is itself at ~1e-6 -- no range to keep to.  MPB_GPMP2_SM = 0 / 1 (environment) forces a form; the start / goal precisions
 * (1e10 .. 1e12 tested) do not matter.
 * Alignment: x and geom 16-byte aligned, workspace 256-byte aligned (rows are moved as 8- / 16-byte pieces); a pointer
 * that is not is refused with MPB_E_INVALID (allocations of hipMalloc / PyTorch-ROCm are 256-byte aligned; a VIEW with a
 * storage offset may not be).
 * sigma_goal <= 0 means "no goal factor" (precision 0: GPMP2 without goals, gpmp2.py:62-78) -- an infinite sigma is not a
 * valid argument (the library is built with -ffinite-math-only).
 * n_fields = number of collision fields chained in `geom` (1..4; see mpb_geom_check): the reference stacks one
 * block of H-1 collision rows per field (gpmp2.py:70-78, cost_functions.py:107-144), the workspace keeps one
 * (h_t, c_t) set per field and the solve sums their rank-1 terms.
 * ------------------------------------------------------------------------------------------- */
size_t mpb_gpmp2_workspace_bytes(int B, int H, int D);
int mpb_gpmp2_linearize(const float *x, const float *geom, int geom_flags, void *workspace, int B, int H, int D,
                        int n_interp, void *stream);
int mpb_gpmp2_diag(void *workspace, double *diag_sum_out, int B, int H, int D, int n_fields, float dt,
                   float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll, void *stream);
int mpb_gpmp2_solve(float *x, const float *start, const float *goal, const double *diag_mean, void *workspace,
                    float *costs_out, int B, int H, int D, int n_fields, float dt,
                    float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                    float delta, int trust_region, float step_size, void *stream);
int mpb_gpmp2_step(float *x, const float *start, const float *goal, const float *geom, int geom_flags, void *workspace,
                   float *costs_out, int B, int H, int D, float dt,
                   float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                   float delta, int trust_region, float step_size, int n_iters, int n_interp, int n_fields,
                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * MPPI -- replaces MPPI.optimize's loop body (mppi.py:145-152): ControlTrajectoryGaussian.sample
 * (priors/gaussian.py:276-298), get_state_trajectories_rollout (mppi.py:190-210) over
 * PointParticleDynamics.dynamics (dynamics/point.py:102-140, deterministic), traj_cost (:154-226),
 * the importance-sampling term (mppi.py:125-128) and update_controller (:72-86).
 * One problem per workgroup; NP independent problems per launch (the reference class is NP = 1);
 * all n_iters iterations inside one launch.
 *
 * mean (NP,T,c) in/out; eps NULL -> standard normals generated on the device (since ABI 4: Philox4x32 with 7 rounds keyed by
 * `seed`, counter = (problem, sample, (t / 4) | dim << 16, iter0 + i), one call per four consecutive time steps; Box-Muller
 * on the top 23 bits of each word, so |n| <= sqrt(2 ln 2^23) = 5.65; ABI 3 drew Philox4x32-10 with 24-bit uniforms: the
 * device-noise streams of STOMP and MPPI are NOT reproducible across ABI 3 -> 4; mpb_debug_mppi_normals of include/mpb_debug.h
 * returns the stream), else (n_iters,NP,c,S,T) standard normals (per
 * control dimension, in the reference's draw order); scale_tril, cov_inv (c,T,T); state0 (NP,c);
 * goal (NP,c); ctrl_min / ctrl_max (c); discount (T); c_weights = {pos, vel, ctrl, pos_T};
 * control_type 0 = velocity (state_dim = c); 1 = acceleration is MPB_E_UNSUPPORTED (the reference's
 * acceleration mode slices an empty tensor, point.py:114-118, and cannot run).
 * geom optional (NULL = no collision term); geom_flags = mpb_geom_flags(host copy of geom) (0 is always valid): with a
 * single grid-backed field the collision cost of a rollout goes through the broad-phase grid (staged in LDS: one
 * candidate look-up per waypoint instead of the exhaustive obstacle loop; same bits).  With geom, the reference's quirk Q6 is reproduced:
 * the per-sample collision costs are summed into ONE scalar that is added to every sample's cost.
 * Outputs of the last iteration: controls (NP,S,T,c), states (NP,S,T,c), costs (NP,S), weights (NP,S).
 * best_cost (NP) in/out + best_states (NP,T,c) out, both or neither (NULL): MPPI._save_best
 * (mppi.py:164-168, called every iteration, mppi.py:148): whenever an iteration's cheapest sample (first
 * index on ties) beats best_cost[problem], its cost and state trajectory are stored.  Initialise
 * best_cost to a large FINITE value (3e38; the library is built with -ffinite-math-only); it carries over between calls
 * like the reference's attribute.
 * How the work is laid out (T <= 64: mean + scale_tril @ eps of all samples on the matrix pipe; 8 or 16 waves per
 * problem depending on NP) does not show in the results: a problem's outputs are the same bits whatever NP is.
 * ------------------------------------------------------------------------------------------- */
int mpb_mppi_step(float *mean, const float *eps, const float *scale_tril, const float *cov_inv,
                  const float *state0, const float *goal, const float *ctrl_min, const float *ctrl_max,
                  const float *discount, const float *c_weights, const float *geom, int geom_flags,
                  float *controls, float *states, float *costs, float *weights,
                  float *best_cost, float *best_states,
                  int NP, int S, int T, int c, int control_type, float dt,
                  float k_sigma, float weight, float temp, float step_size,
                  int n_iters, uint64_t seed, uint32_t iter0, void *stream);
/* The point-particle system as stand-alone entry points, for code written against the reference's system object
 * (dynamics/point.py); mpb_mppi_step fuses the same arithmetic.
 * mpb_point_dynamics -- PointParticleDynamics.dynamics (point.py:102-140): x_next = x + (clamp(u, ctrl_min, ctrl_max) +
 *   dyn_std * noise) * dt over n rows of `dim` entries (the reference's xdot = cat(x[..., state_dim:], u) has an empty
 *   first part); noise NULL = deterministic, else (n, dim) standard normals (the reference draws them with torch.randn).
 * mpb_point_traj_cost -- PointParticleDynamics.traj_cost (point.py:154-226): X (T, B, state_dim), U (T, B, ctrl_dim) in the
 *   reference's time-major layout, goal (state_dim), discount (T) -> costs (B) = sum_t disc_t (w_pos |X - goal|^2 +
 *   w_ctrl |U|^2) + w_pos_T disc_{T-1} |X_{T-1} - goal|^2 + energy; w_vel is accepted and unused (quirk Q8: the reference
 *   slices dX[..., state_dim:control_dim], an empty tensor); `energy` is the scalar the caller's cost object contributes
 *   (quirk Q6: cost.eval(cat(X, U)).sum(-1) collapses to ONE number added to every rollout; 0 without one). */
int mpb_point_dynamics(const float *x, const float *u, const float *ctrl_min, const float *ctrl_max, const float *dyn_std,
                       const float *noise, float *x_next, size_t n, int dim, float dt, void *stream);
int mpb_point_traj_cost(const float *X, const float *U, const float *goal, const float *discount, float w_pos, float w_vel,
                        float w_ctrl, float w_pos_T, float energy, float *costs, int T, int B, int state_dim, int ctrl_dim,
                        void *stream);


/* ---------------------------------------------------------------------------------------------
 * StochGPMP -- replaces StochGPMP.sample_and_eval / _get_costs / _update_distribution
 * (stoch_gpmp.py:235-279).  One iteration is three calls:
 *   mpb_gp_prior_sample   samples (P*S,H,2D) around the particle means from the sampling prior (below);
 *   mpb_stoch_gpmp_costs  costs (P,S) = CostGP.eval + CostGoalPrior.eval + CostCollision.eval
 *                         (cost_functions.py:271-289, :520-536, :171-189) + T * V Sigma^-1 U^T
 *                         (stoch_gpmp.py:239-241), evaluated factor-wise without the dense Sigma^-1;
 *                         means (P,H,2D) particle means, start / goal (P,2D);
 *   mpb_stomp_update      with Sigma == NULL: softmax over S and means += lr * sum_s w_s (sample_s - mean)
 *                         (stoch_gpmp.py:267-275, no covariance in the update).
 * ------------------------------------------------------------------------------------------- */
int mpb_stoch_gpmp_costs(const float *samples, const float *means, const float *start, const float *goal,
                         const float *geom, float *costs, int P, int S, int H, int D, float dt,
                         float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                         float sigma_start_sample, float sigma_gp_sample, float sigma_goal_sample,
                         float temperature, void *stream);

/* The whole loop, n_iters iterations enqueued by one call (device Philox noise, iteration i draws with seed + i exactly
 * like n_iters single-iteration rounds of the three calls above): means (P,H,2D) in/out; means64 (P,H,2D) fp64 scratch;
 * samples (P*S,H,2D), costs (P,S), weights (P,S): outputs of the last iteration; Udiag / Uoff / scale_tril: the sampling
 * prior as for mpb_gp_prior_sample(_dense) (scale_tril NULL or H > 128: chain form). */
int mpb_stoch_gpmp_step(float *means, double *means64, float *samples, float *costs, float *weights,
                        const double *Udiag, const double *Uoff, const double *scale_tril,
                        const float *start, const float *goal, const float *geom,
                        int P, int S, int H, int D, float dt,
                        float sigma_start, float sigma_gp, float sigma_goal, float sigma_coll,
                        float sigma_start_sample, float sigma_gp_sample, float sigma_goal_sample,
                        float temperature, float step_size, int n_iters, uint64_t seed, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GP-prior initial particles -- replaces OptimizationPlanner.get_random_trajs (base.py:155-202) over
 * MultiMPPrior (costs/factors/mp_priors_multi.py:100-110, :213-256): x = mean + scale_tril @ eps with
 * scale_tril = U^-T, K^-1 = U U^T block upper-bidiagonal with (2x2) (x) I_D blocks (fp64 throughout).
 * out (G*n, H, 2D) fp32, particle index = mode * n + sample (base.py:202); means (G,H,2D) fp64;
 * eps NULL -> device Philox, else (n, G, H*2D) fp64 standard normals (MultivariateNormal draw order);
 * Udiag (H,3) = (u00,u01,u11) of U_tt, Uoff (H-1,4) = row-major 2x2 U_{t,t+1}; both fp64, device.
 * ------------------------------------------------------------------------------------------- */
int mpb_gp_prior_sample(float *out, const double *means, const double *eps, const double *Udiag,
                        const double *Uoff, int G, int n, int H, int D, uint64_t seed, void *stream);
/* Same samples from the dense per-dof scale_tril (2H x 2H row-major fp64, index 2t + {0: position, 1: velocity};
 * = U_dof^-T, what MultivariateNormal(precision_matrix=...) holds for one degree of freedom) as a GEMM on the
 * matrix cores (v_mfma_f64_16x16x4_f64); H <= 128.  Same Philox stream as mpb_gp_prior_sample. */
int mpb_gp_prior_sample_dense(float *out, const double *means, const double *eps, const double *scale_tril,
                              int G, int n, int H, int D, uint64_t seed, void *stream);
/* MultiMPPrior with ARBITRARY start / GP / goal precisions (mp_priors_multi.py:213-256 accepts any matrices; the planners
 * of the reference pass isotropic ones, which the two entries above serve): x = mean + L eps from the dense M x M
 * scale_tril L of the whole trajectory (M = 2D*H <= 4096), handed over TRANSPOSED (tril_t[k*M + m] = L[m][k], fp64).
 * means (G,M) fp64; eps NULL (device Philox) or (n,G,M) fp64; out (G*n, M) fp32, index mode * n + sample. */
int mpb_mvn_sample_dense(float *out, const double *means, const double *eps, const double *tril_t,
                         int G, int n, int M, uint64_t seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPB_H */
