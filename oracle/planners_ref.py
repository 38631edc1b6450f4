"""ORACLE (test infrastructure: checker only; never imported by the product path, never the thing
measured as the product; bench.py times it only as ``cpu_baseline`` kind "port").

CPU / PyTorch restatement of the reference's trajectory-optimisation inner loops
(anindex/motion_planning_baselines, mp_baselines/planners).  Every function cites the reference
file:line it follows.  The arithmetic of the planner loop (noise construction, softmax weighting,
covariance-weighted update, CHOMP gradient pipeline, GP factors, (A,b,K) assembly, normal equations,
Cholesky solve, MPPI rollout) is PINNED against golden vectors produced by running the unmodified
reference classes in the build container (tests/golden/make_goldens.py -> tests/golden/*.npz,
checked by tests/test_oracle_vs_golden.py).  The geometry underneath (FK, SDF, hinge) is
build-defined: PARITY UNPINNED vs torch_robotics (see oracle/geometry_ref.py).

Functional style on purpose: state in, state out; no planner objects.
"""
import math

import torch


# ------------------------------------------------------------------------------------------------
# shared cost pieces
# ------------------------------------------------------------------------------------------------

def collision_cost(trajs, robot, field, sigma_coll, weight=1.0):
    """(B,H,d) -> (B,)   weight * 1/sigma^2 * sum_{h>=1} field_cost(q_h).

    Follows CostComposite.eval (cost_functions.py:70-87: weight_cost * cost(...)),
    get_q_pos_vel_and_fk_map (:41-53), CostCollision.eval (:171-189: ``w_mat * err_obst.sum(1)``) and
    FieldFactor.get_error with traj_range [1, None] (field_factor.py:31-39; K = 1/sigma^2, :15).
    """
    q_pos = robot.get_position(trajs)
    link_pos = robot.fk_map_collision(q_pos)
    err = field.compute_cost(q_pos[:, 1:], link_pos[:, 1:]).reshape(trajs.shape[0], -1)
    K = 1.0 / (sigma_coll ** 2)
    return weight * (K * err.sum(1))


# ------------------------------------------------------------------------------------------------
# STOMP  (mp_baselines/planners/stomp.py)
# ------------------------------------------------------------------------------------------------

def stomp_precision(H, dt, sigma_spectral, tensor_args):
    """R = A^T A, A = (H+2)xH second-difference with unit corners, scaled sigma/dt^2 (stomp.py:68-86)."""
    A = torch.zeros(H + 2, H)
    idx = torch.arange(H)
    A[idx + 1, idx] = -2.0
    A[idx[:-1] + 1, idx[:-1] + 1] = 1.0   # upper diagonal of the inner HxH block
    A[idx[1:] + 1, idx[1:] - 1] = 1.0     # lower diagonal
    A[0, 0] = 1.0
    A[-1, -1] = 1.0
    A = A * 1. / dt ** 2 * sigma_spectral
    return (A.t() @ A).to(**tensor_args)


def precision_to_scale_tril(P):
    """torch/distributions/multivariate_normal.py:80-86 (flip-Cholesky, triangular solve)."""
    Lf = torch.linalg.cholesky(torch.flip(P, (-2, -1)))
    L_inv = torch.transpose(torch.flip(Lf, (-2, -1)), -2, -1)
    Id = torch.eye(P.shape[-1], dtype=P.dtype, device=P.device)
    return torch.linalg.solve_triangular(L_inv, Id, upper=False)


def stomp_constants(H, dt, sigma_spectral, tensor_args):
    """(R, Sigma, L): stomp.py:63-64 (Sigma = inverse(R)); :88-95 (MVN(precision_matrix=R) -> scale_tril)."""
    R = stomp_precision(H, dt, sigma_spectral, tensor_args)
    return R, torch.inverse(R), precision_to_scale_tril(R)


def stomp_sample(means, L, eps):
    """eps (S,d,P,H) standard normal in the reference's draw order -> samples (P,S,H,d).

    stomp.py:97-108 + MultivariateNormal.rsample (multivariate_normal.py:250-253: loc + L @ eps, loc = 0).
    Noise rows t=0 and t=H-1 are zeroed (stomp.py:105-106), then the particle mean is added.
    """
    noise = torch.matmul(L, eps.unsqueeze(-1)).squeeze(-1)               # (S,d,P,H)
    noise = noise.transpose(0, 2).transpose(1, 3).transpose(1, 2)        # (P,S,H,d)
    noise = noise.clone()
    noise[..., -1, :] = 0
    noise[..., 0, :] = 0
    return means.unsqueeze(1) + noise


def stomp_weights(costs, temperature):
    """softmax(-c/T) over the sample axis (stomp.py:219-220)."""
    return torch.softmax(-costs / temperature, dim=1)


def stomp_update(means, samples, weights, Sigma, lr):
    """means += lr * Sigma @ sum_s w_s (sample_s - mean)   (stomp.py:199-211). Returns new means."""
    P, S = weights.shape
    w = weights.reshape(P, S, 1, 1)
    delta = (w * (samples - means.unsqueeze(1))).sum(1)
    return means + lr * Sigma @ delta


def stomp_iteration(means, eps, L, Sigma, cost_fn, lr, temperature):
    """One pass of STOMP._run_optimization's loop body (stomp.py:157-160).

    Returns dict(samples (P,S,H,d), costs (P,S), weights (P,S), means (P,H,d)).
    """
    P = means.shape[0]
    samples = stomp_sample(means, L, eps)
    S = samples.shape[1]
    costs = cost_fn(samples.flatten(0, 1)).reshape(P, S)   # stomp.py:183-184
    weights = stomp_weights(costs, temperature)
    new_means = stomp_update(means, samples, weights, Sigma, lr)
    return dict(samples=samples, costs=costs, weights=weights, means=new_means)


# ------------------------------------------------------------------------------------------------
# CHOMP  (mp_baselines/planners/chomp.py)
# ------------------------------------------------------------------------------------------------

def chomp_precision(H, dt, tensor_args):
    """R = K^T K, K = (H+1)xH backward difference with last row -1, scaled 1/dt^2 (chomp.py:81-101)."""
    K = torch.zeros(H + 1, H)
    idx = torch.arange(H)
    K[idx, idx] = 1.0
    K[idx[1:], idx[1:] - 1] = -1.0
    K[-1, -1] = -1.0
    K = K * 1. / dt ** 2
    return (K.t() @ K).to(**tensor_args)


def smoothness_sum(x, R):
    """sum over batch and channels of x_c^T R x_c (chomp.py:165 with the build-defined
    batched_weighted_dot_prod, oracle/ref_stub.py)."""
    r = x.transpose(-2, -1) @ R.unsqueeze(0) @ x
    return r.diagonal(dim1=-2, dim2=-1).sum()


def chomp_iteration(means, R, cost_fn, w_prior, lr, grad_clip):
    """One pass of CHOMP._run_optimization's loop body (chomp.py:134-149) incl. quirk Q3:
    the batch-total smoothness scalar is added to every particle's cost (chomp.py:165-167) and
    ``costs.sum()`` is back-propagated (:139), so the smoothness gradient carries a factor B.
    Returns dict(costs (B,), grad (B,H,d) after clamp+mask, means (B,H,d)).
    """
    x = means.detach().clone().requires_grad_(True)
    costs = cost_fn(x)
    costs = costs + w_prior * smoothness_sum(x, R)
    costs.sum().backward()
    g = x.grad.detach().clone()
    g.clamp_(-grad_clip, grad_clip)          # chomp.py:141
    g[..., 0, :] = 0.                        # :143-144
    g[..., -1, :] = 0.
    new_means = means.detach() + (-lr * g)   # :147
    return dict(costs=costs.detach(), grad=g, means=new_means)


# ------------------------------------------------------------------------------------------------
# GP factors, GPMP2  (costs/factors/*.py, costs/cost_functions.py, gpmp2.py)
# ------------------------------------------------------------------------------------------------

def gp_phi(D, dt, tensor_args):
    """Phi = [[I, dt I],[0, I]] (gp_factor.py:34-40)."""
    I = torch.eye(D, **tensor_args)
    Z = torch.zeros(D, D, **tensor_args)
    return torch.cat((torch.cat((I, dt * I), 1), torch.cat((Z, I), 1)), 0)


def gp_Q_inv(D, dt, sigma_gp, tensor_args):
    """Q^-1 = [[12/dt^3, -6/dt^2],[-6/dt^2, 4/dt]] (x) I/sigma^2 (gp_factor.py:42-50, :24)."""
    Qc = torch.eye(D, **tensor_args) / sigma_gp ** 2
    m1 = 12. * (dt ** -3.) * Qc
    m2 = -6. * (dt ** -2.) * Qc
    m3 = 4. * (dt ** -1.) * Qc
    return torch.cat((torch.cat((m1, m2), -1), torch.cat((m2, m3), -1)), -2)


def gp_error(x, Phi):
    """err_t = x_{t+1} - Phi x_t, (B,H-1,2D) (gp_factor.py:52-56)."""
    return x[:, 1:] - (Phi @ x[:, :-1].unsqueeze(-1)).squeeze(-1)


def cost_gp_eval(x, start_state, D, dt, sigma_start, sigma_gp, tensor_args):
    """CostGP.eval (cost_functions.py:271-289): start unary cost + sum_t err_t^T Q^-1 err_t."""
    dim = 2 * D
    err_p = (start_state - x[:, 0]).unsqueeze(1)                                   # unary_factor.py:24
    Ks = torch.eye(dim, **tensor_args) / sigma_start ** 2
    start_costs = (err_p @ Ks.unsqueeze(0) @ err_p.transpose(1, 2)).reshape(-1)
    e = gp_error(x, gp_phi(D, dt, tensor_args)).unsqueeze(-1)                      # (B,H-1,2D,1)
    Qi = gp_Q_inv(D, dt, sigma_gp, tensor_args).reshape(1, 1, dim, dim)
    gp_costs = (e.transpose(2, 3) @ Qi @ e).sum(1).reshape(-1)
    return start_costs + gp_costs


def cost_gp_trajectory_eval(x, D, dt, sigma_gp, tensor_args):
    """CostGPTrajectory.eval (cost_functions.py:344-354): sum_t err_t^T Q^-1 err_t, no start term."""
    dim = 2 * D
    e = gp_error(x, gp_phi(D, dt, tensor_args)).unsqueeze(-1)
    Qi = gp_Q_inv(D, dt, sigma_gp, tensor_args).reshape(1, 1, dim, dim)
    return (e.transpose(2, 3) @ Qi @ e).sum(1).reshape(-1)


def finite_difference_central(x, dt):
    """Build-defined stand-in for torch_robotics' finite_difference_vector(method='central') (external;
    call sites base.py:211, cost_functions.py:366): interior (x_{t+1}-x_{t-1})/(2 dt), zero end rows."""
    out = torch.zeros_like(x)
    out[..., 1:-1, :] = (x[..., 2:, :] - x[..., :-2, :]) / (2 * dt)
    return out


def cost_gp_trajectory_pos_only_eval(x_pos, D, dt, sigma_gp, tensor_args):
    """CostGPTrajectoryPositionOnlyWrapper.eval (cost_functions.py:365-368)."""
    x = torch.cat((x_pos, finite_difference_central(x_pos, dt)), dim=-1)
    return cost_gp_trajectory_eval(x, D, dt, sigma_gp, tensor_args)


def cost_smoothness_chomp_eval(x, dt, tensor_args):
    """CostSmoothnessCHOMP.eval (cost_functions.py:384-387) with R from CHOMP._get_R_mat (chomp.py:81-101):
    per (trajectory, state column) x^T R x.  The reference returns whatever the external
    batched_weighted_dot_prod returns; the build defines the cost as the sum over columns -> (B,)."""
    R = chomp_precision(x.shape[1], dt, tensor_args)
    per_col = (x.transpose(-2, -1) @ R.unsqueeze(0) @ x).diagonal(dim1=-2, dim2=-1)      # (B, d)
    return per_col.sum(-1), per_col


def cost_joint_limits_eval(x, D, q_min, q_max, eps):
    """CostJointLimits.eval (cost_functions.py:406-426).  `.sum(-1)` of the gathered 1-D violation vector
    is a scalar: the cost is ONE number for the whole batch (quirk, like Q6), broadcast by the composite."""
    pos = x[..., :D]
    lo = torch.clamp(q_min + eps - pos, min=0.0)
    hi = torch.clamp(pos - (q_max - eps), min=0.0)
    return (lo ** 2).sum() + (hi ** 2).sum()


def cost_goal_prior_multi_eval(x, goal_states, trajs_per_goal, sigma_goal):
    """CostGoalPrior.eval (cost_functions.py:523-536): trajectory b belongs to goal b // trajs_per_goal."""
    B = x.shape[0]
    g = goal_states[torch.arange(B) // trajs_per_goal]
    err = g - x[:, -1]
    return (err * err).sum(-1) / sigma_goal ** 2


def interpolate_trajs(x, n):
    """Build-defined stand-in for torch_robotics' interpolate_points_v1 (external; call site
    cost_functions.py:118): n evenly spaced points between consecutive waypoints, linear in joint space,
    (B,H,d) -> (B,(H-1)(n+1)+1,d); original waypoints keep their values."""
    x0, x1 = x[:, :-1, None, :], x[:, 1:, None, :]
    alpha = (torch.arange(n + 1, dtype=x.dtype) / (n + 1)).reshape(n + 1, 1)
    seg = x0 + alpha * (x1 - x0)
    return torch.cat((seg.flatten(1, 2), x[:, -1:]), dim=1)


def gpmp2_linear_system(x, robot, field, start_state, goal_state, D, dt,
                        sigma_start, sigma_gp, sigma_goal, sigma_coll, tensor_args, n_interp=None):
    """Dense (A, b, K) exactly as CostComposite.get_linear_system stacks it (cost_functions.py:107-144)
    for the cost list of build_gpmp2_cost_composite (gpmp2.py:23-91): CostGP (:291-314), CostGoalPrior
    (:538-554), CostCollision (:191-231; Jacobian = -d err/d q by autograd, field_factor.py:41-57).
    x (B,H,2D).  Returns A (B,M,N), b (B,M,1), K (B,M,M) with N = 2D*H, M = N + 2D + (H-1).
    goal_state None: the composite has no CostGoalPrior (gpmp2.py:63-73, goal_directed False :135-137).
    """
    B, H, dim = x.shape
    N = dim * H
    Phi = gp_phi(D, dt, tensor_args)
    Qi = gp_Q_inv(D, dt, sigma_gp, tensor_args)
    I = torch.eye(dim, **tensor_args)
    # ---- start prior + GP factors (cost_functions.py:291-314)
    A1 = torch.zeros(B, N, N, **tensor_args)
    b1 = torch.zeros(B, N, 1, **tensor_args)
    K1 = torch.zeros(B, N, N, **tensor_args)
    A1[:, :dim, :dim] = I
    b1[:, :dim, 0] = start_state - x[:, 0]
    K1[:, :dim, :dim] = I / sigma_start ** 2
    e = gp_error(x, Phi)
    for t in range(H - 1):
        r = slice(dim * (t + 1), dim * (t + 2))
        A1[:, r, dim * t:dim * (t + 1)] = Phi          # H1 = Phi   (gp_factor.py:28)
        A1[:, r, dim * (t + 1):dim * (t + 2)] += -I    # H2 = -I    (gp_factor.py:29-31)
        K1[:, r, r] += Qi
    b1[:, dim:, 0] = e.reshape(B, -1)
    # ---- goal prior (cost_functions.py:538-554)
    if goal_state is not None:
        A2 = torch.zeros(B, dim, N, **tensor_args)
        A2[:, :, -dim:] = I
        b2 = (goal_state - x[:, -1]).reshape(B, dim, 1)
        K2 = (I / sigma_goal ** 2).expand(B, dim, dim)
    else:
        A2, b2, K2 = (torch.zeros(B, 0, N, **tensor_args), torch.zeros(B, 0, 1, **tensor_args),
                      torch.zeros(B, 0, 0, **tensor_args))
    # ---- collision (cost_functions.py:191-231): one block of H-1 rows per collision field (gpmp2.py:70-78)
    fields = list(field) if isinstance(field, (list, tuple)) else [field]
    A3s, b3s, K3s = [], [], []
    for fld in fields:
        xg = x.detach().clone().requires_grad_(True)
        q_pos = robot.get_position(xg)
        link_pos = robot.fk_map_collision(q_pos)
        err = fld.compute_cost(q_pos[:, 1:], link_pos[:, 1:]).reshape(B, H - 1)
        err_j = err
        if n_interp:        # Jacobian of the INTERPOLATED trajectory's error w.r.t. the support points; b keeps
            xi = interpolate_trajs(xg, n_interp)                                      # the support-point error
            qi = robot.get_position(xi)                                               # (field_factor.py:42-54,
            err_j = fld.compute_cost(qi[:, 1:], robot.fk_map_collision(qi)[:, 1:])    #  cost_functions.py:115-119)
        Hobst = -torch.autograd.grad(err_j.sum(), xg)[0][:, 1:, :D]                   # field_factor.py:54
        A3f = torch.zeros(B, H - 1, N, **tensor_args)
        for i in range(H - 1):
            A3f[:, i, (i + 1) * dim:(i + 1) * dim + D] = Hobst[:, i]
        A3s.append(A3f)
        b3s.append(err.detach().unsqueeze(-1))
        K3s.append((torch.eye(H - 1, **tensor_args) / sigma_coll ** 2).expand(B, H - 1, H - 1))
    A3, b3 = torch.cat(A3s, 1), torch.cat(b3s, 1)
    K3 = torch.zeros(B, A3.shape[1], A3.shape[1], **tensor_args)
    for i, Kf in enumerate(K3s):
        K3[:, i * (H - 1):(i + 1) * (H - 1), i * (H - 1):(i + 1) * (H - 1)] = Kf
    A = torch.cat([A1, A2, A3], 1)
    b = torch.cat([b1, b2, b3], 1)
    M = A.shape[1]
    K = torch.zeros(B, M, M, **tensor_args)
    o = 0
    for Kp in (K1, K2, K3):
        m = Kp.shape[1]
        K[:, o:o + m, o:o + m] = Kp
        o += m
    return A, b, K


def gpmp2_normal_equations(A, b, K, delta, trust_region):
    """A^T K A (+ damping), A^T K b  (gpmp2.py:355-368, incl. Q9: batch-mean diagonal damping)."""
    N = A.shape[-1]
    I = torch.eye(N, dtype=A.dtype)
    AtK = A.transpose(-2, -1) @ K
    AtA = AtK @ A
    if trust_region:
        JtJ = AtA + delta * (AtA.mean(0) * I)
    else:
        JtJ = AtA + delta * I
    return JtJ, AtK @ b


def gpmp2_iteration(x, robot, field, start_state, goal_state, D, dt, sigma_start, sigma_gp, sigma_goal,
                    sigma_coll, delta, trust_region, step_size, tensor_args, n_interp=None):
    """One GPMP2._step (gpmp2.py:308-342) with method='cholesky' (:451-452) and cost b^T K b (:493-495)."""
    B, H, dim = x.shape
    A, b, K = gpmp2_linear_system(x, robot, field, start_state, goal_state, D, dt,
                                  sigma_start, sigma_gp, sigma_goal, sigma_coll, tensor_args, n_interp=n_interp)
    JtJ, g = gpmp2_normal_equations(A, b, K, delta, trust_region)
    l, _ = torch.linalg.cholesky_ex(JtJ)
    dtheta = torch.cholesky_solve(g, l).view(B, H, dim)
    costs = (b.transpose(1, 2) @ K @ b).reshape(B)
    return dict(means=x + step_size * dtheta, dtheta=dtheta, costs=costs, JtJ=JtJ, g=g)


# ------------------------------------------------------------------------------------------------
# GP-prior initial sampling  (base.py:155-202, mp_priors_multi.py)
# ------------------------------------------------------------------------------------------------

def gp_prior_mean(start_state, goal_state, H, dt, D, tensor_args):
    """Constant-velocity straight line with zero velocity at both ends
    (mp_priors_multi.py:130-151; num_steps = H-1)."""
    n = H - 1
    traj = torch.zeros(H, 2 * D, **tensor_args)
    mean_vel = (goal_state[:D] - start_state[:D]) / (n * dt)
    for i in range(H):
        traj[i, :D] = start_state[:D] * (n - i) * 1. / n + goal_state[:D] * i * 1. / n
    traj[1:-1, D:] = mean_vel.unsqueeze(0)
    return traj


def gp_prior_precision(H, dt, D, sigma_start, sigma_gp, sigma_goal, goal_directed=True):
    """fp64 K^-1 = A^T Q^-1 A of the start / GP / goal factors (mp_priors_multi.py:213-251)."""
    ta = dict(dtype=torch.float64, device='cpu')
    dim = 2 * D
    M = dim * H
    Phi = gp_phi(D, dt, ta)
    A = torch.eye(M, **ta)
    for t in range(H - 1):
        A[dim * (t + 1):dim * (t + 2), dim * t:dim * (t + 1)] += -Phi
    blocks = [torch.eye(dim, **ta) / sigma_start ** 2] + [gp_Q_inv(D, dt, sigma_gp, ta)] * (H - 1)
    if goal_directed:
        bg = torch.zeros(dim, M, **ta)
        bg[:, -dim:] = torch.eye(dim, **ta)
        A = torch.cat((A, bg))
        blocks.append(torch.eye(dim, **ta) / sigma_goal ** 2)
    Qinv = torch.block_diag(*blocks)
    return A.t() @ Qinv @ A


def gp_prior_precision_general(H, dt, D, K_s_inv, K_gp_inv, K_g_inv=None):
    """fp64 K^-1 = A^T Q^-1 A for ARBITRARY start / GP / goal precisions, assembled densely exactly as
    MultiMPPrior.get_const_vel_covariance does (mp_priors_multi.py:213-251)."""
    ta = dict(dtype=torch.float64, device='cpu')
    dim = 2 * D
    M = dim * H
    Phi = gp_phi(D, dt, ta)
    A = torch.eye(M, **ta)
    for t in range(H - 1):
        A[dim * (t + 1):dim * (t + 2), dim * t:dim * (t + 1)] += -Phi
    blocks = [torch.as_tensor(K_s_inv, **ta)] + [torch.as_tensor(K_gp_inv, **ta)] * (H - 1)
    if K_g_inv is not None:
        bg = torch.zeros(dim, M, **ta)
        bg[:, -dim:] = torch.eye(dim, **ta)
        A = torch.cat((A, bg))
        blocks.append(torch.as_tensor(K_g_inv, **ta))
    return A.t() @ torch.block_diag(*blocks) @ A


# ------------------------------------------------------------------------------------------------
# MPPI on point-particle dynamics  (mppi.py, priors/gaussian.py, dynamics/point.py)
# ------------------------------------------------------------------------------------------------

def mppi_covariance(sigma, T, ctrl_dim, cov_type, tensor_args):
    """diag_Cov (gaussian.py:143-163) / const_ctrl_Cov (:166-198); shape (T,T,ctrl_dim)."""
    if cov_type == 'indep_ctrl':
        Cov = torch.eye(T, **tensor_args).unsqueeze(-1).repeat(1, 1, ctrl_dim)
        return Cov * (torch.tensor(sigma, **tensor_args) ** 2 if isinstance(sigma, (list, tuple)) else sigma ** 2)
    Lm = torch.tril(torch.ones(T, T - 1, **tensor_args), diagonal=-1)
    LLt = Lm @ Lm.t() + torch.ones(T, T, **tensor_args)
    s = torch.tensor(sigma, **tensor_args) if isinstance(sigma, (list, tuple)) else sigma
    return LLt.unsqueeze(-1).repeat(1, 1, ctrl_dim) * s ** 2


def mppi_iteration(mean, eps, scale_tril, cov_inv, state, goal_state, dt, ctrl_min, ctrl_max,
                   c_weights, discount_seq, temp, step_size, state_dim, shift_cost=0.0):
    """One pass of MPPI.optimize's loop body (mppi.py:145-152), velocity- or acceleration-control
    point particle (point.py:102-140, :154-226).

    mean (T,c); eps (c,S,T) standard normal drawn per control dim in order (gaussian.py:276-298);
    scale_tril (c,T,T); cov_inv (c,T,T).  Returns dict(controls (S,T,c), states (S,T,sd), costs (S,1),
    weights (S,1), mean (T,c)).
    """
    c = mean.shape[1]
    S, T = eps.shape[1], eps.shape[2]
    U = torch.stack([mean[:, i] + (scale_tril[i] @ eps[i].unsqueeze(-1)).squeeze(-1) for i in range(c)], -1)
    X = torch.empty(S, T, state.shape[-1], dtype=mean.dtype)
    X[:, 0] = state
    for t in range(T - 1):                                   # mppi.py:205-209
        u = U[:, t].clamp(min=ctrl_min, max=ctrl_max)        # point.py:112
        xdot = torch.cat((X[:, t, state_dim:], u), dim=-1)   # point.py:114-118
        X[:, t + 1] = X[:, t] + xdot * dt
    dX = X - goal_state[..., :state_dim]
    pos = (torch.square(dX[..., :state_dim]) * c_weights['pos']).sum(-1) * discount_seq
    vel = (torch.square(dX[..., state_dim:c]) * c_weights['vel']).sum(-1) * discount_seq   # Q8: empty slice
    ctl = (torch.square(U) * c_weights['ctrl']).sum(-1) * discount_seq
    term = (torch.square(dX[:, -1]) * c_weights['pos_T']).sum(-1) * discount_seq[-1]
    costs = (pos.sum(1) + vel.sum(1) + ctl.sum(1) + term + shift_cost).reshape(S, 1)
    for i in range(c):                                       # mppi.py:125-128
        costs = costs + temp * (U[..., i] @ cov_inv[i] @ mean[..., i]).reshape(-1, 1)
    w = torch.softmax(-costs / temp, dim=0)                  # mppi.py:73-76
    new_mean = mean + step_size * (w.reshape(-1, 1, 1) * (U - mean.unsqueeze(0))).sum(0)
    return dict(controls=U, states=X, costs=costs, weights=w, mean=new_mean)


# ------------------------------------------------------------------------------------------------
# StochGPMP  (mp_baselines/planners/stoch_gpmp.py)
# ------------------------------------------------------------------------------------------------

def goal_prior_eval(x, goal_state, sigma_goal):
    """CostGoalPrior.eval for one goal (cost_functions.py:520-536): (goal - x_{H-1})^T K (goal - x_{H-1})."""
    err = goal_state - x[:, -1]
    return (err * err).sum(-1) / sigma_goal ** 2


def stoch_gpmp_iteration(means, eps, scale_tril, Sigma_inv, cost_fn, temperature, step_size):
    """One pass of StochGPMP.optimize's loop body (stoch_gpmp.py:289-298): sample_and_eval (:244-265) with
    _get_costs (:235-242: composite cost + T * V Sigma^-1 U^T) and _update_distribution (:267-279).

    means (P,H,dim); eps (S,P,M) standard normals in MultivariateNormal's draw order; scale_tril (M,M) of the
    sampling prior; Sigma_inv (M,M) its precision.  Returns dict(samples (P,S,H,dim), costs, weights, means).
    """
    P, H, dim = means.shape
    S = eps.shape[0]
    M = H * dim
    smp = means.reshape(1, P, M) + (scale_tril @ eps.unsqueeze(-1)).squeeze(-1)      # mp_priors_multi.py:253-256
    samples = smp.view(S, P, H, dim).transpose(1, 0)
    costs = cost_fn(samples.reshape(P * S, H, dim)).reshape(P, S)
    V = samples.reshape(P, S, M)
    U = means.reshape(P, 1, M)
    costs = costs + temperature * (V @ Sigma_inv @ U.transpose(1, 2)).squeeze(2)     # stoch_gpmp.py:239-241
    w = torch.softmax(-costs / temperature, dim=1)
    grad = (w.reshape(P, S, 1, 1) * (samples - means.unsqueeze(1))).sum(1)
    return dict(samples=samples, costs=costs, weights=w, means=means + step_size * grad)


# ------------------------------------------------------------------------------------------------
# HybridPlanner warm start  (hybrid_planner.py:42-66)
# ------------------------------------------------------------------------------------------------

def resample_path(path, H, dt):
    """Build-defined stand-in for torch_robotics' smoothen_trajectory(set_average_velocity=True) (external;
    call site hybrid_planner.py:52-55): polyline (L,D) -> (H,2D); positions uniform in arc length and linear
    within a segment, first/last point exact; velocity (last-first)/((H-1) dt) on interior points, 0 at ends."""
    import numpy as np
    P = np.asarray(path, dtype=np.float64)
    L, D = P.shape
    seg = np.sqrt(((P[1:] - P[:-1]) ** 2).sum(-1)) if L > 1 else np.zeros(0)
    cum = np.concatenate([[0.0], np.cumsum(seg)])
    out = np.zeros((H, 2 * D))
    for h in range(H):
        target = cum[-1] * h / (H - 1)
        k0 = int(np.searchsorted(cum, target, side='right') - 1)
        k0 = min(max(k0, 0), L - 1)
        k1 = min(k0 + 1, L - 1)
        span = cum[k1] - cum[k0]
        a = (target - cum[k0]) / span if span > 0 else 0.0
        out[h, :D] = P[k0] + a * (P[k1] - P[k0])
    out[0, :D], out[-1, :D] = P[0], P[-1]
    out[1:-1, D:] = (P[-1] - P[0]) / ((H - 1) * dt)
    return torch.from_numpy(out)
