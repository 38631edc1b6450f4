"""Cost-function objects with the reference's API surface (mp_baselines/planners/costs/cost_functions.py)
whose evaluation runs in the HIP library.

``Cost.__call__(trajs) -> (B,)`` / ``eval`` / ``get_linear_system`` keep the reference's meaning
(cost_functions.py:18-53).  The planners recognise these objects and fuse their arithmetic into the
planner kernels (no per-term launches); called on their own, ``eval`` launches the stand-alone HIP
collision-cost kernel (mpb_cost_collision_eval) or the trajectory-terms kernel (mpb_cost_terms_eval:
GP prior, start / goal priors, CHOMP smoothness, joint limits -- any subset of them in one pass, which
is how CostComposite evaluates them).  There is no CPU path: trajectories must be CUDA tensors.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from ... import ops


class Cost(ABC):
    """cost_functions.py:18-53."""

    def __init__(self, robot, n_support_points, tensor_args=None, **kwargs):
        self.robot = robot
        self.n_dof = robot.q_dim
        self.dim = 2 * self.n_dof
        self.n_support_points = n_support_points
        self.tensor_args = tensor_args

    def set_cost_factors(self):
        pass

    def __call__(self, trajs, **kwargs):
        return self.eval(trajs, **kwargs)

    @abstractmethod
    def eval(self, trajs, **kwargs):
        pass

    @abstractmethod
    def get_linear_system(self, trajs, **kwargs):
        pass

    def get_q_pos_vel_and_fk_map(self, trajs, **kwargs):
        """cost_functions.py:41-53: (trajs (B,H,d), q_pos, q_vel, H_positions) with H_positions = positions of the
        robot's collision spheres (B,H,L,3), computed by the FK kernel (mpb_fk_collision_points)."""
        trajs = self._as_3d(trajs)
        q_pos = self.robot.get_position(trajs)
        q_vel = self.robot.get_velocity(trajs)
        geom = self.__dict__.get('_fk_geom')
        if geom is None or geom.buf.device != trajs.device:
            import numpy as np
            from ...geometry import CollisionField
            far = CollisionField(spheres=np.array([[1.0e6, 1.0e6, 1.0e6, 1.0]], np.float32))     # FK only: the field is unused
            geom = self.__dict__['_fk_geom'] = ops.DeviceGeometry(self.robot, far, trajs.device, keep_all_links=True)
        return trajs, q_pos, q_vel, ops.fk_collision_points(trajs, geom)

    @staticmethod
    def _as_3d(trajs):
        """Accept (B,H,d) or (N,B,H,d) like get_q_pos_vel_and_fk_map (cost_functions.py:41-48)."""
        assert trajs.ndim == 3 or trajs.ndim == 4
        if trajs.ndim == 4:
            trajs = trajs.reshape(-1, trajs.shape[-2], trajs.shape[-1])
        return trajs.contiguous()


class CostCollision(Cost):
    """Hinge collision cost over one collision field (cost_functions.py:147-231).

    eval: ``1/sigma_coll^2 * sum_{h>=1} field_cost(q_h)`` (traj_range [1, None], field_factor.py:31-39).
    """

    def __init__(self, robot, n_support_points, field=None, sigma_coll=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.field = field
        self.sigma_coll = sigma_coll
        self._geom = None

    @property
    def k_sigma(self):
        return 1.0 / (self.sigma_coll ** 2)          # FieldFactor.K (field_factor.py:15)

    def device_geometry(self, device):
        if self._geom is None or self._geom.buf.device != torch.device(device):
            self._geom = ops.DeviceGeometry(self.robot, self.field, device)
        return self._geom

    def eval(self, trajs, **observation):
        if self.field is None:
            return 0
        trajs = self._as_3d(trajs)
        return ops.cost_collision_eval(trajs, self.device_geometry(trajs.device), self.k_sigma)

    def eval_with_grad(self, trajs, weight=1.0):
        """(cost (B,), d cost / d trajs (B,H,d)): what the reference gets from autograd (chomp.py:139)."""
        trajs = self._as_3d(trajs)
        return ops.cost_collision_grad(trajs, self.device_geometry(trajs.device), self.k_sigma, weight=weight)

    def get_linear_system(self, trajs, n_interpolated_points=None, **observation):
        """Dense (A, b, K) rows of the collision factor (cost_functions.py:191-231): row i has H_obst = -d err / d q at
        columns (i+1)*dim .. +n_dof, b = the support points' errors, K = I / sigma^2.  With `n_interpolated_points`
        (forwarded by CostComposite.get_linear_system, cost_functions.py:112-119) the Jacobian is that of the
        INTERPOLATED trajectory's error w.r.t. the support points (field_factor.py:42-54) while b stays the support
        points' own error.  Values come from the kernel GPMP2 itself uses (mpb_gpmp2_linearize); the dense form is
        an inspection aid -- GPMP2 never materialises it."""
        trajs = self._as_3d(trajs)
        B, H, d = trajs.shape
        assert d == self.dim, 'get_linear_system works on (B, H, 2*n_dof) trajectories'
        rows = ops.gpmp2_collision_rows(trajs, self.device_geometry(trajs.device), n_interp=n_interpolated_points or 0)[0]
        N, D = self.dim * H, self.n_dof
        A = torch.zeros(B, H - 1, N, device=trajs.device, dtype=trajs.dtype)
        for i in range(H - 1):
            A[:, i, (i + 1) * self.dim:(i + 1) * self.dim + D] = rows[:, i + 1, :D]
        b = rows[:, 1:, D].unsqueeze(-1).clone()
        K = self.k_sigma * torch.eye(H - 1, device=trajs.device, dtype=trajs.dtype).repeat(B, 1, 1)
        return A, b, K


class _TrajectoryTermCost(Cost):
    """A cost that depends on the trajectory alone (no FK / field): served by mpb_cost_terms_eval.
    ``term_spec(device)`` returns the keyword arguments of ops.cost_terms_eval for this term with unit
    composite weight; CostComposite merges the specs of its members into one launch."""

    def term_spec(self, device):
        raise NotImplementedError

    def _dev(self, name, value, device):
        """float32 device copy of a small constant, cached per device."""
        cache = self.__dict__.setdefault('_dev_cache', {})
        key = (name, str(device))
        if key not in cache:
            cache[key] = torch.as_tensor(np.asarray(value.detach().cpu() if torch.is_tensor(value) else value),
                                         dtype=torch.float32).to(device).contiguous()
        return cache[key]

    def eval(self, trajs, **observation):
        trajs = self._as_3d(trajs)
        out, _ = ops.cost_terms_eval(trajs, self.n_dof, **self.term_spec(trajs.device))
        return out

    def get_linear_system(self, trajs, **observation):
        return None                                   # the reference's `pass` (cost_functions.py:356-357, :389, :428)


class CostGPTrajectory(_TrajectoryTermCost):
    """GP-prior smoothness cost sum_t e_t^T Q^-1 e_t (cost_functions.py:317-357; GPFactor gp_factor.py:4-65)."""

    def __init__(self, robot, n_support_points, dt, sigma_gp=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.dt = dt
        self.sigma_gp = sigma_gp

    def term_spec(self, device):
        return dict(terms={'gp'}, dt=self.dt, k_gp=1.0 / self.sigma_gp ** 2)


class CostGPTrajectoryPositionOnlyWrapper(CostGPTrajectory):
    """Same cost on position-only trajectories: velocities by central differences inside the kernel
    (cost_functions.py:360-368)."""

    def term_spec(self, device):
        return dict(terms={'gp'}, dt=self.dt, k_gp=1.0 / self.sigma_gp ** 2, vel_fd=True)


class CostGP(_TrajectoryTermCost):
    """Start prior + GP prior (cost_functions.py:234-314)."""

    def __init__(self, robot, n_support_points, start_state, dt, sigma_params, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.start_state = start_state
        self.dt = dt
        self.sigma_start = sigma_params['sigma_start']
        self.sigma_gp = sigma_params['sigma_gp']

    def term_spec(self, device):
        return dict(terms={'gp', 'start'}, dt=self.dt, k_gp=1.0 / self.sigma_gp ** 2,
                    k_start=1.0 / self.sigma_start ** 2, start_state=self._dev('start', self.start_state, device))

    def get_linear_system(self, trajs, **observation):
        """Dense (A, b, K) of the start + GP factors (cost_functions.py:291-314).  Inspection aid (torch
        assembly on the device): GPMP2 never materialises it, it works on the block-tridiagonal form."""
        trajs = self._as_3d(trajs)
        B, H, dim = trajs.shape
        D, dt, dev = self.n_dof, self.dt, trajs.device
        kw = dict(device=dev, dtype=trajs.dtype)
        N = dim * H
        A, b, K = torch.zeros(B, N, N, **kw), torch.zeros(B, N, 1, **kw), torch.zeros(B, N, N, **kw)
        I, Z = torch.eye(D, **kw), torch.zeros(D, D, **kw)
        Phi = torch.cat((torch.cat((I, dt * I), 1), torch.cat((Z, I), 1)), 0)
        Qc = I / self.sigma_gp ** 2
        Qi = torch.cat((torch.cat((12. * dt ** -3. * Qc, -6. * dt ** -2. * Qc), 1),
                        torch.cat((-6. * dt ** -2. * Qc, 4. * dt ** -1. * Qc), 1)), 0)
        A[:, :dim, :dim] = torch.eye(dim, **kw)                                                # unary_factor.py:28
        b[:, :dim, 0] = self.start_state.to(**kw) - trajs[:, 0]
        K[:, :dim, :dim] = torch.eye(dim, **kw) / self.sigma_start ** 2
        for t in range(H - 1):
            r = slice(dim * (t + 1), dim * (t + 2))
            A[:, r, dim * t:dim * (t + 1)] = Phi                                               # H1 (gp_factor.py:28)
            A[:, r, r] += -torch.eye(dim, **kw)                                                # H2 (gp_factor.py:29-31)
            K[:, r, r] += Qi
        b[:, dim:, 0] = (trajs[:, 1:] - trajs[:, :-1] @ Phi.t()).reshape(B, -1)
        return A, b, K


class CostSmoothnessCHOMP(_TrajectoryTermCost):
    """x^T R x with CHOMP's precision R (cost_functions.py:371-390; chomp.py:81-101).  The reference
    returns whatever torch_robotics' batched_weighted_dot_prod returns (external); this build defines
    the cost as the sum over state columns -> (B,)."""

    def __init__(self, robot, n_support_points, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.dt = robot.dt
        assert self.dt is not None, 'CostSmoothnessCHOMP reads robot.dt (cost_functions.py:380)'

    def term_spec(self, device):
        return dict(terms={'smooth'}, dt=self.dt, k_smooth=1.0)


class CostJointLimits(_TrajectoryTermCost):
    """Squared violation of q_min + eps / q_max - eps (cost_functions.py:393-429).  As in the reference the
    result is ONE scalar for the whole batch (its `.sum(-1)` acts on a 1-D gather); CostComposite adds
    it to every trajectory's cost."""

    def __init__(self, robot, n_support_points, eps=np.deg2rad(3), **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.eps = float(eps)

    def term_spec(self, device):
        return dict(terms={'jlim'}, k_jlim=1.0, jl_eps=self.eps, q_min=self._dev('q_min', self.robot.q_min, device),
                    q_max=self._dev('q_max', self.robot.q_max, device))

    def eval(self, trajs, **observation):
        assert trajs.ndim == 3                                                                  # cost_functions.py:407
        trajs = trajs.contiguous()
        _, total = ops.cost_terms_eval(trajs, self.n_dof, broadcast_jlim=False, **self.term_spec(trajs.device))
        return total.to(torch.float32)


class CostGoalPrior(_TrajectoryTermCost):
    """Unary prior on the last state, one goal per block of num_particles_per_goal * num_samples
    trajectories (cost_functions.py:488-554)."""

    def __init__(self, robot, n_support_points, multi_goal_states=None, num_particles_per_goal=None,
                 num_samples=None, sigma_goal_prior=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.multi_goal_states = multi_goal_states
        self.num_goals = multi_goal_states.shape[0]
        self.num_particles_per_goal = num_particles_per_goal
        self.num_particles = num_particles_per_goal * self.num_goals
        self.num_samples = num_samples
        self.sigma_goal_prior = sigma_goal_prior

    def term_spec(self, device):
        return dict(terms={'goal'}, k_goal=1.0 / self.sigma_goal_prior ** 2,
                    goal_states=self._dev('goals', self.multi_goal_states, device),
                    trajs_per_goal=self.num_particles_per_goal * self.num_samples)

    def eval(self, trajs, **observation):
        trajs = self._as_3d(trajs)
        assert trajs.shape[0] == self.num_goals * self.num_particles_per_goal * self.num_samples  # the reshape at :525
        return super().eval(trajs, **observation)

    def get_linear_system(self, trajs, **observation):
        """(A, b, K) of the goal factor (cost_functions.py:538-554); one row block per particle."""
        trajs = self._as_3d(trajs)
        B, H, dim = trajs.shape
        kw = dict(device=trajs.device, dtype=trajs.dtype)
        A = torch.zeros(B, dim, dim * H, **kw)
        A[:, :, -dim:] = torch.eye(dim, **kw)
        goals = self.multi_goal_states.to(**kw).repeat_interleave(self.num_particles_per_goal, 0)
        b = (goals - trajs[:, -1]).unsqueeze(-1)
        K = (torch.eye(dim, **kw) / self.sigma_goal_prior ** 2).repeat(B, 1, 1)
        return A, b, K


class CostGoal(Cost):
    """Field cost of the LAST waypoint only (FieldFactor with traj_range [-1, None]; cost_functions.py:432-485);
    0 without a field.  With a CollisionField it is the collision kernel restricted to h = H-1."""

    def __init__(self, robot, n_support_points, field=None, sigma_goal=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.field = field
        self.sigma_goal = sigma_goal
        self._geom = None

    def eval(self, trajs, x_trajs=None, **observation):
        if self.field is None:
            return 0
        trajs = self._as_3d(trajs)
        if self._geom is None or self._geom.buf.device != trajs.device:
            self._geom = ops.DeviceGeometry(self.robot, self.field, trajs.device)
        return ops.cost_collision_eval(trajs, self._geom, 1.0 / self.sigma_goal ** 2, h_begin=trajs.shape[1] - 1)

    def get_linear_system(self, trajs, x_trajs=None, **observation):
        if self.field is None:
            return None, None, None
        trajs = self._as_3d(trajs)
        B, H, d = trajs.shape
        if self._geom is None or self._geom.buf.device != trajs.device:
            self._geom = ops.DeviceGeometry(self.robot, self.field, trajs.device)
        _, pw = ops.cost_collision_eval(trajs, self._geom, 1.0, h_begin=H - 1, per_waypoint=True)
        _, grad = ops.cost_collision_grad(trajs, self._geom, 1.0, h_begin=H - 1)
        A = torch.zeros(B, 1, self.dim * H, device=trajs.device, dtype=trajs.dtype)
        A[:, 0, (H - 1) * self.dim:(H - 1) * self.dim + self.n_dof] = -grad[:, -1, :self.n_dof]
        b = pw[:, -1:].unsqueeze(-1)
        K = (1.0 / self.sigma_goal ** 2) * torch.ones(B, 1, 1, device=trajs.device, dtype=trajs.dtype)
        return A, b, K


def _merge_term_specs(members):
    """Group (spec, weight) pairs into as few mpb_cost_terms_eval launches as possible: a member joins a
    group when none of its terms is already in it and the shared scalar dt agrees."""
    groups = []
    for spec, w in members:
        scaled = dict(spec)
        for k in ('k_gp', 'k_start', 'k_goal', 'k_smooth', 'k_jlim'):
            if k in scaled:
                scaled[k] = scaled[k] * w
        for g in groups:
            if g['terms'] & scaled['terms']:
                continue
            if 'dt' in g and 'dt' in scaled and g['dt'] != scaled['dt']:
                continue
            g.update({k: v for k, v in scaled.items() if k != 'terms'})
            g['terms'] = g['terms'] | scaled['terms']
            break
        else:
            scaled['terms'] = set(scaled['terms'])
            groups.append(scaled)
    return groups


class CostComposite(Cost):
    """Weighted sum of member costs (cost_functions.py:56-144)."""

    def __init__(self, robot, n_support_points, cost_list, weights_cost_l=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.cost_l = cost_list
        self.weight_cost_l = weights_cost_l if weights_cost_l is not None else [1.0] * len(cost_list)

    def eval(self, trajs, trajs_interpolated=None, return_invidual_costs_and_weights=False, **kwargs):
        """cost_functions.py:70-105.  Collision members see `trajs_interpolated` when it is given (:77-84);
        all trajectory-only members are evaluated by ONE mpb_cost_terms_eval launch."""
        trajs = self._as_3d(trajs)
        trajs_coll = trajs if trajs_interpolated is None else self._as_3d(trajs_interpolated)
        if return_invidual_costs_and_weights:
            return [c(trajs_coll if isinstance(c, CostCollision) else trajs, **kwargs) for c in self.cost_l], \
                self.weight_cost_l
        total = None
        term_members = []
        for cost, w in zip(self.cost_l, self.weight_cost_l):
            if isinstance(cost, CostCollision):
                if cost.field is None:
                    continue
                c = ops.cost_collision_eval(trajs_coll, cost.device_geometry(trajs.device), cost.k_sigma, weight=w)
            elif isinstance(cost, _TrajectoryTermCost):
                term_members.append((cost.term_spec(trajs.device), float(w)))
                continue
            else:
                c = w * cost(trajs, **kwargs)
            total = c if total is None else total + c
        for spec in _merge_term_specs(term_members):
            if total is None or not (torch.is_tensor(total) and total.ndim == 1 and total.is_contiguous()):
                extra, _ = ops.cost_terms_eval(trajs, self.n_dof, **spec)
                total = extra if total is None else total + extra
            else:
                ops.cost_terms_eval(trajs, self.n_dof, out=total, accumulate=True, **spec)
        return 0 if total is None else total

    def device_plan(self, device):
        """How a planner kernel pipeline can serve this composite without leaving the device:
        (collision members [(CostCollision, weight)], merged term specs, other members [(cost, weight)])."""
        coll, terms, other = [], [], []
        for cost, w in zip(self.cost_l, self.weight_cost_l):
            if isinstance(cost, CostCollision):
                if cost.field is not None:
                    coll.append((cost, float(w)))
            elif isinstance(cost, _TrajectoryTermCost):
                terms.append((cost.term_spec(device), float(w)))
            else:
                other.append((cost, float(w)))
        return coll, _merge_term_specs(terms), other

    def get_linear_system(self, trajs, n_interpolated_points=None, **kwargs):
        """cost_functions.py:107-144: row-concatenated A, b and block-diagonal K of the members that have a linear
        system; `n_interpolated_points` reaches the collision members (their Jacobian is then the interpolated
        trajectory's, :112-119).  Like the reference, a member whose get_linear_system returns None (the
        reference's bare `pass`: CostGPTrajectory, CostSmoothnessCHOMP, CostJointLimits) raises TypeError at the
        unpack (:122-126); (None, None, None) members are skipped (:129-130).  Composite weights do not enter
        (the reference ignores weight_cost_l here)."""
        trajs = self._as_3d(trajs)
        As, bs, Ks = [], [], []
        for cost in self.cost_l:
            A, b, K = cost.get_linear_system(trajs, n_interpolated_points=n_interpolated_points, **kwargs)
            if A is None or b is None or K is None:
                continue
            As.append(A); bs.append(b); Ks.append(K)
        A, b = torch.cat(As, 1), torch.cat(bs, 1)
        K = torch.zeros(A.shape[0], A.shape[1], A.shape[1], device=A.device, dtype=A.dtype)
        o = 0
        for Kp in Ks:
            m = Kp.shape[1]
            K[:, o:o + m, o:o + m] = Kp
            o += m
        return A, b, K

    def collision_terms(self):
        """[(CostCollision, weight)] of the members that carry a field."""
        return [(c, float(w)) for c, w in zip(self.cost_l, self.weight_cost_l)
                if isinstance(c, CostCollision) and c.field is not None]

    def single_collision_term(self):
        """(collision evaluator, weight) if this composite consists of collision fields only (1..4 of them) -- the
        case the planner kernels fuse; None otherwise."""
        terms = self.collision_terms()
        if not terms or len(terms) != len(self.cost_l):
            return None
        return merge_collision_terms(terms)


class MergedCollision:
    """Several CostCollision members (same robot) as ONE chained geometry buffer: the kernels evaluate
    weight * k_sigma * sum_f s_f cost_f with k_sigma, weight of the first member and
    s_f = (w_f k_f) / (w_0 k_0).  Quacks like CostCollision where the planners need it."""

    def __init__(self, terms):
        (c0, w0) = terms[0]
        assert all(c.robot is c0.robot for c, _ in terms), 'collision members must share the robot'
        self.robot = c0.robot
        self.fields = [c.field for c, _ in terms]
        self.k_sigma = c0.k_sigma
        self.weight = w0
        self.scales = [(w * c.k_sigma) / (w0 * c0.k_sigma) for c, w in terms]
        self._geom = None

    def device_geometry(self, device):
        if self._geom is None or self._geom.buf.device != torch.device(device):
            self._geom = ops.DeviceGeometry(self.robot, self.fields, device, scales=self.scales)
        return self._geom


def merge_collision_terms(terms):
    """[(CostCollision, weight)] -> (evaluator with device_geometry / k_sigma, weight) or None (too many fields)."""
    if len(terms) == 1:
        return terms[0]
    if len(terms) > 4 or terms[0][1] == 0:
        return None
    m = MergedCollision(terms)
    return m, m.weight


def fusable_collision(cost):
    """Return (collision evaluator, weight) when `cost` is a collision cost the kernels can fuse, else None."""
    if isinstance(cost, CostCollision) and cost.field is not None:
        return cost, 1.0
    if isinstance(cost, CostComposite):
        key = tuple((id(c), float(w)) for c, w in zip(cost.cost_l, cost.weight_cost_l))
        cached = cost.__dict__.get('_fused_cache')
        if cached is None or cached[0] != key:
            cost.__dict__['_fused_cache'] = (key, cost.single_collision_term())
        return cost.__dict__['_fused_cache'][1]
    return None


def device_plan(cost, device):
    """(collision evaluator or None, its weight, merged trajectory-term specs) when every member of `cost` is
    served by the HIP library (collision fields -- up to four, chained -- and trajectory-only terms): the case a
    planner can run as sample kernel -> term kernel -> update kernel with no host round trip; None otherwise."""
    if isinstance(cost, CostComposite):
        key = ('plan', str(device)) + tuple((id(c), float(w)) for c, w in zip(cost.cost_l, cost.weight_cost_l))
        cached = cost.__dict__.get('_plan_cache')
        if cached is not None and cached[0] == key:
            return cached[1]
        coll, groups, other = cost.device_plan(device)
        plan = None
        if not other and (coll or groups):
            merged = merge_collision_terms(coll) if coll else (None, 0.0)
            if merged is not None:
                plan = (merged[0], merged[1], groups)
        cost.__dict__['_plan_cache'] = (key, plan)
        return plan
    if isinstance(cost, CostCollision):
        return (cost, 1.0, []) if cost.field is not None else None
    if isinstance(cost, _TrajectoryTermCost):
        return None, 0.0, _merge_term_specs([(cost.term_spec(device), 1.0)])
    return None
