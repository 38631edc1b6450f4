"""GPMP2 with the reference's class surface (mp_baselines/planners/gpmp2.py), Gauss-Newton steps on the GPU.

The reference stacks a dense (A, b, K) per particle, forms A^T K A and calls a dense Cholesky
(gpmp2.py:308-368, :451-452).  mpb_gpmp2_step assembles the same normal equations in their true
block-tridiagonal form and solves them by block Cholesky in fp64 (csrc/mpb_gpmp2.hip); nothing dense is
ever materialised, which is also what lets config C4 (B=2048, H=128, D=7: >150 GB dense) run at all.
"""
import math

import torch
import torch.distributed as dist

from .. import ops
from .base import OptimizationPlanner
from .costs.cost_functions import Cost, CostCollision, CostComposite, CostGP, CostGoalPrior, _TrajectoryTermCost

SOLVE_METHODS = ('cholesky', 'inverse', 'lstq')


def build_gpmp2_cost_composite(robot=None, n_support_points=None, dt=None, start_state=None, multi_goal_states=None,
                               num_particles_per_goal=None, collision_fields=None, extra_costs=[], sigma_start=1e-5,
                               sigma_gp=1e-2, sigma_coll=1e-5, sigma_goal_prior=1e-5, num_samples=64, tensor_args=None,
                               **kwargs):
    """gpmp2.py:23-91: CostGP (start + GP prior), CostGoalPrior when there are goals, one CostCollision per field,
    then the extra costs -- as a CostComposite (the object behind GPMP2.cost / StochGPMP.cost)."""
    cost_func_list = []
    start_zero_vel = torch.cat((start_state, torch.zeros_like(start_state)))
    cost_func_list.append(CostGP(robot, n_support_points, start_zero_vel, dt,
                                 dict(sigma_start=sigma_start, sigma_gp=sigma_gp), tensor_args=tensor_args))
    if multi_goal_states is not None:
        goals_zero_vel = torch.cat((multi_goal_states, torch.zeros_like(multi_goal_states)), dim=-1)
        cost_func_list.append(CostGoalPrior(robot, n_support_points, multi_goal_states=goals_zero_vel,
                                            num_particles_per_goal=num_particles_per_goal, num_samples=num_samples,
                                            sigma_goal_prior=sigma_goal_prior, tensor_args=tensor_args))
    for field in collision_fields:
        cost_func_list.append(CostCollision(robot, n_support_points, field=field, sigma_coll=sigma_coll,
                                            tensor_args=tensor_args))
    if extra_costs:
        cost_func_list.append(*extra_costs)          # gpmp2.py:82-84: works for exactly one extra cost, as there
    return CostComposite(robot, n_support_points, cost_func_list, tensor_args=tensor_args)


class GPMP2(OptimizationPlanner):
    """Drop-in for mp_baselines.planners.gpmp2.GPMP2 (ctor kwargs gpmp2.py:94-115 plus the cost kwargs of
    build_gpmp2_cost_composite, gpmp2.py:23-91: collision_fields, sigma_start, sigma_gp, sigma_coll,
    sigma_goal_prior).

    Differences, all forced by the hot path living on the GPU:
      * ``collision_fields`` (+ a CostCollision given as ``extra_costs``) hold one to four CollisionFields in total
        (chained in one geometry buffer, one block of collision rows per field like the reference); a CostGP / CostGoalPrior
        on the planner's own start / goal states as extra cost is folded into the solve's sigmas (its precision adds);
        any OTHER extra cost that has a get_linear_system (the reference stacks whatever has one) switches the planner to the
        reference's dense step on the device (torch.linalg in fp64: correct, dense-sized); a cost without one raises
        (in the reference it fails at the unpack, cost_functions.py:122-126);
      * ``solver_params['method']``: 'cholesky', 'inverse' and 'lstq' (gpmp2.py:432-491) all run the block solve --
        they are three dense solvers of the same SPD system; 'cholesky-sparse' raises like the reference (:457);
        ``_get_grad_terms`` / ``get_torch_solve`` exist as the reference's dense methods (for callers / subclasses), the
        step itself does not go through them;
      * without goals (``multi_goal_states=None``, gpmp2.py:135-137) the goal factor is left out of the system;
      * extra kwarg ``process_group``: when given, the trust-region damping's batch mean (quirk Q9,
        gpmp2.py:361-367) is all-reduced over the group so that sharded runs equal the unsharded one.
    """

    def __init__(self, robot=None, n_dof=None, n_support_points=None, n_interpolated_points=None,
                 num_particles_per_goal=None, opt_iters=None, dt=None, start_state=None, step_size=1.,
                 multi_goal_states=None, initial_particle_means=None, sigma_start_init=None, sigma_start_sample=None,
                 sigma_goal_init=None, sigma_goal_sample=None, sigma_gp_init=None, solver_params=None,
                 stop_criteria=None, collision_fields=None, extra_costs=None, sigma_start=1e-5, sigma_gp=1e-2,
                 sigma_coll=1e-5, sigma_goal_prior=1e-5, tensor_args=None, process_group=None, **kwargs):
        super().__init__(name='GPMP', n_dof=n_dof, n_support_points=n_support_points,
                         num_particles_per_goal=num_particles_per_goal, opt_iters=opt_iters, dt=dt,
                         start_state=start_state, initial_particle_means=initial_particle_means,
                         multi_goal_states=multi_goal_states, sigma_start_init=sigma_start_init,
                         sigma_goal_init=sigma_goal_init, sigma_gp_init=sigma_gp_init, pos_only=False,
                         tensor_args=tensor_args)
        # quirk Q13: the reference's GPMP2.__init__ accepts n_interpolated_points but never forwards it to the base
        # class (gpmp2.py:94-131), so the attribute is None after construction (base.py:85) and the interpolated
        # collision Jacobian (cost_functions.py:115-119) only runs when a caller sets the attribute afterwards.
        # Same here: the ctor argument is ignored, the attribute is honoured by _step.
        self.n_interpolated_points = None
        collision_fields = list(collision_fields or [])
        extra_costs = list(extra_costs or [])
        scales = [1.0] * len(collision_fields)
        n_fields_own = len(collision_fields)
        # the sigmas the block solve runs on: an extra CostGP / CostGoalPrior on the planner's OWN start / goal states is one
        # more factor of a kind the solve already assembles in registers -- its rows stack under the planner's
        # (cost_functions.py:107-144), i.e. its precision ADDS to theirs: folded into effective sigmas (round 4)
        eff = dict(start=float(sigma_start), gp=float(sigma_gp), goal=float(sigma_goal_prior))
        merge = lambda a, b: (1.0 / a ** 2 + 1.0 / b ** 2) ** -0.5
        same = lambda a, b: a is not None and b is not None and tuple(a.shape) == tuple(b.shape) and \
            bool(torch.equal(a.detach().cpu().float(), b.detach().cpu().float()))
        dense_extras = []

        def _same_problem(c, what):
            """an extra factor is merged only if it is built for THIS planner's horizon and DoF: the reference stacks its rows
            under the planner's (cost_functions.py:107-144) and fails at the concatenation when they do not match"""
            if c.n_support_points != n_support_points or c.n_dof != n_dof:
                raise ValueError('GPMP2 extra_costs: the %s is built for n_support_points=%s, n_dof=%s; this planner has %s, %s'
                                 % (what, c.n_support_points, c.n_dof, n_support_points, n_dof))
        for c in extra_costs:
            if isinstance(c, CostCollision) and c.field is not None:
                # an extra CostCollision is one more block of collision rows with K = I / sigma_e^2: chained as a further
                # field whose share is (sigma_coll / sigma_e)^2 of the common 1 / sigma_coll^2
                _same_problem(c, 'CostCollision')
                collision_fields.append(c.field)
                scales.append((sigma_coll / c.sigma_coll) ** 2)
            elif isinstance(c, CostGP):
                _same_problem(c, 'CostGP')
                if not math.isclose(float(c.dt), float(dt), rel_tol=1e-6, abs_tol=0.0):
                    raise ValueError('GPMP2 extra_costs: the CostGP has dt=%r, the planner dt=%r' % (c.dt, dt))
                if not same(c.start_state, torch.cat((start_state, torch.zeros_like(start_state)))):
                    raise NotImplementedError('GPMP2 extra_costs: a CostGP on a start state other than the planner\'s own is not wired '
                                              'into the block solve')
                eff['start'], eff['gp'] = merge(eff['start'], c.sigma_start), merge(eff['gp'], c.sigma_gp)
            elif isinstance(c, CostGoalPrior):
                _same_problem(c, 'CostGoalPrior')
                if multi_goal_states is None or c.num_particles_per_goal != num_particles_per_goal or \
                        not same(c.multi_goal_states, torch.cat((multi_goal_states, torch.zeros_like(multi_goal_states)), dim=-1)):
                    raise NotImplementedError('GPMP2 extra_costs: a CostGoalPrior is wired into the block solve on the planner\'s own goal '
                                              'states and num_particles_per_goal only (num_samples does '
                                              'not enter its get_linear_system, cost_functions.py:538-554)')
                eff['goal'] = merge(eff['goal'], c.sigma_goal_prior)
            elif callable(getattr(c, 'get_linear_system', None)) and getattr(type(c), 'get_linear_system', None) not in (
                    Cost.get_linear_system, _TrajectoryTermCost.get_linear_system):       # (those two are the reference's bare `pass`)
                # any other cost with a linear system (the reference stacks whatever has one, cost_functions.py:107-144): its rows
                # may couple any waypoints, so the chain structure of the block solve is gone -- the planner then takes the
                # reference's own DENSE step on the device (composite get_linear_system -> _get_grad_terms -> get_torch_solve,
                # in fp64): correct, and as expensive as the reference's (N^2 words per particle)
                dense_extras.append(c)
            else:
                raise NotImplementedError('GPMP2 extra_costs: a cost without a get_linear_system of its own cannot enter the Gauss-Newton step '
                                          '(the reference fails at the unpack, cost_functions.py:122-126)')
        if not collision_fields or len(collision_fields) > 4:
            raise NotImplementedError('GPMP2 on the GPU takes one to four CollisionFields')
        solver_params = solver_params or dict(delta=1e-2, trust_region=True, method='cholesky')
        if solver_params.get('method', 'cholesky') not in SOLVE_METHODS:
            raise NotImplementedError(f"solver_params['method'] must be one of {SOLVE_METHODS}")   # gpmp2.py:457, :489
        self.robot = robot
        self.d_state_opt = 2 * n_dof
        self.sigma_start_sample, self.sigma_goal_sample = sigma_start_sample, sigma_goal_sample
        self.goal_directed = multi_goal_states is not None
        if not self.goal_directed:
            self.num_goals = 1                                                                     # gpmp2.py:136-137
        self._cost_kwargs = dict(robot=robot, n_support_points=n_support_points, dt=dt, start_state=start_state,
                                 multi_goal_states=multi_goal_states, num_particles_per_goal=num_particles_per_goal,
                                 collision_fields=collision_fields[:n_fields_own],
                                 extra_costs=extra_costs, sigma_start=sigma_start, sigma_gp=sigma_gp,
                                 sigma_coll=sigma_coll, sigma_goal_prior=sigma_goal_prior, tensor_args=tensor_args)
        self._cost = None
        self.step_size = step_size
        self.solver_params = solver_params
        self.stop_criteria = stop_criteria
        self.N = self.d_state_opt * n_support_points
        # no goal factor: sigma_goal = 0 is the C-ABI's explicit "precision 0" (include/mpb.h), not an infinite sigma
        self.sigmas = (eff['start'], eff['gp'], eff['goal'] if self.goal_directed else 0.0, sigma_coll)
        self.process_group = process_group
        self._dense_extras = dense_extras
        self.geom = ops.DeviceGeometry(robot, collision_fields, self.device, scales=scales)   # one CostCollision per field (gpmp2.py:70-78)
        self.costs = None
        self._ws = None
        self.reset(initial_particle_means=initial_particle_means)

    @property
    def cost(self):
        """The reference's GPMP2.cost (gpmp2.py:160-168): the CostComposite of the same factors, for callers that
        evaluate it or read its dense get_linear_system; the planner's own step never goes through it."""
        if self._cost is None and getattr(self, '_cost_kwargs', None) is not None:
            self._cost = build_gpmp2_cost_composite(**self._cost_kwargs)
        return self._cost

    @cost.setter
    def cost(self, value):
        self._cost = value

    def reset(self, start_state=None, multi_goal_states=None, initial_particle_means=None):
        """gpmp2.py:172-199 (a leading goal dimension of the initial means is flattened, :199)."""
        if start_state is not None:
            self.start_state = self._full_state(start_state)
        if multi_goal_states is not None:
            self.multi_goal_states = self._full_state(multi_goal_states)
        if initial_particle_means is None:
            m = self.get_random_trajs()
        else:
            m = initial_particle_means
        if m.ndim == 4:
            m = m.flatten(0, 1)
        self._particle_means = m.to(device=self.device, dtype=torch.float32).contiguous()
        B = self._particle_means.shape[0]
        assert B == self.num_particles
        dim = self.d_state_opt
        ss = self.start_state.to(device=self.device, dtype=torch.float32).reshape(-1, dim)
        self._start = ss.expand(B, dim).contiguous() if ss.shape[0] == 1 else ss.contiguous()
        if self.goal_directed:
            gs = self.multi_goal_states.to(device=self.device, dtype=torch.float32).reshape(-1, dim)
            # particles are ordered goal-major (num_goals x particles_per_goal), like the reference (:199)
            self._goal = gs.repeat_interleave(self.num_particles_per_goal, 0).contiguous() if gs.shape[0] != B else gs.contiguous()
        else:
            self._goal = self._start                 # never read: the goal factor's weight is zero
        self._ws = ops.gpmp2_workspace(B, self.n_support_points, self.n_dof, self.device)
        self.costs = torch.zeros(B, device=self.device, dtype=torch.float32)

    def set_prior_factors(self):
        """gpmp2.py:201-248: the initialisation / sampling factor objects (UnaryFactor, GPFactor) as attributes, for callers that
        read them; the planner itself samples through the structured factor (get_random_trajs) and never touches them."""
        from .costs.factors.gp_factor import GPFactor
        from .costs.factors.unary_factor import UnaryFactor
        ta = dict(device=self.device, dtype=torch.float32)
        H = self.n_support_points
        self.start_prior_init = UnaryFactor(self.d_state_opt, self.sigma_start_init, self.start_state, ta)
        self.gp_prior_init = GPFactor(self.n_dof, self.sigma_gp_init, self.dt, H - 1, ta)
        if self.goal_directed:
            self.multi_goal_prior_init = [UnaryFactor(self.d_state_opt, self.sigma_goal_init, self.multi_goal_states[i], ta)
                                          for i in range(self.num_goals)]
        self.start_prior_sample = UnaryFactor(self.d_state_opt, self.sigma_start_sample, self.start_state, ta)
        if self.goal_directed:
            self.multi_goal_prior_sample = [UnaryFactor(self.d_state_opt, self.sigma_goal_sample, self.multi_goal_states[i], ta)
                                            for i in range(self.num_goals)]

    def get_dist(self, start_K, gp_K, goal_K, state_init, particle_means=None, goal_states=None):
        """gpmp2.py:250-271: MultiMPPrior over the planner's horizon (2 n_dof states per support point)."""
        from .costs.factors.mp_priors_multi import MultiMPPrior
        return MultiMPPrior(self.n_support_points - 1, self.dt, 2 * self.n_dof, self.n_dof, start_K, gp_K, state_init,
                            K_g_inv=goal_K, means=particle_means, goal_states=goal_states,
                            tensor_args=dict(device=self.device, dtype=torch.float32))

    def set_problem_states(self, starts, goals):
        """Per-particle start / goal positions (B,D): thousands of independent problems in one planner."""
        z = torch.zeros_like(starts)
        self._start = torch.cat([starts, z], -1).to(device=self.device, dtype=torch.float32).contiguous()
        self._goal = torch.cat([goals, z], -1).to(device=self.device, dtype=torch.float32).contiguous()

    def _step(self):
        H, D, B = self.n_support_points, self.n_dof, self.num_particles
        delta = self.solver_params['delta']
        trust = self.solver_params.get('trust_region', False)
        x = self._particle_means
        if self._dense_extras:
            # gpmp2.py:308-342 as the reference runs it: dense (A, b, K) of every member, normal equations, dense solve (fp64)
            # (dense-sized: refused up front when the three (B, N, N)-class fp64 operands cannot fit, instead of dying in the allocator)
            N = H * 2 * D
            need = 4 * B * N * N * 8
            free = torch.cuda.mem_get_info(self.device)[0] if self.device.type == 'cuda' else need
            if need > free:
                raise MemoryError(f'GPMP2 with a dense extra cost needs ~{need / 2 ** 30:.1f} GiB for B={B}, N={N} (the reference\'s dense '
                                  f'step, gpmp2.py:355-368); {free / 2 ** 30:.1f} GiB free -- lower num_particles or drop the extra cost')
            A, b, K = (t.double() for t in self.cost.get_linear_system(x, n_interpolated_points=self.n_interpolated_points))
            sharded = self.process_group is not None and dist.get_world_size(self.process_group) > 1
            if trust and sharded:
                # quirk Q9 under sharding (gpmp2.py:361-367): the damping is the mean of diag(A^T K A) over the GLOBAL batch --
                # all-reduce the local sum of the diagonals and the local particle count, as the block path does
                A_t_K = A.transpose(-2, -1) @ K
                A_t_A = A_t_K @ A
                dsum = torch.diagonal(A_t_A, dim1=-2, dim2=-1).sum(0)
                nb = torch.tensor([float(B)], device=dsum.device, dtype=torch.float64)
                dist.all_reduce(dsum, group=self.process_group)
                dist.all_reduce(nb, group=self.process_group)
                JtJ, g = A_t_A + delta * torch.diag(dsum / nb), A_t_K @ b
            else:
                JtJ, g = self._get_grad_terms(A, b, K, delta=delta, trust_region=trust)
            d_theta = self.get_torch_solve(JtJ, g, method=self.solver_params.get('method', 'cholesky')).reshape(B, H, 2 * D)
            self.costs = (b.transpose(1, 2) @ K @ b).reshape(B).to(torch.float32)
            x.add_((self.step_size * d_theta).to(x.dtype))
            return
        if trust and self.process_group is not None and dist.get_world_size(self.process_group) > 1:
            ops.gpmp2_linearize(x, self.geom, self._ws, n_interp=self.n_interpolated_points)
            dsum = ops.gpmp2_diag(self._ws, B, H, D, self.sigmas, self.dt, n_fields=self.geom.n_fields)
            nb = torch.tensor([float(B)], device=self.device, dtype=torch.float64)
            dist.all_reduce(dsum, group=self.process_group)     # one H*2D fp64 vector per iteration
            dist.all_reduce(nb, group=self.process_group)
            ops.gpmp2_solve(x, self._start, self._goal, dsum / nb, self._ws, self.sigmas, self.dt, delta, True,
                            self.step_size, costs_out=self.costs, n_fields=self.geom.n_fields)
        else:
            ops.gpmp2_step(x, self._start, self._goal, self.geom, self._ws, self.sigmas, self.dt, delta, trust,
                           self.step_size, n_iters=1, costs_out=self.costs, n_interp=self.n_interpolated_points)

    # ---- the reference's dense building blocks, for subclasses / callers that use them directly ------------------------
    def _get_grad_terms(self, A, b, K, delta=0., trust_region=False, sparse_computation=False,
                        sparse_computation_block_diag=False):
        """gpmp2.py:344-368 (dense branch): J^T J = A^T K A + delta I (or + delta * mean_b(A^T K A) o I with the trust region,
        quirk Q9) and g = A^T K b from a DENSE (A, b, K) -- e.g. self.cost.get_linear_system(x) -- on the tensors' device.
        The planner's own step never forms these (mpb_gpmp2_step assembles the same normal equations block-tridiagonally,
        in registers); this is the reference's method for code that calls or overrides it.  The sparse_computation branches
        of the reference build the same two tensors another way and are served by this one."""
        N = A.shape[-1]
        I = torch.eye(N, N, device=A.device, dtype=A.dtype)
        A_t_K = A.transpose(-2, -1) @ K
        A_t_A = A_t_K @ A
        J_t_J = A_t_A + delta * (A_t_A.mean(0) * I if trust_region else I)
        return J_t_J, A_t_K @ b

    def get_torch_solve(self, A, b, method):
        """gpmp2.py:432-491: d_theta = A^-1 b for dense batched SPD A (B,N,N), b (B,N,1) by the named dense method, on the
        tensors' device ('cholesky-sparse' raises like the reference, :457).  The planner's step solves the same system by
        block elimination inside mpb_gpmp2_solve and does not come through here."""
        if method == 'inverse':
            return torch.linalg.solve(A, b)
        if method == 'cholesky':
            l, _ = torch.linalg.cholesky_ex(A)
            return torch.cholesky_solve(b, l)
        if method == 'lstq':
            return torch.linalg.lstsq(A, b)[0]
        raise NotImplementedError(method)

    def optimize(self, opt_iters=None, debug=False, **observation):
        """gpmp2.py:273-306 incl. the optional relative-change stop criterion (:286-293)."""
        if opt_iters is None:
            opt_iters = self.opt_iters
        costs_previous = None
        for opt_step in range(opt_iters):
            self._step()
            if self.stop_criteria is not None:
                costs = self.costs.clone()
                if opt_step > 0 and torch.all(torch.abs((costs - costs_previous) / costs) < self.stop_criteria):
                    break
                costs_previous = costs
        self._recent_state_trajectories = self._particle_means[..., :self.n_dof].clone()
        self._recent_control_particles = self._particle_means[..., -self.n_dof:].clone()
        return self._get_traj()

    def get_recent_samples(self):
        m = self.num_goals
        pos = self._recent_state_trajectories.detach().clone()
        vel = self._recent_control_particles.detach().clone()
        return pos.reshape(m, -1, *pos.shape[1:]), vel.reshape(m, -1, *vel.shape[1:])
