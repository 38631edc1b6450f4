"""StochGPMP with the reference's class surface (mp_baselines/planners/stoch_gpmp.py), iterations on the GPU.

Per iteration (stoch_gpmp.py:289-298): sample S trajectories per particle from the full GP prior around the
particle mean, evaluate the composite cost + importance term, softmax, mean += step * sum_s w_s (x_s - mean).
The reference materialises a dense N x N precision / scale_tril (N = 2D*H); here the prior's
(2x2) (x) I_D block-bidiagonal factor is used for sampling (csrc/mpb_prior.hip) and the importance term is
evaluated factor-wise (csrc/mpb_stoch_gpmp.hip).
"""
import torch

from .. import ops
from .base import OptimizationPlanner, gp_prior_factor, gp_prior_scale_tril


class StochGPMP(OptimizationPlanner):
    """Drop-in for mp_baselines.planners.stoch_gpmp.StochGPMP (ctor kwargs stoch_gpmp.py:16-38 plus the cost
    kwargs of build_gpmp2_cost_composite: collision_fields (one to four), sigma_start, sigma_gp, sigma_coll,
    sigma_goal_prior).  Extra kwargs: noise 'torch_cpu' | 'philox', seed."""

    def __init__(self, robot=None, n_dof=None, n_support_points=None, num_particles_per_goal=None, opt_iters=None,
                 dt=None, start_state=None, step_size=1., multi_goal_states=None, initial_particle_means=None,
                 sigma_start_init=None, sigma_start_sample=None, sigma_goal_init=None, sigma_goal_sample=None,
                 sigma_gp_init=None, sigma_gp_sample=None, num_samples=2, temperature=1., collision_fields=None,
                 sigma_start=1e-5, sigma_gp=1e-2, sigma_coll=1e-5, sigma_goal_prior=1e-5, tensor_args=None,
                 noise='torch_cpu', seed=0, **kwargs):
        super().__init__(name='StochGPMP', n_dof=n_dof, n_support_points=n_support_points,
                         num_particles_per_goal=num_particles_per_goal, opt_iters=opt_iters, dt=dt,
                         start_state=start_state, initial_particle_means=initial_particle_means,
                         multi_goal_states=multi_goal_states, sigma_start_init=sigma_start_init,
                         sigma_goal_init=sigma_goal_init, sigma_gp_init=sigma_gp_init, pos_only=False,
                         tensor_args=tensor_args)
        if not collision_fields or len(collision_fields) > 4:
            raise NotImplementedError('StochGPMP on the GPU takes one to four CollisionFields')
        assert multi_goal_states is not None, 'StochGPMP kernels need goal states'
        self.robot = robot
        self.d_state_opt = 2 * n_dof
        self.num_samples = num_samples
        self.step_size = step_size
        self.temperature = temperature
        self.sig_sample = (sigma_start_sample, sigma_gp_sample, sigma_goal_sample)
        self.sig_cost = (sigma_start, sigma_gp, sigma_goal_prior, sigma_coll)
        self.noise, self.seed, self._iter = noise, int(seed), 0
        self.geom = ops.DeviceGeometry(robot, list(collision_fields), self.device)
        H, D = n_support_points, n_dof
        Ud, Uo = gp_prior_factor(H, dt, sigma_start_sample, sigma_gp_sample, sigma_goal_sample)
        f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).to(self.device).contiguous()
        self._Ud, self._Uo = f64(Ud), f64(Uo)
        self._tril = f64(gp_prior_scale_tril(Ud, Uo)) if H <= 128 else None   # per-dof scale_tril: sampling as an MFMA GEMM
        self._weights = None
        self.reset(initial_particle_means=initial_particle_means)

    def set_prior_factors(self):
        """stoch_gpmp.py:149-198: the initialisation / sampling factor objects (UnaryFactor, GPFactor) as attributes, for callers that
        read them; the planner itself samples through the structured factor (get_random_trajs) and never touches them."""
        from .costs.factors.gp_factor import GPFactor
        from .costs.factors.unary_factor import UnaryFactor
        ta = dict(device=self.device, dtype=torch.float32)
        H = self.n_support_points
        self.start_prior_init = UnaryFactor(self.d_state_opt, self.sigma_start_init, self.start_state, ta)
        self.gp_prior_init = GPFactor(self.n_dof, self.sigma_gp_init, self.dt, H - 1, ta)
        if True:
            self.multi_goal_prior_init = [UnaryFactor(self.d_state_opt, self.sigma_goal_init, self.multi_goal_states[i], ta)
                                          for i in range(self.num_goals)]
        self.start_prior_sample = UnaryFactor(self.d_state_opt, self.sig_sample[0], self.start_state, ta)
        self.gp_prior_sample = GPFactor(self.n_dof, self.sig_sample[1], self.dt, H - 1, ta)
        self.multi_goal_prior_sample = [UnaryFactor(self.d_state_opt, self.sig_sample[2], self.multi_goal_states[i], ta)
                                        for i in range(self.num_goals)]

    def get_prior_dist(self, start_K, gp_K, goal_K, state_init, particle_means=None, goal_states=None):
        """stoch_gpmp.py:212-233: MultiMPPrior over the planner's horizon (2 n_dof states per support point)."""
        from .costs.factors.mp_priors_multi import MultiMPPrior
        return MultiMPPrior(self.n_support_points - 1, self.dt, 2 * self.n_dof, self.n_dof, start_K, gp_K, state_init,
                            K_g_inv=goal_K, means=particle_means, goal_states=goal_states,
                            tensor_args=dict(device=self.device, dtype=torch.float32))

    def reset(self, start_state=None, multi_goal_states=None, initial_particle_means=None):
        """stoch_gpmp.py:97-141: new start / goal states re-target the planner (:99-103)."""
        if start_state is not None:
            self.start_state = self._full_state(start_state)
        if multi_goal_states is not None:
            self.multi_goal_states = self._full_state(multi_goal_states)
        if initial_particle_means is None:
            m = self.get_random_trajs()
        elif isinstance(initial_particle_means, str) and initial_particle_means == 'const_vel':
            # stoch_gpmp.py:107-111 -> :197-215: one straight line per goal (velocity channel (goal - start) / (H dt),
            # the reference's own denominator), repeated for the goal's particles: (G, ppg, H, 2D)
            lines = self.const_vel_trajectories(self.start_state, self.multi_goal_states)
            m = lines.unsqueeze(1).expand(-1, self.num_particles_per_goal, -1, -1)
        else:
            m = initial_particle_means
        if m.ndim == 4:
            m = m.flatten(0, 1)
        self._particle_means = m.to(device=self.device, dtype=torch.float32).contiguous()
        P, dim = self.num_particles, self.d_state_opt
        assert self._particle_means.shape[0] == P
        ss = self.start_state.to(device=self.device, dtype=torch.float32).reshape(-1, dim)
        self._start = ss.expand(P, dim).contiguous()
        gs = self.multi_goal_states.to(device=self.device, dtype=torch.float32).reshape(-1, dim)
        self._goal = gs.repeat_interleave(self.num_particles_per_goal, 0).contiguous()
        self.costs = torch.zeros(P, self.num_samples, device=self.device, dtype=torch.float32)
        self._weights_buf = torch.empty(P, self.num_samples, device=self.device, dtype=torch.float32)
        self.state_samples = self._sample()      # the reference samples once in reset (:141)

    def _sample(self):
        P, S, H, dim = self.num_particles, self.num_samples, self.n_support_points, self.d_state_opt
        eps = None
        if self.noise != 'philox':   # MultivariateNormal.sample((S,)) of batch shape (P,) and event shape (M,)
            eps = torch.empty(S, P, H * dim, dtype=torch.float64).normal_().to(self.device)
        # persistent buffers: no allocator traffic inside the iteration loop
        if getattr(self, '_means64', None) is None or self._means64.shape != self._particle_means.shape:
            self._means64 = torch.empty_like(self._particle_means, dtype=torch.float64)
            self._samples_buf = torch.empty(P * S, H, dim, device=self.device, dtype=torch.float32)
        self._means64.copy_(self._particle_means)
        out = ops.gp_prior_sample(self._means64, eps, self._Ud, self._Uo, S, self.n_dof, seed=self.seed + self._iter,
                                  scale_tril=self._tril, out=self._samples_buf)
        self._iter += 1
        return out.reshape(P, S, H, dim)

    def sample_trajectories(self, **kwargs):
        """stoch_gpmp.py:229-233: S fresh trajectories per particle from the sampling prior, (P, S, H, 2D)."""
        self.state_samples = self._sample()
        return self.state_samples

    def sample_and_eval(self, **observation):
        """stoch_gpmp.py:244-265: draw the samples and evaluate cost + importance term -> (costs (P,S), samples)."""
        P, S, H, dim = self.num_particles, self.num_samples, self.n_support_points, self.d_state_opt
        self.state_samples = self._sample()
        flat = self.state_samples.reshape(P * S, H, dim)
        ops.stoch_gpmp_costs(flat, self._particle_means, self._start, self._goal, self.geom, self.costs, S,
                             self.sig_cost, self.sig_sample, self.dt, self.temperature)
        return self.costs, self.state_samples

    def _update_distribution(self, costs, traj_samples):
        """stoch_gpmp.py:267-279: softmax weights over the samples, mean += step * sum_s w_s (x_s - mean)."""
        ops.stomp_update(self._particle_means, traj_samples, costs, self._weights_buf, None, self.step_size,
                         self.temperature)
        self._weights = self._weights_buf.reshape(self.num_particles, self.num_samples, 1, 1)
        self._recent_weights = self._weights

    def optimize(self, opt_iters=None, debug=False, **observation):
        """stoch_gpmp.py:281-313."""
        if opt_iters is None:
            opt_iters = self.opt_iters
        if self.noise == 'philox' and opt_iters > 0:
            # device noise: the whole loop is one C call (mpb_stoch_gpmp_step) -- no per-iteration host work
            P, S, H, dim = self.num_particles, self.num_samples, self.n_support_points, self.d_state_opt
            if getattr(self, '_means64', None) is None or self._means64.shape != self._particle_means.shape:
                self._means64 = torch.empty_like(self._particle_means, dtype=torch.float64)
                self._samples_buf = torch.empty(P * S, H, dim, device=self.device, dtype=torch.float32)
            ops.stoch_gpmp_step(self._particle_means, self._means64, self._samples_buf, self.costs, self._weights_buf,
                                self._Ud, self._Uo, self._tril, self._start, self._goal, self.geom, S, self.sig_cost,
                                self.sig_sample, self.dt, self.temperature, self.step_size, n_iters=opt_iters,
                                seed=self.seed + self._iter)
            self._iter += opt_iters
            self.state_samples = self._samples_buf.reshape(P, S, H, dim)
            self._weights = self._weights_buf.reshape(P, S, 1, 1)
            self._recent_weights = self._weights
            return self._get_traj()
        for _ in range(opt_iters):
            costs, samples = self.sample_and_eval(**observation)
            self._update_distribution(costs, samples)
        return self._get_traj()

    def get_recent_samples(self):
        D = self.n_dof
        return (self.state_samples[..., :D].clone(), self._particle_means[..., :D].clone(),
                self.state_samples[..., -D:].clone(), self._particle_means[..., -D:].clone(), self._weights.clone())
