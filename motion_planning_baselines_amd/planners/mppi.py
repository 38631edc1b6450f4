"""MPPI with the reference's class surface (mp_baselines/planners/mppi.py); iterations on the GPU.

Host side: the per-control-dimension covariance priors (priors/gaussian.py:143-198), their Cholesky
factors and inverses, computed with the same torch calls as the reference.  Device side:
mpb_mppi_step (csrc/mpb_mppi.hip) runs sampling, rollout, cost, importance term, softmax and mean update.
"""
import numpy as np
import torch

from .. import ops
from .base import MPPlanner, require_cuda
from .costs.cost_functions import fusable_collision


from .dynamics.point import PointParticleDynamics  # noqa: F401  (import path of the reference kept: planners.dynamics.point)
from .priors.gaussian import (ControlTrajectoryGaussian, check_Cov_is_valid, const_ctrl_Cov, diag_Cov,  # noqa: F401
                              get_multivar_gaussian_prior)


BEST_COST_NONE = 3.0e38     # "no sample seen yet" of the device-side best-cost tracker


class MPPI(MPPlanner):
    """Drop-in for mp_baselines.planners.mppi.MPPI (ctor kwargs mppi.py:8-21).

    ``optimize(state=..., goal_state=..., cost=...)`` as in the reference (mppi.py:136-162).  Extra kwargs:
    noise 'philox' | 'torch_cpu' | 'torch' and seed, as for STOMP.
    """

    def __init__(self, system, num_ctrl_samples, rollout_steps, opt_iters, control_std=None, initial_mean=None,
                 step_size=1., temp=1., cov_prior_type='indep_ctrl', tensor_args=None, noise='philox', seed=0,
                 **kwargs):
        super().__init__(name='MPPI', tensor_args=tensor_args)
        self.device = require_cuda(tensor_args)
        assert cov_prior_type in ('indep_ctrl', 'const_ctrl')
        self.system = system
        self.state_dim = system.state_dim
        self.control_dim = system.control_dim
        self.rollout_steps = rollout_steps
        self.num_ctrl_samples = num_ctrl_samples
        self.opt_iters = opt_iters
        self.step_size = step_size
        self.temp = temp
        self.control_std = control_std
        self.cov_prior_type = cov_prior_type
        self.noise = noise
        self.seed = int(seed)
        self._iter = 0
        cpu = dict(device='cpu', dtype=torch.float32)
        gen = const_ctrl_Cov if cov_prior_type == 'const_ctrl' else diag_Cov
        Cov = gen(control_std, rollout_steps, self.control_dim, cpu)
        self.Cov = Cov.to(self.device)
        # MultivariateNormal(covariance_matrix=C).scale_tril == cholesky(C); Cov_inv as mppi.py:49-52
        self._scale_tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(self.control_dim)]).to(self.device).contiguous()
        self.Cov_inv = torch.stack([Cov[..., i].inverse() for i in range(self.control_dim)]).to(self.device).contiguous()
        T, c, S = rollout_steps, self.control_dim, num_ctrl_samples
        f = lambda a: torch.as_tensor(a, dtype=torch.float32).to(self.device).contiguous()
        self._cmin, self._cmax = f(system.ctrl_min), f(system.ctrl_max)
        self._disc = f(system.discount_seq)
        cw = system._c_weights
        self._cw = f([cw['pos'], cw['vel'], cw['ctrl'], cw['pos_T']])
        self._controls = torch.empty(1, S, T, c, device=self.device)
        self._states = torch.empty(1, S, T, c, device=self.device)
        self._costs = torch.empty(1, S, device=self.device)
        self._weights = torch.empty(1, S, device=self.device)
        # MPPI._save_best state (mppi.py:164-168): kept on the device, updated inside the kernel every iteration
        self._best_cost = torch.full((1,), BEST_COST_NONE, device=self.device)   # finite sentinel: the library is built with -ffinite-math-only
        self._best_traj = torch.zeros(1, T, c, device=self.device)
        self.weights = None
        self.reset(initial_mean=initial_mean)

    def reset(self, initial_mean=None):
        T, c = self.rollout_steps, self.control_dim
        if initial_mean is not None:
            self._mean = initial_mean.clone().to(device=self.device, dtype=torch.float32).reshape(T, c).contiguous()
        else:
            self._mean = torch.zeros(T, c, device=self.device, dtype=torch.float32)

    def _draw_eps(self, n_iters):
        if self.noise == 'philox':
            return None
        S, T, c = self.num_ctrl_samples, self.rollout_steps, self.control_dim
        dev = 'cpu' if self.noise == 'torch_cpu' else self.device
        # the reference draws one (S,T) block per control dimension per iteration (gaussian.py:291-297)
        blocks = [torch.stack([torch.empty(S, T, device=dev).normal_() for _ in range(c)]) for _ in range(n_iters)]
        return torch.stack(blocks).reshape(n_iters, 1, c, S, T).to(self.device).contiguous()

    def _problem(self, **observation):
        """Device-side (state, goal, geometry, k_sigma, weight) of one optimize / rollout call."""
        c = self.control_dim
        state = observation['state'].to(device=self.device, dtype=torch.float32).reshape(1, c).contiguous()
        goal = observation.get('goal_state', self.system.goal_state)
        goal = goal.to(device=self.device, dtype=torch.float32)[..., :c].reshape(1, c).contiguous()
        cost = observation.get('cost', None)
        geom, k_sigma, weight = None, 0.0, 1.0
        if cost is not None:
            fused = fusable_collision(cost)
            if fused is not None:          # (any other cost object: _energy_shift, outside the kernel)
                cc, weight = fused
                geom, k_sigma = cc.device_geometry(self.device), cc.k_sigma
        return state, goal, geom, k_sigma, weight

    @property
    def ctrl_dist(self):
        """The sampling distribution as the reference's object (mppi.py:42-49: `planner.ctrl_dist.Cov`, `.sample(n)`), its means
        synchronised with the planner's current mean on access (the one-launch loop samples inside the kernel and does
        not go through this object)."""
        if getattr(self, '_ctrl_dist', None) is None:
            self._ctrl_dist = ControlTrajectoryGaussian(self.rollout_steps, self.control_dim, self._mean, self.Cov,
                                                        tensor_args=dict(device=self.device, dtype=torch.float32), seed=self.seed)
        self._ctrl_dist.update_means(self._mean)
        return self._ctrl_dist

    @property
    def best_cost(self):
        """Lowest sample cost seen by optimize() so far (inf before the first call), as a 0-dim tensor."""
        b = self._best_cost[0]
        return torch.where(b >= BEST_COST_NONE, torch.full_like(b, float('inf')), b)

    @property
    def best_traj(self):
        """State trajectory (T, state_dim) of that sample."""
        return self._best_traj[0]

    def _launch(self, n_iters, step_size, track_best=False, **observation):
        state, goal, geom, k_sigma, weight = self._problem(**observation)
        mean = self._mean.reshape(1, self.rollout_steps, self.control_dim)
        ops.mppi_step(mean, self._draw_eps(n_iters), self._scale_tril, self.Cov_inv, state, goal, self._cmin,
                      self._cmax, self._disc, self._cw, geom, self._controls, self._states, self._costs, self._weights,
                      self.system.dt, k_sigma=k_sigma, weight=weight, temp=self.temp, step_size=step_size,
                      n_iters=n_iters, seed=self.seed, iter0=self._iter,
                      best_cost=self._best_cost if track_best else None,
                      best_states=self._best_traj if track_best else None)
        self._iter += n_iters
        self.costs = self._costs.reshape(-1, 1)
        self.state_trajectories = self._states[0]

    def update_ctrl_dist(self):
        """mppi.py:68-70: the sampling distribution reads self._mean directly here; nothing to refresh."""

    @staticmethod
    def _generic_cost(observation):
        """The caller's cost object when it is NOT the single collision cost the kernel fuses (else None)."""
        cost = observation.get('cost', None)
        return cost if (cost is not None and fusable_collision(cost) is None) else None

    def _energy_shift(self, cost):
        """point.py:191-196: `cost.eval(cat(X, U)).sum(-1)` -- the per-rollout costs of ANY cost object summed into ONE
        scalar (quirk Q6) that is added to every sample's cost: a shift the softmax weights do not see, but `costs` and
        the best-sample tracking do.  Evaluated on the rollouts of the last kernel iteration."""
        full_traj = torch.cat((self._states[0], self._controls[0]), dim=-1)
        e = cost.eval(full_traj)
        return e.sum(-1).to(device=self.device, dtype=torch.float32)

    def sample_and_eval(self, **observation):
        """mppi.py:88-134: sample controls, roll out, evaluate costs (incl. the importance term); the mean is left
        untouched (one kernel iteration with a zero step)."""
        self._launch(1, 0.0, **observation)
        generic = self._generic_cost(observation)
        if generic is not None:
            self._costs += self._energy_shift(generic)
        return self._controls[0], self._states[0], self.costs

    def update_controller(self, costs, U_sampled):
        """mppi.py:72-86: softmax weights over the samples and mean += step * sum_s w_s (U_s - mean) -- the STOMP
        update kernel without a covariance product."""
        S, T, c = self.num_ctrl_samples, self.rollout_steps, self.control_dim
        ops.stomp_update(self._mean.reshape(1, T, c), U_sampled.reshape(1, S, T, c).contiguous(),
                         costs.reshape(1, S).to(torch.float32).contiguous(), self._weights, None, self.step_size, self.temp)
        self.weights = self._weights.reshape(-1, 1).clone()
        self.update_ctrl_dist()

    def get_state_trajectories_rollout(self, controls=None, num_ctrl_samples=None, **observation):
        """mppi.py:190-210: Euler rollout of the given control sequences (default: the mean) from observation['state'].
        Every sequence becomes a one-sample problem with zero noise, so the same kernel serves it."""
        T, c = self.rollout_steps, self.control_dim
        U = (self._mean.unsqueeze(0) if controls is None else controls).to(device=self.device, dtype=torch.float32)
        U = U.reshape(-1, T, c).contiguous().clone()
        n = U.shape[0]
        state, goal, _, _, _ = self._problem(**{k: v for k, v in observation.items() if k != 'cost'})
        eps = torch.zeros(1, n, c, 1, T, device=self.device)
        ctr, st = torch.empty(n, 1, T, c, device=self.device), torch.empty(n, 1, T, c, device=self.device)
        cs, ws = torch.empty(n, 1, device=self.device), torch.empty(n, 1, device=self.device)
        ops.mppi_step(U, eps, self._scale_tril, self.Cov_inv, state.expand(n, c).contiguous(), goal.expand(n, c).contiguous(),
                      self._cmin, self._cmax, self._disc, self._cw, None, ctr, st, cs, ws, self.system.dt, temp=self.temp,
                      step_size=0.0, n_iters=1)
        return st[:, 0]

    def optimize(self, opt_iters=None, **observation):
        if opt_iters is None:
            opt_iters = self.opt_iters
        if self._generic_cost(observation) is not None:
            # any other Cost / CostComposite (point.py:191-196): its scalar shift differs from iteration to iteration,
            # so the loop of mppi.py:145-152 runs iteration by iteration -- kernel (sample, rollout, costs) -> the caller's
            # cost on the rollouts -> _save_best -> update kernel
            for _ in range(opt_iters):
                _, _, costs = self.sample_and_eval(**observation)
                self._save_best()
                self.update_controller(costs, self._controls[0])
            self._recent_control_samples = self._controls[0]
            self._recent_state_trajectories = self._states[0]
            self._recent_weights = self.weights
            return self._controls[0], self._states[0], self.costs
        # _save_best runs inside the kernel after every iteration's sample_and_eval, as in mppi.py:145-152
        self._launch(opt_iters, self.step_size, track_best=True, **observation)
        self.weights = self._weights.reshape(-1, 1)
        self._recent_control_samples = self._controls[0]
        self._recent_state_trajectories = self._states[0]
        self._recent_weights = self.weights
        return self._controls[0], self._states[0], self.costs

    def _save_best(self):
        """mppi.py:164-168 for callers that drive sample_and_eval / update_controller themselves."""
        best_cost = torch.min(self.costs)
        if best_cost < self._best_cost[0]:
            self._best_cost[0] = best_cost
            self._best_traj[0] = self.state_trajectories[torch.argmin(self.costs)]

    def pop(self):
        action = self._mean[0, :].clone().detach()
        self.shift()
        return action

    def shift(self):
        """mppi.py:176-178 -- quirk Q7: the roll is along the control axis, reproduced as is."""
        self._mean = self._mean.roll(shifts=-1, dims=-1).contiguous()
        self._mean[-1:] = 0.

    def get_recent_samples(self):
        return (self._recent_control_samples.detach().clone(), self._recent_state_trajectories.detach().clone(),
                self._recent_weights.detach().clone())

    def get_mean_controls(self):
        return self._mean

    def render(self, ax, **kwargs):
        raise NotImplementedError
