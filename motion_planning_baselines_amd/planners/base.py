"""Planner base classes with the reference's surface (mp_baselines/planners/base.py).

``MPPlanner`` (base.py:12-57) and ``OptimizationPlanner`` (base.py:60-226): shape bookkeeping, the
``get_traj`` / ``_get_costs`` contract.  The per-iteration arithmetic lives in the HIP library; this
file is host-side bookkeeping only.
"""
import abc
from abc import ABC, abstractmethod

import torch

from .._lib import MPBError


def require_cuda(tensor_args):
    """The product has no CPU path: planners must be placed on a GPU."""
    dev = torch.device((tensor_args or {}).get('device', 'cpu'))
    if dev.type != 'cuda':
        raise MPBError(f"tensor_args['device'] must be a CUDA/ROCm device (got {dev}); there is no CPU fallback")
    if not torch.cuda.is_available():
        raise MPBError('no GPU visible to PyTorch-ROCm; the HIP kernels cannot run')
    if (tensor_args or {}).get('dtype', torch.float32) != torch.float32:
        raise MPBError('the HIP planners store trajectories in float32')
    return dev


def finite_difference_vector(x, dt=1.0):
    """Central finite difference along the horizon axis, zero at both ends (build-defined stand-in for
    torch_robotics.trajectory.utils.finite_difference_vector, called at base.py:211)."""
    out = torch.zeros_like(x)
    out[..., 1:-1, :] = (x[..., 2:, :] - x[..., :-2, :]) / (2 * dt)
    return out


class MPPlanner(ABC):
    """base.py:12-57."""

    def __init__(self, name=None, tensor_args=None, **kwargs):
        self.name = name
        self.tensor_args = tensor_args
        self._kwargs = kwargs

    @abstractmethod
    def optimize(self, opt_iters=1, **observation):
        pass

    def __call__(self, opt_iters=1, **observation):
        return self.optimize(opt_iters, **observation)

    def __repr__(self):
        return f"{self.name}({self._kwargs})"

    @abc.abstractmethod
    def render(self, ax, **kwargs):
        raise NotImplementedError


class OptimizationPlanner(MPPlanner):
    """base.py:60-226 (ctor bookkeeping :82-113; _get_traj :204-213; _get_costs :218-223)."""

    def __init__(self, name='OptimizationPlanner', n_dof=None, n_support_points=None, n_interpolated_points=None,
                 num_particles_per_goal=None, opt_iters=None, dt=None, start_state=None, cost=None,
                 initial_particle_means=None, multi_goal_states=None, sigma_start_init=0.001,
                 sigma_goal_init=0.001, sigma_gp_init=10., pos_only=False, tensor_args=None, **kwargs):
        super().__init__(name, tensor_args, **kwargs)
        self.device = require_cuda(tensor_args)
        self.n_dof = n_dof
        self.dim = 2 * n_dof
        self.n_support_points = n_support_points
        self.n_interpolated_points = n_interpolated_points
        self.num_particles_per_goal = num_particles_per_goal
        self.opt_iters = opt_iters
        self.dt = dt
        self.pos_only = pos_only
        self.start_state = start_state
        self.multi_goal_states = multi_goal_states
        if multi_goal_states is None:
            self.num_goals = 1
        else:
            assert multi_goal_states.ndim == 2
            self.num_goals = multi_goal_states.shape[0]
        self.num_particles = self.num_goals * self.num_particles_per_goal
        self.cost = cost
        self.initial_particle_means = initial_particle_means
        self._particle_means = None
        if pos_only:
            self.d_state_opt = n_dof
        else:
            self.d_state_opt = 2 * n_dof
            self.start_state = torch.cat([start_state, torch.zeros_like(start_state)], dim=-1)
            if multi_goal_states is not None:
                self.multi_goal_states = torch.cat([multi_goal_states, torch.zeros_like(multi_goal_states)], dim=-1)
        self.sigma_start_init = sigma_start_init
        self.sigma_goal_init = sigma_goal_init
        self.sigma_gp_init = sigma_gp_init

    def get_random_trajs(self):
        """Initial particles from the GP prior (base.py:155-202).  SURVEY.md 8(f) rank 1 ("next"): the
        native block-tridiagonal sampler is not built yet -- pass ``initial_particle_means``."""
        raise NotImplementedError(
            'GP-prior initial sampling (base.py:155-202) is outside the hot path built so far; '
            'pass initial_particle_means (e.g. workloads.straight_line_means)')

    def _get_traj(self):
        trajs = self._particle_means.clone()
        if self.pos_only:
            vels = finite_difference_vector(trajs, dt=self.dt)
            trajs = torch.cat((trajs, vels), dim=1)   # quirk Q12: concatenated along the horizon axis
        return trajs

    def get_traj(self):
        return self._get_traj()

    def _get_costs(self, state_trajectories, **observation):
        if self.cost is None:
            return torch.zeros(state_trajectories.shape[0], device=state_trajectories.device)
        return self.cost(state_trajectories, **observation)

    def render(self, ax, **kwargs):
        raise NotImplementedError
