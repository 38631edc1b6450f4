"""Planner base classes with the reference's surface (mp_baselines/planners/base.py).

``MPPlanner`` (base.py:12-57) and ``OptimizationPlanner`` (base.py:60-226): shape bookkeeping, the
``get_traj`` / ``_get_costs`` contract.  The per-iteration arithmetic lives in the HIP library; this
file is host-side bookkeeping only.
"""
import abc
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import ops
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


def gp_prior_factor(H, dt, sigma_start, sigma_gp, sigma_goal=None):
    """Block upper-bidiagonal factor U of the GP-prior precision K^-1 = A^T Q^-1 A = U U^T
    (mp_priors_multi.py:213-251), in its (2x2) (x) I_D form, fp64 on the host.

    K^-1 diagonal 2x2 blocks: [t=0] K_s + Phi^T Qi Phi, [0<t<H-1] Qi + Phi^T Qi Phi, [t=H-1] Qi (+ K_g);
    off-diagonal (t,t+1): -Phi^T Qi  (gp_factor.py:34-50).  Returns Udiag (H,3) = (u00,u01,u11), Uoff (H-1,4).
    """
    k = 1.0 / sigma_gp ** 2
    Qi = np.array([[12.0 / dt ** 3, -6.0 / dt ** 2], [-6.0 / dt ** 2, 4.0 / dt]]) * k
    Phi = np.array([[1.0, dt], [0.0, 1.0]])
    PQP = Phi.T @ Qi @ Phi
    off = -Phi.T @ Qi
    Ud = np.zeros((H, 3))
    Uo = np.zeros((max(H - 1, 0), 4))
    Unext = None
    for t in range(H - 1, -1, -1):
        Pt = np.zeros((2, 2))
        if t == 0:
            Pt += np.eye(2) / sigma_start ** 2
        if t < H - 1:
            Pt += PQP
        if t > 0:
            Pt += Qi
        if t == H - 1 and sigma_goal is not None:
            Pt += np.eye(2) / sigma_goal ** 2
        if t < H - 1:
            O = off @ np.linalg.inv(Unext).T          # U_{t,t+1} = P_{t,t+1} U_{t+1,t+1}^-T
            Pt = Pt - O @ O.T
            Uo[t] = O.reshape(-1)
        u11 = np.sqrt(Pt[1, 1])                        # "upper Cholesky": Pt = U U^T, U upper triangular
        u01 = Pt[0, 1] / u11
        u00 = np.sqrt(Pt[0, 0] - u01 ** 2)
        Unext = np.array([[u00, u01], [0.0, u11]])
        Ud[t] = (u00, u01, u11)
    return Ud, Uo


def gp_prior_scale_tril(Ud, Uo):
    """Dense per-dof scale_tril T = U_dof^-T (2H x 2H, index 2t + {0: pos, 1: vel}) from the bidiagonal factor of
    gp_prior_factor -- what MultivariateNormal(precision_matrix=K^-1) derives (multivariate_normal.py:80-86),
    restricted to one degree of freedom.  fp64 on the host; O(H^2) numbers."""
    H = Ud.shape[0]
    U = np.zeros((2 * H, 2 * H))
    for t in range(H):
        U[2 * t, 2 * t], U[2 * t, 2 * t + 1], U[2 * t + 1, 2 * t + 1] = Ud[t]
        if t < H - 1:
            U[2 * t:2 * t + 2, 2 * t + 2:2 * t + 4] = Uo[t].reshape(2, 2)
    from scipy.linalg import solve_triangular
    return np.ascontiguousarray(solve_triangular(U, np.eye(2 * H), lower=False).T)


def const_vel_mean(start_pos, goal_pos, H, dt):
    """Straight line with constant velocity, zero velocity at both ends (mp_priors_multi.py:130-151)."""
    D = start_pos.shape[-1]
    n = H - 1
    traj = torch.zeros(H, 2 * D, dtype=torch.float64)
    i = torch.arange(H, dtype=torch.float64).reshape(H, 1)
    traj[:, :D] = start_pos.double().cpu() * (n - i) * 1. / n + goal_pos.double().cpu() * i * 1. / n
    traj[1:-1, D:] = (goal_pos.double().cpu() - start_pos.double().cpu()) / (n * dt)
    return traj


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

    def _full_state(self, state):
        """A start / goal state for reset(): the reference stores what it is given (gpmp2.py:178-182,
        stoch_gpmp.py:99-103), i.e. it expects the 2*n_dof-wide state the constructor built; a position-only
        state (n_dof wide, what the constructor itself takes) gets its zero velocities appended here."""
        if state.shape[-1] == self.n_dof:
            return torch.cat([state, torch.zeros_like(state)], -1)
        assert state.shape[-1] == 2 * self.n_dof, 'state must be n_dof or 2*n_dof wide'
        return state.detach().clone()

    def get_random_trajs(self, noise='torch_cpu', seed=0):
        """Initial particles from the constant-velocity GP prior (base.py:155-202): num_particles_per_goal
        samples per goal around the straight line, returned as (num_goals * ppg, H, 2D) fp32 -- 2D wide
        whatever ``pos_only`` is, like the reference.  The (2x2) (x) I_D structure of the prior precision is
        exploited (csrc/mpb_prior.hip); eps is drawn like MultivariateNormal.sample((ppg,)) does (fp64, CPU
        generator) unless noise='philox'."""
        D, H = self.n_dof, self.n_support_points
        start = self.start_state[..., :D]
        goal_directed = self.multi_goal_states is not None
        goals = self.multi_goal_states[..., :D] if goal_directed else start.reshape(1, D)
        G = goals.shape[0]
        if goal_directed:
            means = torch.stack([const_vel_mean(start, goals[i], H, self.dt) for i in range(G)])
        else:   # no goal: the prior mean is the start state repeated (mp_priors_multi.py:176)
            s = torch.cat([start.double().cpu(), torch.zeros(D, dtype=torch.float64)])
            means = s.repeat(1, H, 1)
        Ud, Uo = gp_prior_factor(H, self.dt, self.sigma_start_init, self.sigma_gp_init,
                                 self.sigma_goal_init if goal_directed else None)
        n = self.num_particles_per_goal
        eps = None
        if noise != 'philox':
            eps = torch.empty(n, G, H * 2 * D, dtype=torch.float64).normal_().to(self.device)
        f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).to(self.device).contiguous()
        tril = f64(gp_prior_scale_tril(Ud, Uo)) if H <= 128 else None      # dense GEMM on the matrix cores
        return ops.gp_prior_sample(f64(means), eps, f64(Ud), f64(Uo), n, D, seed=seed, scale_tril=tril)

    def get_GP_prior(self, start_K, gp_K, goal_K, state_init, particle_means=None, goal_states=None, tensor_args=None):
        """base.py:115-139: the GP trajectory prior over the planner's horizon as a MultiMPPrior (GPU sampling)."""
        from .costs.factors.mp_priors_multi import MultiMPPrior
        return MultiMPPrior(self.n_support_points - 1, self.dt, self.dim, self.n_dof, start_K, gp_K, state_init,
                            K_g_inv=goal_K, means=particle_means, goal_states=goal_states,
                            tensor_args=self.tensor_args if tensor_args is None else tensor_args)

    def const_vel_trajectories(self, start_state, multi_goal_states):
        """base.py:141-153, incl. its quirk: the velocity channel is (goal - start) / (H * dt), not / ((H-1) * dt).
        One (H, 2D) straight line per goal; host-side set-up arithmetic (a few KB), placed on the planner's device."""
        H, D = self.n_support_points, self.n_dof
        start_state = torch.as_tensor(start_state, dtype=torch.float32).reshape(-1, start_state.shape[-1]).cpu()
        goals = torch.as_tensor(multi_goal_states, dtype=torch.float32).cpu()
        traj = torch.zeros(goals.shape[0], H, 2 * D)
        mean_vel = (goals[:, :D] - start_state[:, :D]) / (H * self.dt)
        for i in range(H):
            traj[:, i, :D] = start_state[:, :D] * (H - i - 1) / (H - 1) + goals[:, :D] * i / (H - 1)
        traj[:, :, D:] = mean_vel.unsqueeze(1).repeat(1, H, 1)
        return traj.to(self.device)

    def _get_traj(self):
        trajs = self._particle_means.clone()
        if self.pos_only:
            # central-difference velocities by mpb_traj_finite_difference ([pos, vel] channels; the vel half is taken)
            vels = ops.traj_finite_difference(trajs.contiguous(), self.dt)[..., self.n_dof:]
            trajs = torch.cat((trajs, vels), dim=1)   # quirk Q12: concatenated along the horizon axis
        return trajs

    def get_traj(self):
        return self._get_traj()

    def _get_costs(self, state_trajectories, **observation):
        if self.cost is None:
            return torch.zeros(self.num_particles, )     # quirk Q11 (base.py:219-220): CPU, default dtype
        return self.cost(state_trajectories, **observation)

    def render(self, ax, **kwargs):
        raise NotImplementedError
