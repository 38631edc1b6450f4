"""CHOMP with the reference's class surface (mp_baselines/planners/chomp.py), iterations on the GPU.

The reference obtains the gradient by autograd through FK + SDF and a dense x^T R x (chomp.py:135-169);
mpb_chomp_step evaluates the same gradient analytically (J^T grad sdf, tridiagonal R x) and runs the
whole ``opt_iters`` loop -- clamp, endpoint mask, step -- inside one launch.
"""
import torch

from .. import ops
from .base import OptimizationPlanner
from .costs.cost_functions import device_plan, fusable_collision


def chomp_precision_matrix(dt=0.01, n_support_points=64, tensor_args=None):
    """R = K^T K with K the (H+1) x H backward-difference operator (last row -1) scaled 1/dt^2 --
    CHOMP._get_R_mat (chomp.py:81-101)."""
    H = n_support_points
    K = torch.eye(H) - torch.diag(torch.ones(H - 1), -1)
    K = torch.cat((K, torch.zeros(1, H)), dim=0)
    K[-1, -1] = -1.
    K = K * 1. / dt ** 2
    return (K.t() @ K).to(**tensor_args)


class CHOMP(OptimizationPlanner):
    """Drop-in for mp_baselines.planners.chomp.CHOMP (ctor kwargs chomp.py:11-30).

    Extra keyword argument: ``global_batch`` -- total number of particles across all shards (quirk Q3:
    the reference's smoothness gradient is scaled by the batch size; a shard must use the global one).
    """

    def __init__(self, n_dof, n_support_points, num_particles_per_goal, opt_iters, dt, start_state, cost=None,
                 weight_prior_cost=0.1, initial_particle_means=None, step_size=1., grad_clip=.01,
                 multi_goal_states=None, sigma_start_init=0.001, sigma_goal_init=0.001, sigma_gp_init=10.,
                 pos_only=True, global_batch=None, **kwargs):
        super().__init__(name='CHOMP', n_dof=n_dof, n_support_points=n_support_points,
                         num_particles_per_goal=num_particles_per_goal, opt_iters=opt_iters, dt=dt,
                         start_state=start_state, cost=cost, initial_particle_means=initial_particle_means,
                         multi_goal_states=multi_goal_states, sigma_start_init=sigma_start_init,
                         sigma_goal_init=sigma_goal_init, sigma_gp_init=sigma_gp_init, pos_only=pos_only, **kwargs)
        self.lr = step_size
        self.grad_clip = grad_clip
        self.global_batch = global_batch
        cpu = dict(device='cpu', dtype=torch.float32)
        R = self._get_R_mat(dt=dt, n_support_points=n_support_points, tensor_args=cpu)
        self.Sigma_inv = R.to(self.device).contiguous()
        self.Sigma = torch.inverse(R).to(self.device)
        self.reset(initial_particle_means=initial_particle_means)
        self.weight_prior_cost = weight_prior_cost
        self.costs = torch.zeros(self.num_particles, device=self.device, dtype=torch.float32)

    @classmethod
    def _get_R_mat(cls, dt=0.01, n_support_points=64, tensor_args=None, **kwargs):
        return chomp_precision_matrix(dt=dt, n_support_points=n_support_points, tensor_args=tensor_args)

    def _get_R_mat2(self):
        """chomp.py:60-79: the STOMP-style second-difference precision (unused by optimize, kept for callers)."""
        from .stomp import stomp_precision_matrix
        return stomp_precision_matrix(self.n_support_points, self.dt, 1.0,
                                      dict(device='cpu', dtype=torch.float32)).to(self.device)

    def _eval(self, x, **observation):
        """chomp.py:153-169: collision costs (B,) plus the smoothness prior -- the batch-wide scalar
        weight_prior_cost * sum_{b,channel} x^T R x (quirk Q3) -- added to every entry."""
        from .costs.cost_functions import CostSmoothnessCHOMP  # noqa: F401  (same kernel term)
        fused = fusable_collision(self.cost)
        if x.ndim == 2:
            x = x.unsqueeze(0)
        x = x.contiguous()
        costs = torch.zeros(x.shape[0], device=x.device, dtype=torch.float32)
        if fused is not None:
            cc, weight = fused
            costs = ops.cost_collision_eval(x, cc.device_geometry(x.device), cc.k_sigma, weight=weight)
        smooth, _ = ops.cost_terms_eval(x, self.n_dof, dt=self.dt, k_smooth=1.0, terms={'smooth'})
        return costs + self.weight_prior_cost * smooth.sum()

    def reset(self, initial_particle_means=None):
        """chomp.py:103-111."""
        if initial_particle_means is not None:
            m = initial_particle_means.clone()
        else:
            m = self.get_random_trajs()
        self._particle_means = m.to(device=self.device, dtype=torch.float32).contiguous()

    def optimize(self, opt_iters=None, **observation):
        """chomp.py:113-125."""
        self._run_optimization(opt_iters, **observation)
        return self._get_traj()

    def _run_optimization(self, opt_iters, **observation):
        if opt_iters is None:
            opt_iters = self.opt_iters
        fused = fusable_collision(self.cost)
        B_global = self.global_batch or self.num_particles
        if fused is not None:
            cc, weight = fused
            ops.chomp_step(self._particle_means, self.Sigma_inv, cc.device_geometry(self.device), self.n_dof,
                           cc.k_sigma, weight, self.weight_prior_cost, self.lr, self.grad_clip, n_iters=opt_iters,
                           B_global=B_global, costs_out=self.costs)
            return
        # any composite of HIP-served members (collision fields + GP / smoothness / joint-limit / start / goal terms): the
        # reference differentiates it by autograd (chomp.py:135-139); here per iteration the collision gradient kernel
        # (J^T grad sdf), then ONE pass that adds the closed-form gradients of the trajectory terms and of the
        # smoothness prior, clamps, masks the end rows and steps (chomp.py:141-147)
        plan = device_plan(self.cost, self.device) if self.cost is not None else None
        if plan is None:
            return self._run_optimization_autograd(opt_iters, B_global, **observation)
        cc, weight, groups = plan
        x = self._particle_means
        grad = torch.empty_like(x) if (cc is not None or len(groups) > 1) else None
        prior_bw = float(B_global) * float(self.weight_prior_cost)
        for _ in range(opt_iters):
            have = False
            if cc is not None:
                self.costs, _ = ops.cost_collision_grad(x, cc.device_geometry(self.device), cc.k_sigma, weight=weight, grad=grad)
                have = True
            for spec in groups[:-1]:     # term groups that could not be merged into one launch: accumulate
                ops.cost_terms_grad(x, self.n_dof, grad_in=grad if have else None, grad_out=grad, jl_scale=float(B_global), **spec)
                have = True
            last = groups[-1] if groups else dict(terms=())
            ops.cost_terms_grad(x, self.n_dof, grad_in=grad if have else None, apply=True, R=self.Sigma_inv, prior_bw=prior_bw,
                                lr=self.lr, grad_clip=self.grad_clip, jl_scale=float(B_global), **last)

    def _run_optimization_autograd(self, opt_iters, B_global, **observation):
        """A caller-supplied cost (any callable on device tensors that torch can differentiate -- e.g. written against
        robot_field.DeviceRobot / DeviceField, whose kernels carry hand-written vector-Jacobian products): the reference
        differentiates whatever it is handed (chomp.py:135-139).  Per iteration the CALLER's cost is differentiated by
        torch.autograd.grad; the smoothness prior's gradient (B w (R + R^T) x, quirk Q3), the clamp, the end-row mask
        and the step (chomp.py:141-147) stay in the HIP kernel, which takes that gradient as `grad_in`."""
        prior_bw = float(B_global) * float(self.weight_prior_cost)
        x = self._particle_means
        for _ in range(opt_iters):
            if self.cost is None:
                g = None                          # quirk Q11: no cost -> only the prior acts
            else:
                xg = x.detach().requires_grad_(True)
                costs = self._get_costs(xg, **observation)
                if not (isinstance(costs, torch.Tensor) and costs.requires_grad):
                    raise NotImplementedError(
                        'CHOMP differentiates its cost: the cost returned a tensor torch.autograd cannot trace back to the '
                        'trajectories (build it from differentiable torch ops, robot_field.DeviceRobot / DeviceField, or '
                        'the cost classes of cost_functions.py)')
                (g,) = torch.autograd.grad(costs.sum(), xg)
                g = g.to(torch.float32).contiguous()
                self.costs = costs.detach()
            ops.cost_terms_grad(x, self.n_dof, grad_in=g, apply=True, R=self.Sigma_inv, prior_bw=prior_bw, lr=self.lr,
                                grad_clip=self.grad_clip, jl_scale=float(B_global), terms=())
