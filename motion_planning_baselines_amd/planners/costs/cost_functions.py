"""Cost-function objects with the reference's API surface (mp_baselines/planners/costs/cost_functions.py)
whose evaluation runs in the HIP library.

``Cost.__call__(trajs) -> (B,)`` / ``eval`` / ``get_linear_system`` keep the reference's meaning
(cost_functions.py:18-53).  The planners recognise these objects and fuse their arithmetic into the
planner kernels (no per-term launches); called on their own, ``eval`` launches the stand-alone HIP
collision-cost kernel (mpb_cost_collision_eval).  There is no CPU path: trajectories must be CUDA
tensors.
"""
from abc import ABC, abstractmethod

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

    def get_linear_system(self, trajs, **observation):
        """Dense (A, b, K) rows of the collision factor (cost_functions.py:191-231), assembled from the
        HIP per-waypoint cost and Jacobian.  Debug / inspection aid: GPMP2 itself never materialises it."""
        trajs = self._as_3d(trajs)
        B, H, d = trajs.shape
        geom = self.device_geometry(trajs.device)
        _, pw = ops.cost_collision_eval(trajs, geom, 1.0, per_waypoint=True)
        _, grad = ops.cost_collision_grad(trajs, geom, 1.0)
        N = self.dim * H
        A = torch.zeros(B, H - 1, N, device=trajs.device, dtype=trajs.dtype)
        for i in range(H - 1):
            A[:, i, (i + 1) * self.dim:(i + 1) * self.dim + self.n_dof] = -grad[:, i + 1, :self.n_dof]
        b = pw[:, 1:].unsqueeze(-1)
        K = self.k_sigma * torch.eye(H - 1, device=trajs.device, dtype=trajs.dtype).repeat(B, 1, 1)
        return A, b, K


class CostComposite(Cost):
    """Weighted sum of member costs (cost_functions.py:56-144)."""

    def __init__(self, robot, n_support_points, cost_list, weights_cost_l=None, **kwargs):
        super().__init__(robot, n_support_points, **kwargs)
        self.cost_l = cost_list
        self.weight_cost_l = weights_cost_l if weights_cost_l is not None else [1.0] * len(cost_list)

    def eval(self, trajs, trajs_interpolated=None, return_invidual_costs_and_weights=False, **kwargs):
        trajs = self._as_3d(trajs)
        if return_invidual_costs_and_weights:
            return [c(trajs, **kwargs) for c in self.cost_l], self.weight_cost_l
        total = 0
        for cost, w in zip(self.cost_l, self.weight_cost_l):
            if isinstance(cost, CostCollision) and cost.field is not None:
                total = total + ops.cost_collision_eval(trajs, cost.device_geometry(trajs.device), cost.k_sigma, weight=w)
            else:
                total = total + w * cost(trajs, **kwargs)
        return total

    def get_linear_system(self, trajs, **kwargs):
        As, bs, Ks = [], [], []
        for cost in self.cost_l:
            r = cost.get_linear_system(trajs, **kwargs)
            if r is None or any(v is None for v in r):
                continue
            As.append(r[0]); bs.append(r[1]); Ks.append(r[2])
        A, b = torch.cat(As, 1), torch.cat(bs, 1)
        K = torch.zeros(A.shape[0], A.shape[1], A.shape[1], device=A.device, dtype=A.dtype)
        o = 0
        for Kp in Ks:
            m = Kp.shape[1]
            K[:, o:o + m, o:o + m] = Kp
            o += m
        return A, b, K

    def single_collision_term(self):
        """(CostCollision, weight) if this composite is exactly one collision field -- the case the
        planner kernels fuse; None otherwise."""
        if len(self.cost_l) == 1 and isinstance(self.cost_l[0], CostCollision) and self.cost_l[0].field is not None:
            return self.cost_l[0], float(self.weight_cost_l[0])
        return None


def fusable_collision(cost):
    """Return (CostCollision, weight) when `cost` is a collision cost the kernels can fuse, else None."""
    if isinstance(cost, CostCollision) and cost.field is not None:
        return cost, 1.0
    if isinstance(cost, CostComposite):
        return cost.single_collision_term()
    return None
