"""FieldFactor (mp_baselines/planners/costs/factors/field_factor.py:4-57): hinge cost of a collision field over a
range of waypoints, and its Jacobian w.r.t. the joint positions.  The reference calls `field.compute_cost` and
differentiates with autograd; here both come from the HIP collision kernels (analytic gradient)."""
from .... import ops


class FieldFactor:

    def __init__(self, n_dof, sigma, traj_range):
        self.sigma = sigma
        self.n_dof = n_dof
        self.traj_range = traj_range
        self.K = 1. / (sigma ** 2)

    def _range(self, H):
        lo, hi = self.traj_range
        lo = lo if lo >= 0 else H + lo
        hi = H if hi is None else (hi if hi >= 0 else H + hi)
        return lo, hi

    def get_error(self, q_trajs, field, q_pos=None, q_vel=None, H_pos=None, calc_jacobian=True, robot=None,
                  n_interpolated_points=None, **kwargs):
        """error (B, len) = field cost of the waypoints in traj_range; with calc_jacobian also
        H = -d(sum error)/d q over the same range, (B, len, n_dof).  `field` is a CollisionField (or a prepared
        ops.DeviceGeometry); `robot` is needed with a bare CollisionField.  n_interpolated_points: the Jacobian is
        that of the interpolated trajectory's error (field_factor.py:42-54), the error stays the support points'."""
        q_trajs = q_trajs.contiguous()
        B, H, d = q_trajs.shape
        geom = field if isinstance(field, ops.DeviceGeometry) else ops.DeviceGeometry(robot, field, q_trajs.device)
        lo, hi = self._range(H)
        _, pw = ops.cost_collision_eval(q_trajs, geom, 1.0, h_begin=lo, per_waypoint=True)
        error = pw[:, lo:hi]
        if not calc_jacobian:
            return error
        if n_interpolated_points:
            ws = ops.gpmp2_workspace(B, H, self.n_dof, q_trajs.device)
            x = q_trajs if d == 2 * self.n_dof else ops.traj_finite_difference(q_trajs[..., :self.n_dof].contiguous(), 1.0)
            ops.gpmp2_linearize(x, geom, ws, n_interp=n_interpolated_points)
            import torch
            jac = ws[:B * H * (self.n_dof + 1) * 4].view(torch.float32).reshape(B, H, self.n_dof + 1)
            return error, jac[:, lo:hi, :self.n_dof].clone()
        _, grad = ops.cost_collision_grad(q_trajs, geom, 1.0, h_begin=lo)
        return error, -grad[:, lo:hi, :self.n_dof]
