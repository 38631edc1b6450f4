"""Duck-typed `robot` / `field` objects on the GPU with the API the reference's cost layer consumes from torch_robotics
(SURVEY 8b): robot.q_dim / q_min / q_max / dt / get_position / get_velocity / fk_map_collision (cost_functions.py:21,
:50-52, :380, :412-420), field.compute_cost / zero_grad (field_factor.py:39, :52, :56).

With these the reference's own, unmodified CostCollision / FieldFactor / CostComposite run on GPU tensors against this
package's geometry; forward kinematics and the field cost are HIP kernels (csrc/mpb_points.hip) and torch.autograd
differentiates through them by their hand-written vector-Jacobian products.  The planners of this package do NOT go
through these objects: they use the fused evaluators.
"""
import torch

from . import ops


class _FKPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, geom):
        q3 = q.reshape(-1, q.shape[-2], q.shape[-1]).to(torch.float32).contiguous()
        ctx.geom, ctx.q_shape = geom, q.shape
        ctx.save_for_backward(q3)
        pts = ops.fk_collision_points(q3, geom)
        return pts.reshape(*q.shape[:-1], pts.shape[-2], 3)

    @staticmethod
    def backward(ctx, grad_pts):
        (q3,) = ctx.saved_tensors
        g = grad_pts.reshape(q3.shape[0], q3.shape[1], -1, 3).to(torch.float32).contiguous()
        gq = ops.fk_collision_points_vjp(q3, ctx.geom, g)                     # (B,H,n_dof)
        out = torch.zeros(q3.shape, device=q3.device, dtype=torch.float32)
        out[..., :gq.shape[-1]] = gq
        return out.reshape(ctx.q_shape), None


class _FieldCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, geom):
        p4 = pts.reshape(-1, pts.shape[-3], pts.shape[-2], 3).to(torch.float32).contiguous()
        ctx.geom, ctx.p_shape = geom, pts.shape
        ctx.save_for_backward(p4)
        return ops.field_cost_points(p4, geom).reshape(pts.shape[:-2])

    @staticmethod
    def backward(ctx, grad_cost):
        (p4,) = ctx.saved_tensors
        g = grad_cost.reshape(p4.shape[0], p4.shape[1]).to(torch.float32).contiguous()
        return ops.field_cost_points_vjp(p4, ctx.geom, g).reshape(ctx.p_shape), None


class DeviceRobot:
    """robot API of the reference's cost layer, evaluated on the GPU."""

    def __init__(self, robot, geom, device):
        self._robot, self._geom = robot, geom
        self.q_dim = robot.q_dim
        self.dt = robot.dt
        self.q_min = robot.q_min.to(device)
        self.q_max = robot.q_max.to(device)

    def get_position(self, x):
        return x[..., :self.q_dim]

    def get_velocity(self, x):
        return x[..., self.q_dim:2 * self.q_dim]

    def fk_map_collision(self, q_pos, **kwargs):
        """(..., H, >=q_dim) joint positions -> (..., H, L, 3) collision-sphere positions (differentiable)."""
        return _FKPoints.apply(q_pos, self._geom)


class DeviceField:
    """field API of the reference's cost layer: hinge cost of the robot's collision spheres, evaluated on the GPU."""

    def __init__(self, field, geom):
        self._field, self._geom = field, geom

    def compute_cost(self, q_pos, link_pos, **kwargs):
        """link_pos (..., H, L, 3) -> (..., H) cost per waypoint (differentiable w.r.t. link_pos); extra keyword arguments
        (`obstacle_spheres`, `trajs_interp`, ...) are accepted and ignored like the call sites require."""
        return _FieldCost.apply(link_pos, self._geom)

    def zero_grad(self):
        pass


def device_robot_field(robot, field, device):
    """(DeviceRobot, DeviceField) for one robot / collision field pair (geometry.Robot*, geometry.CollisionField)."""
    geom = ops.DeviceGeometry(robot, field, device, keep_all_links=True)
    return DeviceRobot(robot, geom, device), DeviceField(field, geom)
