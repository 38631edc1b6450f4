"""ORACLE (test infrastructure, never shipped, never imported by the product path).

CPU / PyTorch evaluation of the BUILD-DEFINED geometry back-end (robot forward kinematics of the
collision spheres, sphere / box signed distance, hinge collision cost) from the same arrays the
HIP kernels read (motion_planning_baselines_amd/geometry.py ``spec()``).

PARITY UNPINNED: in the reference these functions live in the absent third-party package
``torch_robotics`` (no version pinned; call sites mp_baselines/planners/costs/cost_functions.py:50-52
``robot.get_position / get_velocity / fk_map_collision`` and
mp_baselines/planners/costs/factors/field_factor.py:39,52,56 ``field.compute_cost`` / ``zero_grad``).
The classes below are duck-typed to exactly those call sites so that the UNMODIFIED reference planner
loop can run on them (tests/golden/make_goldens.py) and so that the oracle restatement
(oracle/planners_ref.py) and the reference see identical geometry.

Everything is differentiable torch code: CHOMP / GPMP2 in the reference obtain collision gradients
by autograd through these calls (chomp.py:139, field_factor.py:54).
"""
import torch


def _safe_norm(v):
    """sqrt(x^2 + y^2 + z^2) with a zero (not NaN) sub-gradient at the origin."""
    return torch.sqrt((v * v).sum(-1).clamp_min(1e-30))


class RefRobot:
    """Duck-typed ``robot`` (q_dim, get_position, get_velocity, fk_map_collision, q_min, q_max, dt)."""

    def __init__(self, robot_spec, q_min=None, q_max=None, dt=None, tensor_args=None):
        ta = tensor_args or dict(device='cpu', dtype=torch.float32)
        self.tensor_args = ta
        self.kind = int(robot_spec['kind'])
        self.q_dim = int(robot_spec['n_dof'])
        self.joint_tf = torch.as_tensor(robot_spec['joint_tf']).to(**ta)          # (n_tf,3,4)
        self.link_frame = torch.as_tensor(robot_spec['link_frame']).long()         # (L,)
        self.link_offset = torch.as_tensor(robot_spec['link_offset']).to(**ta)     # (L,3)
        self.link_radius = torch.as_tensor(robot_spec['link_radius']).to(**ta)     # (L,)
        self.q_min = None if q_min is None else torch.as_tensor(q_min).to(**ta)
        self.q_max = None if q_max is None else torch.as_tensor(q_max).to(**ta)
        self.dt = dt

    def get_position(self, x):
        return x[..., :self.q_dim]

    def get_velocity(self, x):
        return x[..., self.q_dim:2 * self.q_dim]

    def fk_map_collision(self, q):
        """q (..., D) -> collision-sphere centres (..., L, 3)."""
        if self.kind == 0:
            if self.q_dim == 2:
                z = torch.zeros_like(q[..., :1])
                p = torch.cat([q[..., :2], z], dim=-1)
            else:
                p = q[..., :3]
            return p.unsqueeze(-2)
        D = self.q_dim
        lead = q.shape[:-1]
        qf = q.reshape(-1, D)
        n = qf.shape[0]
        R = torch.eye(3, **self.tensor_args).expand(n, 3, 3)
        t = torch.zeros(n, 3, **self.tensor_args)
        centres = [None] * len(self.link_frame)
        for j in range(self.joint_tf.shape[0]):
            P = self.joint_tf[j]
            # frame * P_j
            t = t + (R @ P[:, 3])
            R = R @ P[:, :3]
            if j < D:
                c, s = torch.cos(qf[:, j]), torch.sin(qf[:, j])
                c0, c1 = R[:, :, 0], R[:, :, 1]
                n0 = c0 * c[:, None] + c1 * s[:, None]
                n1 = c1 * c[:, None] - c0 * s[:, None]
                R = torch.stack([n0, n1, R[:, :, 2]], dim=-1)
            frame = j + 1
            for l in (self.link_frame == frame).nonzero().flatten().tolist():
                centres[l] = t + (R @ self.link_offset[l])
        out = torch.stack(centres, dim=-2)
        return out.reshape(*lead, len(self.link_frame), 3)


class RefCollisionField:
    """Duck-typed ``field`` (compute_cost(q_pos, link_pos, **kw), zero_grad())."""

    def __init__(self, field_spec, link_radius, tensor_args=None):
        ta = tensor_args or dict(device='cpu', dtype=torch.float32)
        self.tensor_args = ta
        self.spheres = torch.as_tensor(field_spec['spheres']).to(**ta).reshape(-1, 4)
        self.boxes = torch.as_tensor(field_spec['boxes']).to(**ta).reshape(-1, 6)
        self.margin = float(field_spec['margin'])
        self.link_radius = torch.as_tensor(link_radius).to(**ta)

    def signed_distance(self, x):
        """x (..., 3) -> min over all obstacles of the signed distance, shape (...)."""
        sds = []
        if len(self.spheres):
            d = x.unsqueeze(-2) - self.spheres[:, :3]
            dist = _safe_norm(d)
            sds.append(dist - self.spheres[:, 3])
        if len(self.boxes):
            qv = (x.unsqueeze(-2) - self.boxes[:, :3]).abs() - self.boxes[:, 3:6]
            qo = torch.clamp(qv, min=0.0)
            outside = _safe_norm(qo)
            inside = torch.clamp(qv.max(dim=-1)[0], max=0.0)
            sds.append(outside + inside)
        sd = torch.cat(sds, dim=-1)
        return sd.min(dim=-1)[0]

    def compute_cost(self, q_pos, link_pos, **kwargs):
        """link_pos (B, H, L, 3) -> (B, H) hinge cost summed over collision spheres."""
        sd = self.signed_distance(link_pos)                      # (B,H,L)
        hinge = torch.relu(self.margin + self.link_radius - sd)  # (B,H,L)
        return hinge.sum(-1)

    def zero_grad(self):
        pass


def make_ref_geometry(robot, field, tensor_args=None):
    """Build (RefRobot, RefCollisionField) from the product's spec-holding objects."""
    rs = robot.spec()
    ref_robot = RefRobot(rs, q_min=robot.q_min_np, q_max=robot.q_max_np, dt=robot.dt, tensor_args=tensor_args)
    ref_field = RefCollisionField(field.spec(), rs['link_radius'], tensor_args=tensor_args)
    return ref_robot, ref_field
