"""ORACLE (test infrastructure). Stand-ins for the 7 ``torch_robotics`` symbols the reference imports.

The reference (mp_baselines/planners/base.py:9, chomp.py:4-5, costs/cost_functions.py:12-15,
gpmp2.py:20, hybrid_planner.py:5-7) imports helper functions from the absent, un-pinned third-party
package ``torch_robotics``.  ``install()`` registers minimal stand-in modules in ``sys.modules`` so the
UNMODIFIED reference planner classes can be imported from /root/reference IN THIS CONTAINER ONLY, for
generating golden vectors (tests/golden/make_goldens.py).  Nothing here is copied from torch_robotics
(it is not on disk); the functions are build-defined and documented in DESIGN.md:

  batched_weighted_dot_prod(x, M, y) : per (batch, channel) x_c^T M y_c   -> (..., d)
  finite_difference_vector(x, dt)    : central difference along the horizon axis, zero at both ends
  TimerCUDA                          : wall-clock context manager
  interpolate_points_v1(x, n)        : n evenly spaced points inserted between every pair of consecutive
                                       waypoints (linear, in joint space): (B,H,d) -> (B,(H-1)(n+1)+1,d)
  link_pos_from_link_tensor, tensor_linspace_v1, smoothen_trajectory :
      imported by the reference but never executed on the paths in scope; they raise if called.
"""
import sys
import time
import types

import torch


def batched_weighted_dot_prod(x, M, y, with_einsum=False):
    r = x.transpose(-2, -1) @ M.unsqueeze(0) @ y
    return r.diagonal(dim1=-2, dim2=-1)


def interpolate_points_v1(points, num_interpolated_points):
    x0, x1 = points[..., :-1, None, :], points[..., 1:, None, :]
    n = num_interpolated_points
    alpha = (torch.arange(n + 1, dtype=points.dtype, device=points.device) / (n + 1)).reshape(n + 1, 1)
    seg = x0 + alpha * (x1 - x0)                                   # (..., H-1, n+1, d): start point + n interior
    return torch.cat((seg.flatten(-3, -2), points[..., -1:, :]), dim=-2)


def finite_difference_vector(x, dt=1.0, method='central'):
    out = torch.zeros_like(x)
    if method == 'central':
        out[..., 1:-1, :] = (x[..., 2:, :] - x[..., :-2, :]) / (2 * dt)
    elif method == 'forward':
        out[..., :-1, :] = (x[..., 1:, :] - x[..., :-1, :]) / dt
    elif method == 'backward':
        out[..., 1:, :] = (x[..., 1:, :] - x[..., :-1, :]) / dt
    else:
        raise NotImplementedError(method)
    return out


class TimerCUDA:
    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *a):
        self.elapsed = time.perf_counter() - self._t0

    def __str__(self):
        return f'{getattr(self, "elapsed", float("nan")):.6f}'


def _not_in_scope(name):
    def f(*a, **k):
        raise NotImplementedError(f'torch_robotics.{name} is outside the hot path in scope')
    return f


def install():
    """Register the stand-in ``torch_robotics`` module tree (idempotent)."""
    if 'torch_robotics' in sys.modules and getattr(sys.modules['torch_robotics'], '_mpb_stub', False):
        return
    tree = {
        'torch_robotics': {},
        'torch_robotics.trajectory': {},
        'torch_robotics.trajectory.utils': dict(
            finite_difference_vector=finite_difference_vector,
            smoothen_trajectory=_not_in_scope('smoothen_trajectory')),
        'torch_robotics.torch_utils': {},
        'torch_robotics.torch_utils.torch_utils': dict(
            batched_weighted_dot_prod=batched_weighted_dot_prod,
            tensor_linspace_v1=_not_in_scope('tensor_linspace_v1')),
        'torch_robotics.torch_utils.torch_timer': dict(TimerCUDA=TimerCUDA),
        'torch_robotics.torch_kinematics_tree': {},
        'torch_robotics.torch_kinematics_tree.geometrics': {},
        'torch_robotics.torch_kinematics_tree.geometrics.utils': dict(
            link_pos_from_link_tensor=_not_in_scope('link_pos_from_link_tensor')),
        'torch_robotics.torch_planning_objectives': {},
        'torch_robotics.torch_planning_objectives.fields': {},
        'torch_robotics.torch_planning_objectives.fields.distance_fields': dict(
            interpolate_points_v1=interpolate_points_v1),
    }
    for name, attrs in tree.items():
        m = types.ModuleType(name)
        m.__path__ = []
        m._mpb_stub = True
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    for name in tree:
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(sys.modules[parent], child, sys.modules[name])


def import_reference(ref_root='/root/reference'):
    """Import the reference package (this container only). Returns the ``mp_baselines`` module."""
    install()
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    import mp_baselines  # noqa: F401
    return mp_baselines
