import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def ref_geometry_from_golden(g, dtype=torch.float32):
    """(RefRobot, RefCollisionField) of the oracle from the arrays stored in a golden file."""
    from oracle.geometry_ref import RefRobot, RefCollisionField
    ta = dict(device='cpu', dtype=dtype)
    spec = dict(kind=int(g['robot_kind']), n_dof=int(g['n_dof']), joint_tf=g['joint_tf'],
                link_frame=g['link_frame'], link_offset=g['link_offset'], link_radius=g['link_radius'])
    robot = RefRobot(spec, tensor_args=ta)
    field = RefCollisionField(dict(spheres=g['spheres'], boxes=g['boxes'], margin=g['margin']),
                              g['link_radius'], tensor_args=ta)
    n_extra = int(g['n_extra_fields']) if 'n_extra_fields' in g else 0
    if n_extra:       # goldens with several collision fields: a list, first field first
        field = [field] + [RefCollisionField(dict(spheres=g[f'extra{i}_spheres'], boxes=g[f'extra{i}_boxes'],
                                                  margin=g[f'extra{i}_margin']), g['link_radius'], tensor_args=ta)
                           for i in range(n_extra)]
    return robot, field


def product_geometry_from_golden(g):
    """(robot, field) product spec objects rebuilt from the arrays stored in a golden file."""
    from motion_planning_baselines_amd import geometry as G
    if int(g['robot_kind']) == 0:
        robot = G.RobotPointMass(int(g['n_dof']), radius=float(g['link_radius'][0]))
    else:
        D = int(g['n_dof'])
        robot = G.RobotSerialChain(g['joint_tf'], g['link_frame'], g['link_offset'], g['link_radius'],
                                   q_min=[-3.2] * D, q_max=[3.2] * D)
    field = G.CollisionField(spheres=g['spheres'] if len(g['spheres']) else None,
                             boxes=g['boxes'] if len(g['boxes']) else None, margin=float(g['margin']))
    n_extra = int(g['n_extra_fields']) if 'n_extra_fields' in g else 0
    if n_extra:
        field = [field] + [G.CollisionField(spheres=g[f'extra{i}_spheres'] if len(g[f'extra{i}_spheres']) else None,
                                            boxes=g[f'extra{i}_boxes'] if len(g[f'extra{i}_boxes']) else None,
                                            margin=float(g[f'extra{i}_margin'])) for i in range(n_extra)]
    return robot, field


@pytest.fixture(scope='session')
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def rel_err_waypoint(a, b, n_pos=None, floor=1e-2):
    """Per-waypoint L2-relative error (north_star: "1e-4 relative on final trajectory waypoints"), the stricter companion
    of the global-max norm `max|a-b| / max|b|`: for every waypoint (p, h) and, separately, for its position channels
    [:n_pos] and its velocity channels [n_pos:] (different units),  ||a_ph - b_ph||_2 / max(||b_ph||_2, floor * max_ph ||b_ph||_2);
    the floor (1 % of the largest waypoint norm of that channel group) only keeps waypoints at the origin from dividing by
    zero.  Returns the maximum over waypoints and channel groups."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    n_pos = a.shape[-1] if n_pos is None else n_pos
    worst = 0.0
    for sl in (slice(0, n_pos), slice(n_pos, a.shape[-1])):
        if sl.start >= a.shape[-1]:
            continue
        nb = b[..., sl].norm(dim=-1)
        den = nb.clamp_min(floor * float(nb.max().clamp_min(1e-12)))
        worst = max(worst, float(((a[..., sl] - b[..., sl]).norm(dim=-1) / den).max()))
    return worst
