"""Synthetic inputs of the BASELINE.md configs (C1..C5): geometry, start/goal problems, initial means.

Used by bench.py, __graft_entry__.smoke() and the tests so that they all see the same workload.
Everything is built from seeded numpy generators on the host; collision checks of candidate
configurations run through the HIP collision kernel.
"""
import numpy as np
import torch

from . import geometry as G
from . import ops


def straight_line_means(starts, goals, H, dt, pos_only, device):
    """(P,D) starts/goals -> (P,H,d) straight lines; constant velocity channel when not pos_only."""
    starts = torch.as_tensor(starts, dtype=torch.float32)
    goals = torch.as_tensor(goals, dtype=torch.float32)
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    pos = starts[:, None, :] * (1 - a) + goals[:, None, :] * a
    if not pos_only:
        vel = ((goals - starts) / ((H - 1) * dt))[:, None, :].expand(-1, H, -1)
        pos = torch.cat([pos, vel], -1)
    return pos.contiguous().to(device)


def collision_free_configs(robot, field, n, seed, device, lo=None, hi=None):
    """n configurations uniformly within [lo,hi] (default: joint limits) whose collision cost is 0."""
    rng = np.random.RandomState(seed)
    lo = robot.q_min_np if lo is None else np.asarray(lo, np.float32)
    hi = robot.q_max_np if hi is None else np.asarray(hi, np.float32)
    geom = ops.DeviceGeometry(robot, field, device)
    out = []
    while sum(len(o) for o in out) < n:
        q = (lo + (hi - lo) * rng.rand(4 * n + 64, robot.q_dim)).astype(np.float32)
        c = ops.cost_collision_eval(torch.from_numpy(q).to(device).reshape(-1, 1, robot.q_dim).contiguous(), geom,
                                    1.0, h_begin=0)
        out.append(q[(c == 0).cpu().numpy()])
    return np.concatenate(out)[:n]


def panda_spheres_stomp(P, device, H=64, S=32, pos_only=False, seed=0, first_particle=0):
    """C3 / C5: Panda + 16 obstacle spheres, P independent start/goal problems, straight-line means.
    Particle i of the global problem set is the same whatever shard it lands on (first_particle)."""
    robot = G.RobotPanda()
    field = G.env_spheres_3d(seed=0)
    dt = 5.0 / H
    n_total = first_particle + P
    q = collision_free_configs(robot, field, 2 * n_total, seed + 17, device)
    starts, goals = q[:n_total][first_particle:], q[n_total:][first_particle:]
    means0 = straight_line_means(starts, goals, H, dt, pos_only, device)
    params = dict(n_dof=7, n_support_points=H, num_particles_per_goal=P, num_samples=S, dt=dt,
                  temperature=1.0, step_size=0.1, sigma_spectral=0.1, pos_only=pos_only)
    return dict(robot=robot, field=field, starts=starts, goals=goals, means0=means0, params=params,
                sigma_coll=1e-3)


def panda_crowded_stomp(P, device, H=64, S=32, pos_only=False, seed=0):
    """C3's shape in a crowded scene (200 spheres + 32 boxes, geometry.env_spheres_boxes_3d: the list grid of geometry version 7)."""
    robot = G.RobotPanda()
    field = G.env_spheres_boxes_3d(seed=seed)
    dt = 5.0 / H
    q = collision_free_configs(robot, field, 2 * P, seed + 31, device)
    means0 = straight_line_means(q[:P], q[P:], H, dt, pos_only, device)
    params = dict(n_dof=7, n_support_points=H, num_particles_per_goal=P, num_samples=S, dt=dt,
                  temperature=1.0, step_size=0.1, sigma_spectral=0.1, pos_only=pos_only)
    return dict(robot=robot, field=field, starts=q[:P], goals=q[P:], means0=means0, params=params, sigma_coll=1e-3)


def pointmass_grid_circles_stomp(device, P=4, S=4, H=64):
    """C1 (examples/pointmass_grid_circles_2d_STOMP.py:53-96 parameters)."""
    robot = G.RobotPointMass(2, radius=0.01)
    field = G.env_grid_circles_2d()
    dt = 0.04
    starts = np.tile(np.array([[-0.8, -0.8]], np.float32), (P, 1))
    goals = np.tile(np.array([[0.8, 0.8]], np.float32), (P, 1))
    means0 = straight_line_means(starts, goals, H, dt, False, device)
    params = dict(n_dof=2, n_support_points=H, num_particles_per_goal=P, num_samples=S, dt=dt,
                  temperature=1.0, step_size=0.1, sigma_spectral=0.1, pos_only=False)
    return dict(robot=robot, field=field, starts=starts, goals=goals, means0=means0, params=params,
                sigma_coll=1e-3)


def pointmass_dense_chomp(B, device, H=64, seed=3):
    """C2: dense 2-D circles + boxes, B random free start/goal problems, straight line + N(0,0.01^2)."""
    robot = G.RobotPointMass(2, radius=0.01)
    field = G.env_dense_2d(seed=seed)
    dt = 0.04
    q = collision_free_configs(robot, field, 2 * B, seed + 5, device, lo=[-0.95, -0.95], hi=[0.95, 0.95])
    means0 = straight_line_means(q[:B], q[B:], H, dt, False, device)
    gen = torch.Generator().manual_seed(seed)
    means0[:, 1:-1, :2] += (0.01 * torch.randn(B, H - 2, 2, generator=gen)).to(device)
    params = dict(n_dof=2, n_support_points=H, num_particles_per_goal=B, dt=dt, weight_prior_cost=1e-4,
                  step_size=0.05, grad_clip=0.05, pos_only=False)
    return dict(robot=robot, field=field, starts=q[:B], goals=q[B:], means0=means0, params=params,
                sigma_coll=1.0, weight=10.0)
