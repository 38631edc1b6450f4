"""GPMP2 at the C4 shape for rocprofv3: bench.py's workload (collision-free start / goal configurations, straight-line means), a few
iterations, every one from the initial state (the low-rank form's time depends on the active collision rows).  MPB_GPMP2_FORM = block
selects the block elimination of rounds 1-5."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, workloads
dev = torch.device('cuda:0')
B, H, D = int(os.environ.get('GP_B', 2048)), 128, 7
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
dt = 5.0 / H
x0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
x0[:, 0, D:] = 0
x0[:, -1, D:] = 0
z = torch.zeros(B, D, device=dev)
start = torch.cat([torch.from_numpy(q[:B]).to(dev), z], -1).contiguous()
goal = torch.cat([torch.from_numpy(q[B:]).to(dev), z], -1).contiguous()
ws = ops.gpmp2_workspace(B, H, D, dev)
costs = torch.empty(B, device=dev)
x = x0.clone()
for _ in range(6):
    x.copy_(x0)
    ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), dt, 1e-2, True, 1.0, n_iters=1, costs_out=costs)
torch.cuda.synchronize()
print('ok', float(costs.mean()))
