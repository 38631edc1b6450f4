"""Tuning aid: the GPMP2 low-rank kernels on ONE particle of the C4 workload -- the one with the most active collision rows (or MPB_PICK=k: the
k-th largest) -- so that a kernel's duration under rocprofv3 is that particle's.    python scripts/ab_gpmp2_one.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, workloads
dev = torch.device('cuda:0')
B, H, D = 2048, 128, 7
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
dt = 5.0 / H
x0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
x0[:, 0, D:] = 0
x0[:, -1, D:] = 0
rows = ops.gpmp2_collision_rows(x0, geom)[0]
n_act = (rows[..., :D].abs().sum(-1) > 0).to(torch.float32).sum(1)
order = torch.argsort(n_act, descending=True)
pick = int(order[int(os.environ.get('MPB_PICK', 0))])
print('particle', pick, 'active rows', int(n_act[pick]))
x1 = x0[pick:pick + 1].contiguous()
z = torch.zeros(1, D, device=dev)
start = torch.cat([torch.from_numpy(q[pick:pick + 1]).to(dev), z], -1).contiguous()
goal = torch.cat([torch.from_numpy(q[B + pick:B + pick + 1]).to(dev), z], -1).contiguous()
ws = ops.gpmp2_workspace(1, H, D, dev)
x = x1.clone()
for i in range(8):
    x.copy_(x1)
    ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), dt, 1e-2, True, 1.0)
torch.cuda.synchronize()
print('ok')
