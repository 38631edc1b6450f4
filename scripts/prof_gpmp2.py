"""GPMP2 at the C4 shape for rocprofv3 (a few iterations)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
dev = torch.device('cuda:0')
B, H, D = int(os.environ.get('GP_B', 2048)), 128, 7
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
g = torch.Generator().manual_seed(0)
qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
s = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
e = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
a = torch.linspace(0, 1, H).reshape(1, H, 1)
x = torch.cat([s * (1 - a) + e * a, ((e - s) / ((H - 1) * 5 / 128)).expand(B, H, D)], -1).contiguous().to(dev)
start = torch.cat([s[:, 0], torch.zeros(B, D)], -1).contiguous().to(dev)
goal = torch.cat([e[:, 0], torch.zeros(B, D)], -1).contiguous().to(dev)
ws = ops.gpmp2_workspace(B, H, D, dev)
costs = torch.empty(B, device=dev)
for _ in range(5):
    ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), 5 / 128, 1e-2, True, 1.0, n_iters=1, costs_out=costs)
torch.cuda.synchronize()
print('ok', float(costs.mean()))
