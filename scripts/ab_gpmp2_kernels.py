"""Tuning aid: per-kernel durations (hipEvent pairs around each launch are too coarse: rocprofv3 instead) -- here simply the event-timed
whole iteration at C4 plus, with MPB_ONE=1, the same on the single particle with the most active rows (pure latency).  MPB_LIB_PATH selects
the build."""
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
z = torch.zeros(B, D, device=dev)
start = torch.cat([torch.from_numpy(q[:B]).to(dev), z], -1).contiguous()
goal = torch.cat([torch.from_numpy(q[B:]).to(dev), z], -1).contiguous()
if os.environ.get('MPB_ONE'):
    rows = ops.gpmp2_collision_rows(x0, geom)[0]
    pick = int(torch.argmax((rows[..., :D].abs().sum(-1) > 0).to(torch.float32).sum(1)))
    x0, start, goal, B = x0[pick:pick + 1].contiguous(), start[pick:pick + 1].contiguous(), goal[pick:pick + 1].contiguous(), 1
ws = ops.gpmp2_workspace(B, H, D, dev)
x = x0.clone()
ts = []
for i in range(14):
    x.copy_(x0); torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), dt, 1e-2, True, 1.0)
    a1.record(); torch.cuda.synchronize()
    if i >= 4: ts.append(a0.elapsed_time(a1))
print('%-28s B=%4d one iteration: min %.4f ms median %.4f ms' % (os.path.basename(os.environ.get('MPB_LIB_PATH', 'product')), B, min(ts), sorted(ts)[len(ts) // 2]))
