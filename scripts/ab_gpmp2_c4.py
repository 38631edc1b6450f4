"""Tuning aid: one GPMP2 iteration at C4 on bench.py's workload (collision-free start / goal configurations, straight-line means),
event-timed from the same state every time; the active collision rows per particle next to it.  MPB_GPMP2_FORM = lr / block selects
the form of the solve, MPB_LIB_PATH the build.    python scripts/ab_gpmp2_c4.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, workloads
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H, D = 128, 7
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
sig = (1e-5, 1e-2, 1e-5, 1e-5)
rows = ops.gpmp2_collision_rows(x0, geom)[0]
n_act = (rows[..., :D].abs().sum(-1) > 0).to(torch.float32).sum(1)
print('active rows per particle: mean %.1f median %d p90 %d p99 %d max %d' % (
    float(n_act.mean()), int(n_act.median()), int(n_act.quantile(0.9)), int(n_act.quantile(0.99)), int(n_act.max())))
x = x0.clone()
ts = []
for i in range(12):
    x.copy_(x0); torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    ops.gpmp2_step(x, start, goal, geom, ws, sig, dt, 1e-2, True, 1.0)
    a1.record(); torch.cuda.synchronize()
    if i >= 2: ts.append(a0.elapsed_time(a1))
print('GPMP2 C4 (bench workload, B=%d, form %s) one iteration: min %.4f ms  median %.4f ms' % (
    B, os.environ.get('MPB_GPMP2_FORM', 'launcher'), min(ts), sorted(ts)[len(ts) // 2]))
# ten iterations in a row (the trajectories leave the obstacles: the active sets shrink)
x.copy_(x0); torch.cuda.synchronize()
a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a0.record()
ops.gpmp2_step(x, start, goal, geom, ws, sig, dt, 1e-2, True, 1.0, n_iters=10)
a1.record(); torch.cuda.synchronize()
rows = ops.gpmp2_collision_rows(x, geom)[0]
n_act = (rows[..., :D].abs().sum(-1) > 0).to(torch.float32).sum(1)
print('ten iterations in one call: %.4f ms per iteration; active rows afterwards: mean %.1f max %d' % (a0.elapsed_time(a1) / 10, float(n_act.mean()), int(n_act.max())))
