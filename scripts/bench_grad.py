"""Tuning aid: the gradient evaluators on the Panda with the compile-time model and with the table-driven walk
(DeviceGeometry(use_model=False)): CHOMP B=1024 per iteration, GPMP2 linearisation at C4 (B=2048, H=128), stand-alone
cost + gradient at B=4096."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, workloads
from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
dev = torch.device('cuda:0')
robot, field = G.RobotPanda(), G.env_spheres_3d()
def ev(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e3
for use_model in (True, False):
    geom = ops.DeviceGeometry(robot, field, dev, use_model=use_model)
    B, H = 1024, 64
    q = workloads.collision_free_configs(robot, field, 2 * B, 5, dev)
    m0 = workloads.straight_line_means(q[:B], q[B:], H, 5 / H, False, dev)
    R = chomp_precision_matrix(dt=5 / H, n_support_points=H, tensor_args=dict(device='cpu', dtype=torch.float32)).to(dev).contiguous()
    def chomp(k):
        m = m0.clone()
        return lambda: ops.chomp_step(m, R, geom, 7, 1.0, 10.0, 1e-4, 0.05, 0.05, n_iters=k)
    t = (ev(chomp(200)) - ev(chomp(100))) / 100
    B4, H4 = 2048, 128
    q4 = workloads.collision_free_configs(robot, field, 2 * B4, 23, dev)
    x4 = workloads.straight_line_means(q4[:B4], q4[B4:], H4, 5 / H4, False, dev)
    ws = ops.gpmp2_workspace(B4, H4, 7, dev)
    tl = ev(lambda: ops.gpmp2_linearize(x4, geom, ws))
    xg = workloads.straight_line_means(q4[:4096 // 2 * 2][:2048], q4[2048:], 64, 5 / 64, False, dev)
    xg = torch.cat([xg, xg])
    tg = ev(lambda: ops.cost_collision_grad(xg, geom, 1.0))
    print('model=%d: CHOMP-Panda B=1024 %.2f us/iter | GPMP2 linearise C4 %.1f us | cost+grad B=4096 H=64 %.1f us' % (use_model, t, tl, tg), flush=True)
