"""Collision cost vs cost+gradient kernels at the C3 sample batch (B=4096, H=64, Panda)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
dev = torch.device('cuda:0')
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
B, H, D = 4096, 64, 7
g = torch.Generator().manual_seed(0)
qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
for label, spread in (('smooth trajectories (CHOMP / GPMP2-like)', 0.0), ('wide noise (STOMP C3-like)', 2.0)):
    s = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g); e = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    x = (s * (1 - a) + e * a + spread * torch.randn(B, H, D, generator=g)).contiguous().to(dev)
    for name, fn in (('cost', lambda: ops.cost_collision_eval(x, geom, 1.0)), ('cost+grad', lambda: ops.cost_collision_grad(x, geom, 1.0))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'{label}: {name:9s} {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us', flush=True)
