"""Diagnostic (MPB_GP_STAMPS build of the library): cycles per phase of the GPMP2 elimination loop, wave 0 of block 0."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, _lib
dev = torch.device('cuda:0')
names = ['top: LDS rows + prefetch issue + sync', 'GP factor', 'assembly + r + sync', 'Gauss-Jordan', 'W->LDS, z = W r, stores', 'next tile + carry']
for B in (1, 256, 2048):
    H, D = 128, 7
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
    for _ in range(3):
        ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), 5 / 128, 1e-2, True, 1.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    _lib.lib().mpb_debug_read_gp_phases.argtypes = [ctypes.c_void_p]
    assert _lib.lib().mpb_debug_read_gp_phases(buf) == 0
    steps = (H - 1) // 2 + 1
    tot = sum(buf[:6])
    print(f'B={B}: {tot / steps:.0f} s_memtime ticks per waypoint step (wave 0: {steps} steps)')
    for n, v in zip(names, buf[:6]):
        print(f'   {n:42s} {v / steps:8.0f}  {100 * v / tot:5.1f} %')
