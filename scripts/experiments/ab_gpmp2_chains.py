"""Tuning aid: one GPMP2 iteration (B particles, H=128, D=7) with the solve's chains-per-wave forced (MPB_GPMP2_CHAINS =
1: gpmp2_solve_kernel; 2 | 4: gpmp2_solve_mc_kernel), event-timed from the same state, and the results compared bit for bit.
    python scripts/ab_gpmp2_chains.py [B ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
dev = torch.device('cuda:0')
H, D = 128, 7
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
for B in [int(a) for a in sys.argv[1:]] or [2048]:
    g = torch.Generator().manual_seed(0)
    qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
    s = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
    e = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    x0 = torch.cat([s * (1 - a) + e * a, ((e - s) / ((H - 1) * 5 / 128)).expand(B, H, D)], -1).contiguous().to(dev)
    start = torch.cat([s[:, 0], torch.zeros(B, D)], -1).contiguous().to(dev)
    goal = torch.cat([e[:, 0], torch.zeros(B, D)], -1).contiguous().to(dev)
    ws = ops.gpmp2_workspace(B, H, D, dev)
    ref = None
    for rnd in range(2):
        for nch in (1, 2, 4):
            os.environ['MPB_GPMP2_CHAINS'] = str(nch)
            x = x0.clone()
            ts = []
            for i in range(12):
                x.copy_(x0); torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), 5 / 128, 1e-2, True, 1.0)
                a1.record(); torch.cuda.synchronize()
                if i >= 2: ts.append(a0.elapsed_time(a1))
            if ref is None: ref = x.clone()
            print('B=%5d chains/wave %d: min %.4f ms  median %.4f ms   same bits as 1 chain: %s  finite: %s'
                  % (B, nch, min(ts), sorted(ts)[len(ts) // 2], bool(torch.equal(x, ref)), bool(torch.isfinite(x).all())), flush=True)
