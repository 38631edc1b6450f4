"""STOMP iteration time against the horizon length (H = 64 has the MFMA fast path; others use the generic kernel)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril

dev = torch.device('cuda:0')
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
P, S, D = 128, 32, 7
for H in (32, 48, 64, 96, 128, 192):
    for d in (7, 14):
        cpu = dict(device='cpu', dtype=torch.float32)
        R = stomp_precision_matrix(H, 5 / H, 0.1, cpu)
        Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
        g = torch.Generator().manual_seed(0)
        qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
        s = qmin + (qmax - qmin) * torch.rand(P, 1, D, generator=g)
        e = qmin + (qmax - qmin) * torch.rand(P, 1, D, generator=g)
        a = torch.linspace(0, 1, H).reshape(1, H, 1)
        pos = s * (1 - a) + e * a
        means = (pos if d == 7 else torch.cat([pos, torch.zeros(P, H, D)], -1)).contiguous().to(dev)
        samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); w = torch.empty(P, S, device=dev)
        run = lambda n: ops.stomp_step(means, None, samples, costs, w, L, Sigma, geom, S, D, 1e6, 1.0, 0.0, 1.0, n_iters=n)
        run(10); torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
        ta = []
        for fn in (lambda: ops.stomp_sample(means, None, samples, L, S, seed=0, it=1, geom=geom, costs=costs, k_sigma=1e6),
                   lambda: ops.stomp_update(means, samples, costs, w, Sigma, 0.0, 1.0)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize(); ta.append((time.perf_counter() - t0) / 50)
        t = min(ts)
        print(f'H={H:3d} d={d:2d}: {t*1e6:7.1f} us/iter (runs {[round(x*1e6,1) for x in ts]}; A alone {ta[0]*1e6:.1f}, B alone {ta[1]*1e6:.1f})'
              f'  {t*1e9/(P*S*H):.2f} ns per rollout-waypoint', flush=True)
