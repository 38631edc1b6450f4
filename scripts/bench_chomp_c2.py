import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
dev = torch.device('cuda:0')
for label, robot, field, D in (('C2 point-mass 2D', G.RobotPointMass(2, radius=0.01), G.env_dense_2d(), 2), ('Panda + spheres', G.RobotPanda(), G.env_spheres_3d(), 7)):
    geom = ops.DeviceGeometry(robot, field, dev)
    B, H = 1024, 64
    g = torch.Generator().manual_seed(0)
    qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
    s = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g); e = qmin + (qmax - qmin) * torch.rand(B, 1, D, generator=g)
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    x0 = torch.cat([s * (1 - a) + e * a, torch.zeros(B, H, D)], -1).contiguous().to(dev)
    R = chomp_precision_matrix(0.04, H, dict(device='cpu', dtype=torch.float32)).to(dev)
    ts = []
    for rep in range(6):
        x = x0.clone()
        ops.chomp_step(x, R, geom, D, 1.0, 10.0, 1e-4, 0.05, 0.05, n_iters=500, B_global=B); torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.chomp_step(x, R, geom, D, 1.0, 10.0, 1e-4, 0.05, 0.05, n_iters=500, B_global=B); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 500 * 1e6)
    print(f'{label}: {min(ts):.2f} us/iter (runs {[round(t, 2) for t in ts]})', flush=True)
