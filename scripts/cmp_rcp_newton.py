"""Tuning aid: the GPMP2 update with one Newton step in the pivot reciprocal (the product build) against two (a variant built
with -DGP_RCP_NEWTON=2 at build_variants/rcp2.so), on B = 512 random problems; one child process per library."""
import os, sys, subprocess, numpy as np
child = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from motion_planning_baselines_amd import geometry as G, ops
dev = torch.device('cuda:0')
B, H, D = 512, 128, 7
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
x0 = x.clone()
ops.gpmp2_step(x, start, goal, geom, ws, (1e-5, 1e-2, 1e-5, 1e-5), 5 / 128, 1e-2, bool(int(sys.argv[2])), 1.0)
torch.cuda.synchronize()
np.savez(sys.argv[1], x=x.cpu().numpy(), x0=x0.cpu().numpy())
'''
for trust in (1, 0):
    out = {}
    for name, lib in (('two', os.getcwd() + '/build_variants/rcp2.so'), ('one', None)):
        env = dict(os.environ)
        if lib: env['MPB_LIB_PATH'] = lib
        f = '/tmp/rcp_%s_%d.npz' % (name, trust)
        subprocess.run([sys.executable, '-c', child, f, str(trust)], env=env, check=True)
        out[name] = np.load(f)
    d2, d1, x0 = out['two']['x'], out['one']['x'], out['two']['x0']
    step = np.abs(d2 - x0).max()
    print('trust_region=%d: max |x_one - x_two| = %.3e   (max |x| %.3f, max step %.3e, relative to the step %.3e; fp32 storage eps 6e-8)'
          % (trust, np.abs(d1 - d2).max(), np.abs(d2).max(), step, np.abs(d1 - d2).max() / step))
