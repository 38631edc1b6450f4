"""Where does the GPMP2 step lose accuracy on a very stiff system (sigma_start = sigma_goal = 1e-6, sigma_gp = 1, no trust
region)?  Per-waypoint error of the step against the dense fp64 solution refined in long double.  Run with MPB_LIB_PATH set
to a -DGP_RCP_NEWTON=2 variant to compare the pivot reciprocal's second Newton step."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from motion_planning_baselines_amd import geometry as G, ops, workloads
from oracle import planners_ref as O
from oracle.geometry_ref import make_ref_geometry
from test_gpu_parity_gpmp2_mppi import _refined_solve

dev = torch.device('cuda:0')
H, B, D = int(os.environ.get('H', 128)), 2, 7
sig = tuple(float(v) for v in os.environ.get('SIG', '1e-6,1.0,1e-6,1e-5').split(','))
robot, field = G.RobotPanda(), G.env_spheres_3d()
geom = ops.DeviceGeometry(robot, field, dev)
q = workloads.collision_free_configs(robot, field, 2 * B, 100 + H, dev)
dt = 5.0 / H
gen = torch.Generator().manual_seed(H)
x0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, 'cpu')
x0[:, 1:-1, :D] += 0.05 * torch.randn(B, H - 2, D, generator=gen)
x0 = x0.float().contiguous()
z = torch.zeros(B, D)
start = torch.cat([torch.from_numpy(q[:B]), z], -1).contiguous()
goal = torch.cat([torch.from_numpy(q[B:]), z], -1).contiguous()
x = x0.clone().to(dev)
ws = ops.gpmp2_workspace(B, H, D, dev)
ops.gpmp2_step(x, start.to(dev), goal.to(dev), geom, ws, sig, dt, 1e-2, False, 1.0)
torch.cuda.synchronize()
f64 = dict(device='cpu', dtype=torch.float64)
rrobot, rfield = make_ref_geometry(robot, field, f64)
As, bs, Ks = [], [], []
for i in range(B):
    A, b, K = O.gpmp2_linear_system(x0[i:i + 1].double(), rrobot, rfield, start[i].double(), goal[i].double(), D, dt, *sig[:2], sig[2], sig[3], f64)
    As.append(A); bs.append(b); Ks.append(K)
A, b, K = torch.cat(As), torch.cat(bs), torch.cat(Ks)
rows = ops.gpmp2_collision_rows(x0.to(dev), geom).cpu().double()
N, dim = 2 * D * H, 2 * D
r0 = N + dim
for i in range(H - 1):
    A[:, r0 + i, (i + 1) * dim:(i + 1) * dim + D] = rows[0, :, i + 1, :D]
    b[:, r0 + i, 0] = rows[0, :, i + 1, D]
JtJ, g = O.gpmp2_normal_equations(A, b, K, 1e-2, False)
l, _ = torch.linalg.cholesky_ex(JtJ)
d0 = torch.cholesky_solve(g, l).view(B, H, dim)
d1 = _refined_solve(JtJ, g, l, True).view(B, H, dim)
dg = x.cpu().double() - x0.double()
print('lib', os.environ.get('MPB_LIB_PATH', 'product'), 'sig', sig)
print('dense unrefined vs refined', float((d0 - d1).abs().max() / d1.abs().max()))
print('gpu vs refined           ', float((dg - d1).abs().max() / d1.abs().max()), ' |step| max', float(d1.abs().max()))
e = (dg - d1).abs().amax(dim=(0, 2)) / d1.abs().max()
print('per waypoint (every 8th):', ' '.join('%.1e' % v for v in e[::8].tolist()), 'last', '%.1e' % e[-1])
# the fp32 storage of x: the step the kernel computed is only visible through x0 + step rounded to fp32
print('fp32 storage resolution of x relative to the step:', float((x0.abs().max() * 2 ** -24) / d1.abs().max()))
