"""Tuning aid: candidate-trip statistics of the broad-phase grid on the C3 STOMP workload.  One STOMP iteration on the
GPU produces the sample trajectories; the chain walk and the cell look-ups are redone here in numpy for several cell
sizes, and the number of candidate trips a wave executes (max over its 64 waypoints x 4 spheres) is histogrammed."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads, geometry
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
P, S, H = 128, 32, 64
wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=False)
d = wl['means0'].shape[-1]
cpu = dict(device='cpu', dtype=torch.float32)
R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
means = wl['means0'].clone()
samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
ws = ops.stomp_workspace(P, S, H, d, dev)
n_it = int(os.environ.get('MPB_ITERS', 1))
ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=n_it)
torch.cuda.synchronize()
q = samples[..., :7].double().cpu().numpy().reshape(-1, 7)          # (P*S*H, 7)
robot, field = wl['robot'], wl['field']
tf = robot.joint_tf.astype(np.float64)
N = q.shape[0]
Rm = np.broadcast_to(np.eye(3), (N, 3, 3)).copy(); t = np.zeros((N, 3))
pos = np.zeros((N, len(robot.link_frame), 3)); frame = 0
for l, f in enumerate(robot.link_frame):
    while frame < f:
        A = tf[frame]
        t = t + Rm @ A[:, 3]
        Rm = Rm @ A[:, :3]
        if frame < 7:
            c, s = np.cos(q[:, frame]), np.sin(q[:, frame])
            Rz = np.zeros((N, 3, 3)); Rz[:, 0, 0] = c; Rz[:, 0, 1] = -s; Rz[:, 1, 0] = s; Rz[:, 1, 1] = c; Rz[:, 2, 2] = 1
            Rm = Rm @ Rz
        frame += 1
    pos[:, l] = t + (Rm @ robot.link_offset[l].astype(np.float64)[None, :, None])[..., 0]
sph = np.asarray(field.spheres, dtype=np.float64).reshape(-1, 4)
rl = robot.link_radius.astype(np.float64)
a_max = field.margin + rl.max()
groups = [[0, 1, 2]] + [list(range(i, min(i + 4, 31))) for i in range(3, 31, 4)]
print('obstacles', len(sph), 'radius', sph[:, 3].min(), sph[:, 3].max(), 'margin', field.margin)
def stats(cell, maxdim, per_link=False):
    geometry.GRID_CELL, geometry.GRID_MAX_DIM = cell, maxdim
    g = geometry.build_grid(sph.astype(np.float32), a_max)
    dims, lo, inv, words = g['dims'], g['lo'].astype(np.float64), g['inv'].astype(np.float64), g['words']
    ijk = np.clip(np.floor((pos - lo) * inv), 0, dims - 1).astype(np.int64)
    w = words[(ijk[..., 2] * dims[1] + ijk[..., 1]) * dims[0] + ijk[..., 0]]
    n = len(sph)
    cnt = sum((((w >> (8 * k)) & 0xFF) != n).astype(np.int64) for k in range(4))
    cnt = np.where(w == geometry.GRID_OVERFLOW, 9, cnt)
    cnt = cnt.reshape(P * S, H, 31)
    trips = []
    for gset in groups:
        trips.append(np.maximum(cnt[:, :, gset].max(axis=(1, 2)), 1))
    trips = np.stack(trips, 1)                                       # (waves, groups)
    zero = np.stack([cnt[:, :, gset].max(axis=(1, 2)) == 0 for gset in groups], 1)
    print('   fraction of (wave, group) pairs with NO candidate at all, by group:', np.round(zero.mean(0), 2), 'overall %.3f' % zero.mean())
    hist = np.bincount(cnt.reshape(-1), minlength=10) / cnt.size
    print(f'cell {cell:.3f} dims {dims} cells {dims.prod()} grid {g["stats"]}  lookup hist {np.round(hist[:6], 3)} overflow {hist[9]:.4f}'
          f'  mean trips/group {trips.mean():.3f}  (>=2: {np.mean(trips >= 2):.2f}, >=3: {np.mean(trips >= 3):.2f}, 4+: {np.mean(trips >= 4):.2f})')
for cell, md in ((0.14, 16), (0.11, 20)):
    stats(cell, md)
# what a larger LDS grid would buy (MPB_GRID_MAX_CELLS is 4096 words = 16 KB): the finest cell that fits 8 192 / 16 384 words
for mc in (8192, 16384):
    geometry.GRID_MAX_CELLS = mc
    print('GRID_MAX_CELLS', mc)
    stats(0.14, 64)
geometry.GRID_MAX_CELLS = 4096
