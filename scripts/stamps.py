"""Diagnostic: per-phase s_memtime stamps of kernel A (needs build_variants/stamps.so, MPB_LIB_PATH set)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads, _lib
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
ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, n_iters=30)
torch.cuda.synchronize()
ops.stomp_sample(means, None, samples, L, S, seed=0, it=99, geom=geom, costs=costs, k_sigma=1e6)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(4096 * 8, dtype=np.uint64)
assert h.mpb_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(4096, 8).astype(np.int64)
t0 = t[:, 0].min()
names = ['entry', 'L staged + barrier', 'philox', 'mfma + tile write', 'barrier', 'row + store', 'grid staged (2 barriers)', 'cost + reduce']
print('kernel span (cycles @100MHz ticks?):', t[:, 7].max() - t0)
d_ = np.diff(t, axis=1)
print('start skew: median', np.median(t[:, 0] - t0), 'max', (t[:, 0] - t0).max())
for k in range(7):
    print(f'{names[k + 1]:28s} median {np.median(d_[:, k]):9.0f}  p90 {np.percentile(d_[:, k], 90):9.0f}  max {d_[:, k].max():9.0f}')
print('wave total median', np.median(t[:, 7] - t[:, 0]), 'max', (t[:, 7] - t[:, 0]).max())
