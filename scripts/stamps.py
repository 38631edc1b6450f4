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
ka_ms, kb_ms = ops.stomp_step_profile(means, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, n_iters=1, seed=0, iter0=99)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(4096 * 10, dtype=np.uint64)
assert h.mpb_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
tt = buf.reshape(4096, 10).astype(np.int64)
t = tt[:, :8]
rt = tt[:, 8:]
t0 = t[:, 0].min()
names = ['entry', 'L + grid loads issued', 'philox', 'L -> LDS, barrier, mfma', 'barrier', 'tile, row + store', 'to the cost loop', 'cost + reduce']
life = (t[:, 7] - t[:, 0]).astype(np.float64)
rl = (rt[:, 1] - rt[:, 0]).astype(np.float64)       # 100 MHz ticks
print('dispatch-event duration %.2f us; wave life: median %.2f us, max %.2f us (s_memrealtime); kernel span first entry -> last exit %.2f us'
      % (ka_ms * 1e3, np.median(rl) / 100, rl.max() / 100, (rt[:, 1].max() - rt[:, 0].min()) / 100))
print('shader clock during the kernel: median %.3f GHz (d s_memtime / d s_memrealtime)' % np.median(life / rl * 0.1))
print('entry skew (s_memrealtime): p50 %.2f us, p99 %.2f us, max %.2f us' % tuple(np.percentile(rt[:, 0] - rt[:, 0].min(), [50, 99, 100]) / 100))
d_ = np.diff(t, axis=1)
print('start skew: median', np.median(t[:, 0] - t0), 'max', (t[:, 0] - t0).max())
for k in range(7):
    print(f'{names[k + 1]:28s} median {np.median(d_[:, k]):9.0f}  p90 {np.percentile(d_[:, k], 90):9.0f}  max {d_[:, k].max():9.0f}')
print('wave total median', np.median(t[:, 7] - t[:, 0]), 'max', (t[:, 7] - t[:, 0]).max())
