"""Diagnostic: per-phase s_memtime stamps of one iteration of the persistent STOMP kernel at C3
(needs a -DMPB_STAMPS build: build_variants/stamps.so, MPB_LIB_PATH set)."""
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
ws = ops.stomp_workspace(P, S, H, d, dev)
for _ in range(3):
    ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=100)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=200)
e1.record()
torch.cuda.synchronize()
print('200 iterations: %.2f us / iteration' % (e0.elapsed_time(e1) * 1e3 / 200))
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 16 * 12, dtype=np.uint64)
assert h.mpb_debug_read_fstamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
simd = (buf.reshape(256, 16, 12)[:, :, 0] >> np.uint64(56)).astype(np.int64)
buf = buf & np.uint64(0x00FFFFFFFFFFFFFF)
t = buf.reshape(256, 16, 12).astype(np.int64)
names = ['A tile row, x = mean + noise, store', 'B cost (FK + SDF)', 'wait at barrier 1', 'C softmax stats + partial delta',
         'D publish + barrier 2 + flag', 'next noise (Philox + MFMA + tile)', 'poll + barrier 3', 'combine partials',
         'E weights, delta -> LDS, barrier 4', 'matvec', 'barrier 5']
dd = np.diff(t, axis=2).reshape(-1, 11)
for k in range(11):
    print(f'{names[k]:40s} median {np.median(dd[:, k]):8.0f}  p90 {np.percentile(dd[:, k], 90):8.0f}  max {dd[:, k].max():8.0f}')
tot = (t[:, :, 11] - t[:, :, 0]).reshape(-1)
print('iteration (stamp 0 -> 11): median', np.median(tot), 'max', tot.max(), '(shader cycles; ~2.1 GHz)')
for wg in (0, 1, 100):
    t0 = t[wg, :, 0].min()
    print(f'workgroup {wg}: wave simd | A start, B start, B end, barrier-1 release (cycles from the first wave\'s start)')
    for w in np.argsort(simd[wg] * 100 + np.arange(16)):
        print(f'   wave {w:2d} simd {simd[wg, w]} | {t[wg, w, 0] - t0:6d} {t[wg, w, 1] - t0:6d} {t[wg, w, 2] - t0:6d} {t[wg, w, 3] - t0:6d}')
