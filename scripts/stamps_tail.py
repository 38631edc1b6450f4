"""Diagnostic: who is late at the end of a persistent STOMP launch?  Per-workgroup speed (exit - iter2) / (K - 3) of two launches
(needs a -DMPB_STAMPS build: build_variants/stamps.so, MPB_LIB_PATH set): by XCD (blockIdx % 8), and the correlation of a workgroup's
speed between launches (the ticket -> particle assignment changes from launch to launch, the workgroup -> CU placement does not)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from motion_planning_baselines_amd import _lib
dev = torch.device('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=500); torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
runs = []
for rep in range(6):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    pl.optimize(opt_iters=K); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert h.mpb_debug_read_lstamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    t = buf.reshape(1024, 8)[:256].astype(np.int64)
    runs.append(((t[:, 7] - t[:, 6]) * 0.01 / (K - 3), (t[:, 7] - t[:, 0].min()) * 0.01))
sp = np.stack([r[0] for r in runs])          # (runs, 256) us per iteration
ex = np.stack([r[1] for r in runs])
print('us / iteration per workgroup: median %.3f, min %.3f, max %.3f (max / median %.3f)' % (np.median(sp), sp.min(1).mean(), sp.max(1).mean(), (sp.max(1) / np.median(sp, 1)).mean()))
print('by XCD (blockIdx %% 8), mean over runs:', np.round(np.stack([sp[:, x::8].mean() for x in range(8)]), 3))
c = np.corrcoef(sp)
print('correlation of the per-workgroup speed between launches (same blockIdx):', np.round(c[0, 1:], 2))
# pairs: blocks 2k, 2k+1 are usually partners?  speed of partners is identical by construction (they wait for each other)
order = np.argsort(sp.mean(0))
print('slowest 8 blocks (mean over runs):', order[-8:], np.round(sp.mean(0)[order[-8:]], 3))
print('fastest 8 blocks:', order[:8], np.round(sp.mean(0)[order[:8]], 3))
print('exit spread (us) per run: ', np.round(ex.max(1) - ex.min(1), 1), ' last - median:', np.round(ex.max(1) - np.median(ex, 1), 1))
