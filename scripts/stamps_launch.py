"""Diagnostic: where the fixed part of a persistent STOMP launch goes -- s_memrealtime stamps of wave 0 of every workgroup
along a K-iteration launch at C3 (needs a -DMPB_STAMPS build: build_variants/stamps.so, MPB_LIB_PATH set)."""
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
names = ['entry', 'ticket', 'constants', 'noise0', 'iter0', 'iter1', 'iter2', 'exit']
rows = []
for rep in range(7):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    pl.optimize(opt_iters=K); torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert h.mpb_debug_read_lstamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    t = buf.reshape(1024, 8)[:256].astype(np.int64)
    t0 = t[:, 0].min()
    rows.append((t - t0) * 0.01)          # us since the first workgroup's entry
r = np.median(np.stack(rows), axis=0)     # (256, 8)
print('K = %d; us since the first workgroup entered the kernel (over the 256 workgroups: min / median / max)' % K)
for k, n in enumerate(names):
    print('  %-10s %8.2f %8.2f %8.2f' % (n, r[:, k].min(), np.median(r[:, k]), r[:, k].max()))
