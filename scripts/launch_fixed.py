"""Tuning aid: where the fixed part of a persistent STOMP call goes -- event time and wall time of optimize(K) for small K
(min / median of N runs each, every run from the initial means after a synchronize, like bench.py's timed block)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=500); torch.cuda.synchronize()
def block(k):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); pl.optimize(opt_iters=k); b.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6, a.elapsed_time(b) * 1e3
def block_noev(k):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    t0 = time.perf_counter(); pl.optimize(opt_iters=k); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
for k in (0, 1, 2, 3, 5, 10, 20, 40, 80):
    r = sorted(block(k) for _ in range(31))
    ev = sorted(x[1] for x in r)
    nv = sorted(block_noev(k) for _ in range(31))
    print('K=%3d  wall min %7.1f med %7.1f | events min %7.1f med %7.1f | wall without events min %7.1f med %7.1f us'
          % (k, r[0][0], r[15][0], ev[0], ev[15], nv[0], nv[15]))
