"""Tuning aid: wall time of optimize(K) + synchronize against K, with torch.cuda.synchronize() alone and with a host spin on
an event query in front of it (does the runtime's blocking wait add a wake-up latency past some kernel duration?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=500); torch.cuda.synchronize()
def block(k, spin):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    t0 = time.perf_counter(); pl.optimize(opt_iters=k)
    if spin:
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
for k in (10, 14, 16, 18, 19, 20, 21, 22, 24, 28, 32, 40):
    a = sorted(block(k, False) for _ in range(41)); b = sorted(block(k, True) for _ in range(41))
    print('K=%3d  synchronize: min %7.1f med %7.1f max %7.1f | spin on event first: min %7.1f med %7.1f max %7.1f us'
          % (k, a[0], a[20], a[-1], b[0], b[20], b[-1]))
