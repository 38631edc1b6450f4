"""Tuning aid: one iteration per optimize() call, each waited for (check='sync', the reference examples' loop) at C3."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
pl.check = 'sync'
pl.optimize(opt_iters=50); torch.cuda.synchronize()
best = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(500):
        pl.optimize(opt_iters=1)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / 500 * 1e6)
print('us per optimize(1), sync: min %.2f median %.2f' % (min(best), sorted(best)[2]))
