"""Tuning aid: the host-side segments of one timed block of bench.py's protocol at C3 -- optimize(K) entry -> the C launch call
-> launch returned -> optimize() returned (the wait) -> torch.cuda.synchronize() returned; medians over N blocks."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from motion_planning_baselines_amd import ops
dev = torch.device('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl, cost, pl = bench.make_stomp(128, 32, dev, 0)
m0 = pl._particle_means.clone()
pl.optimize(opt_iters=500); torch.cuda.synchronize()
marks = {}
orig = ops.StompRunPlan.launch
def launch(self, *a, **k):
    marks['pre'] = time.perf_counter()
    r = orig(self, *a, **k)
    marks['post'] = time.perf_counter()
    return r
ops.StompRunPlan.launch = launch
rows = []
for _ in range(41):
    pl._particle_means.copy_(m0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pl.optimize(opt_iters=K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    span = pl._status.device_span_ms() * 1e3
    rows.append(((marks['pre'] - t0) * 1e6, (marks['post'] - marks['pre']) * 1e6, (t1 - marks['post']) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6, span))
import numpy as np
r = np.median(np.array(rows), 0)
print('K = %d: python before the launch call %.1f | launch call %.1f | wait in optimize() %.1f | synchronize %.1f | block %.1f us; device span of the launch %.1f us'
      % (K, *r))
