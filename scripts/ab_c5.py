"""Tuning aid: per-iteration time of the persistent STOMP launch at C5's per-GPU load (4096 particles x 32 samples); env
MPB_STOMP_BATCHES=1/2 selects the layout (exchange between two workgroups per particle / one workgroup, two batches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
P = int(os.environ.get('MPB_P', 4096))
wl, cost, planner = bench.make_stomp(P, 32, dev, 0)
m0 = wl['means0'].clone()
def t(n):
    planner._particle_means.copy_(m0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); planner.optimize(opt_iters=n); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
for _ in range(3): t(50)
print('P=%d: ms/iter %.4f' % (P, min(t(50) for _ in range(7)) / 50), 'cost mean %.1f' % float(planner.costs.mean()))
