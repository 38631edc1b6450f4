"""Tuning aid: C4 GPMP2 (B=2048) and C3 STOMP on the generic (table-driven) chain walk; MPB_LIB_PATH selects the build."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from motion_planning_baselines_amd import ops
dev = torch.device('cuda:0')
# STOMP, generic walk: the same planner with the geometry packed without the model id
wl, cost, planner = bench.make_stomp(128, 32, dev, 0)
cc = cost.cost_l[0]
cc._geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev, use_model=False)   # (device_geometry() keeps it)
m0 = wl['means0'].clone()
def t(n):
    planner._particle_means.copy_(m0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); planner.optimize(opt_iters=n); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
for _ in range(5): t(200)
print('STOMP C3 generic walk us/iter %.2f' % (min(t(200) for _ in range(10)) / 200 * 1e3))
r = bench.bench_c4(dev, 20, with_cpu=False)
print('GPMP2 C4 ms/iter %.4f' % r['ms_per_step'])
