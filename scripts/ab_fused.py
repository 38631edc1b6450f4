"""Tuning aid: per-iteration time of the persistent STOMP launch at C3 = (t(400) - t(200)) / 200, min over 15 of each after a warm-up; MPB_LIB_PATH selects the library build (scripts/ab_fused.sh alternates base and variants)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
P, S, H = 128, 32, 64
wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=False)
d = wl['means0'].shape[-1]
cpu = dict(device='cpu', dtype=torch.float32)
R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
means0 = wl['means0'].clone()
samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
ws = ops.stomp_workspace(P, S, H, d, dev)
def run(n):
    means = means0.clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=n)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
run(50)
for _ in range(20): run(400)
t4, t2 = [], []
for _ in range(15):
    t4.append(run(400)); t2.append(run(200))
best = (min(t4) - min(t2)) / 200      # (min of each: a device stall in one run(200) must not shrink the difference)
print('us/iter %.2f' % (best * 1e3), 'cost mean', float(costs.mean()))
