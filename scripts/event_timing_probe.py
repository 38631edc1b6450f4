"""How well do HIP events around single in-situ launches of kernel A (sample -> update pairs) recover its duration?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
P, S, H = 128, 32, 64
wl = workloads.panda_spheres_stomp(P, dev, S=S)
d = wl['means0'].shape[-1]
cpu = dict(device='cpu', dtype=torch.float32)
R = stomp_precision_matrix(H, 5 / 64, 0.1, cpu)
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
means = wl['means0'].clone()
samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); w = torch.empty(P, S, device=dev)
ops.stomp_step(means, None, samples, costs, w, L, Sigma, geom, S, 7, 1e6, 1.0, 0.0, 1.0, n_iters=500); torch.cuda.synchronize()
n = 100
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for i in range(n):
    ev[i][0].record()
    ops.stomp_sample(means, None, samples, L, S, seed=0, it=i, geom=geom, costs=costs, k_sigma=1e6)
    ev[i][1].record()
    ops.stomp_update(means, samples, costs, w, Sigma, 0.0, 1.0)
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f'in-situ events around kernel A: median {ts[n//2]:.1f} us, mean {sum(ts)/n:.1f}, min {ts[0]:.1f}, p90 {ts[int(n*0.9)]:.1f}')
