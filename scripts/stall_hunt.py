"""Look for sporadic long batches: time many batches of 100 STOMP iterations, separating the host time spent inside
the C call (launch submission) from the time until the GPU drains."""
import os, sys, time
import torch
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
run = lambda n: ops.stomp_step(means, None, samples, costs, w, L, Sigma, geom, S, 7, 1e6, 1.0, 0.0, 1.0, n_iters=n)
run(20); torch.cuda.synchronize()
N = int(os.environ.get('NB', 400))
rec = []
t_start = time.perf_counter()
for b in range(N):
    t0 = time.perf_counter(); run(100); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rec.append((t0 - t_start, t1 - t0, t2 - t0))
tot = sorted(r[2] for r in rec)
print(f'batches {N}: median {tot[N//2]*1e3:.2f} ms, p99 {tot[int(N*0.99)]*1e3:.2f} ms, max {tot[-1]*1e3:.2f} ms')
for at, host, total in rec:
    if total > 1.5 * tot[N // 2]:
        print(f'  outlier at t={at:7.3f}s: host-side call {host*1e3:7.2f} ms, until drained {total*1e3:7.2f} ms')
