import os, sys, time
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
means = means0.clone()
def run(n, reset=True, it0=0):
    if reset: means.copy_(means0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=n, iter0=it0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3, (time.perf_counter() - t0) * 1e6
run(50)
for n in (1, 2, 5, 10, 20, 50, 100, 200, 400, 800):
    r = sorted(run(n) for _ in range(5))[2]
    print(f'n={n:4d}: events {r[0]:9.1f} us  wall {r[1]:9.1f} us   per-iter {r[0]/n:7.2f}')
# continuing (no reset): iterations 800.. 
run(800)
for n in (200, 400):
    r = sorted(run(n, reset=False, it0=800) for _ in range(5))[2]
    print(f'converged state n={n:4d}: events {r[0]:9.1f} us per-iter {r[0]/n:7.2f}')
