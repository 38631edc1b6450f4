import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
for P in (128, 4096):
    S, H = 32, 64
    wl = workloads.panda_spheres_stomp(P, dev, S=S)
    d = wl['means0'].shape[-1]
    R = stomp_precision_matrix(H, 5 / 64, 0.1, dict(device='cpu', dtype=torch.float32))
    L = precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    means = wl['means0'].clone(); samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev)
    fn = lambda i: ops.stomp_sample(means, None, samples, L, S, seed=0, it=i, geom=geom, costs=costs, k_sigma=1e6)
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): fn(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print(f'P={P}: kernel A {best:.1f} us', flush=True)
