"""Per-kernel timing of the STOMP path at the C3 shape (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril

dev = torch.device('cuda:0')
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3

import os
QUICK = os.environ.get('MPB_QUICK')
for pos_only in ((False,) if QUICK else (False, True)):
    for (P, S) in (((128, 32),) if QUICK else ((128, 32), (1024, 32), (4096, 32))):
        wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=pos_only)
        H, d = 64, wl['means0'].shape[-1]
        cpu = dict(device='cpu', dtype=torch.float32)
        R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
        Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
        geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
        means = wl['means0'].clone()
        samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
        it = [0]
        def full():
            it[0] += 1
            ops.stomp_sample(means, None, samples, L, S, seed=0, it=it[0], geom=geom, costs=costs, k_sigma=1e6)
        def sample_only():
            it[0] += 1
            ops.stomp_sample(means, None, samples, L, S, seed=0, it=it[0])
        flat = samples.flatten(0, 1)
        def cost_only():
            ops.cost_collision_eval(flat, geom, 1e6)
        def update():
            ops.stomp_update(means, samples, costs, weights, Sigma, 0.0, 1.0)
        def step10():
            ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.0, 1.0, n_iters=10)
        print(f'P={P} S={S} d={d}: A(sample+cost) med/min us {timeit(full)}, sample-only {timeit(sample_only)}, '
              f'cost-only {timeit(cost_only)}, B(update) {timeit(update)}, fused-step/iter {tuple(t / 10 for t in timeit(step10, n=20))}', flush=True)
