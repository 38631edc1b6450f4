"""Tuning aid: per-iteration time of the persistent STOMP launch at C3's shape with pos_only = True (d = 7: two rollouts per
noise product, mpb_stomp_noise.h stomp_noise_bf16_pair); MPB_LIB_PATH selects the library build (-DFUSED_NO_PAIR: one each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
for P, S in ((128, 32), (4096, 32)):
    H = 64
    wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=True)
    d = wl['means0'].shape[-1]
    R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, dict(device='cpu', dtype=torch.float32))
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
    ws = ops.stomp_workspace(P, S, H, d, dev)
    n = 200 if P <= 128 else 10
    def run(k):
        means = wl['means0'].clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=k)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b)
    run(n)
    for _ in range(5): run(2 * n)
    t2 = min(run(2 * n) for _ in range(9)); t1 = min(run(n) for _ in range(9))
    print('P=%d S=%d d=%d: %.2f us/iter  cost mean %.1f' % (P, S, d, (t2 - t1) / n * 1e3, float(costs.mean())))
