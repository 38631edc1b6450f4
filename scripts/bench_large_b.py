"""Large-B sweep of the STOMP iteration: tensors far beyond the 256 MB Infinity Cache."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
S, H = 32, 64
for P in (128, 1024, 8192, 16384, 32768):
    wl = workloads.panda_spheres_stomp(min(P, 1024), dev, S=S, pos_only=False)
    m0 = wl['means0']
    means = m0.repeat((P + m0.shape[0] - 1) // m0.shape[0], 1, 1)[:P].contiguous()
    d = means.shape[-1]
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
    n = 10
    ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, n_iters=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, n_iters=n)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    B = P * S
    alg = 4 * (B * H * d + 2 * P * H * d + 2 * B)
    print(f'P={P:6d} B={B:8d} samples {B*H*d*4/1e6:8.1f} MB: {t*1e6:9.1f} us/iter, {1/t:9.1f} it/s, {B/t/1e6:7.1f} M rollouts/s, '
          f'algorithmic {alg/t/1e9:7.1f} GB/s ({alg/t/8e12*100:.1f}% of 8 TB/s)', flush=True)
    del samples
