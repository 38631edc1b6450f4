"""Small driver for rocprofv3: a few fused STOMP iterations at the C3 shape (no CPU baseline)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import stomp_precision_matrix, precision_to_scale_tril
dev = torch.device('cuda:0')
P, S, H = int(os.environ.get('MPB_P', 128)), 32, 64
wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=False)
d = wl['means0'].shape[-1]
cpu = dict(device='cpu', dtype=torch.float32)
R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
means = wl['means0'].clone()
samples = torch.empty(P, S, H, d, device=dev); costs = torch.empty(P, S, device=dev); weights = torch.empty(P, S, device=dev)
n = int(os.environ.get('MPB_ITERS', 20))
if os.environ.get('MPB_FUSED'):      # the persistent one-launch loop: a few launches of n iterations each
    ws = ops.stomp_workspace(P, S, H, d, dev)
    for _ in range(int(os.environ.get('MPB_LAUNCHES', 6))):
        ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=n)
else:
    ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, n_iters=n)
torch.cuda.synchronize()
print('done', float(costs.mean()))
