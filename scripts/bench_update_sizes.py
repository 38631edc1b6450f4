"""How the update kernel's time depends on the number of samples it has to read (P=128, H=64, d=14)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import ops
dev = torch.device('cuda:0')
P, H, d = 128, 64, 14
Sigma = torch.eye(H, device=dev)
for S in (4, 8, 16, 32):
    means = torch.zeros(P, H, d, device=dev); samples = torch.randn(P, S, H, d, device=dev)
    costs = torch.rand(P, S, device=dev); w = torch.empty(P, S, device=dev)
    fn = lambda: ops.stomp_update(means, samples, costs, w, Sigma, 0.0, 1.0)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'S={S:2d}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per back-to-back launch')
