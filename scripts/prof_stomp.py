"""Driver for the rocprofv3 PMC passes: exactly the launches bench.py times at C3 -- MPB_LAUNCHES persistent launches of
MPB_ITERS iterations each, every one from the initial means (no CPU baseline, no other configs).  MPB_FUSED unset: the
two-kernel path (MPB_ITERS iterations, once)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device('cuda:0')
P = int(os.environ.get('MPB_P', 128))
wl, cost, planner = bench.make_stomp(P, 32, dev, 0, H=int(os.environ.get('MPB_H', 64)))    # MPB_H=128: the generalised kernel
n = int(os.environ.get('MPB_ITERS', 200))
means_init = wl['means0'].clone()
if os.environ.get('MPB_FUSED'):
    for _ in range(int(os.environ.get('MPB_LAUNCHES', 6))):
        planner._particle_means.copy_(means_init)
        planner.optimize(opt_iters=n)
else:
    two = bench.STOMP_two_kernel(wl, cost, dev, 0, P)
    two.optimize(opt_iters=n)
torch.cuda.synchronize()
print('done', float(planner.costs.mean()))
