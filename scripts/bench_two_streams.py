"""Experiment: C3 as ONE STOMP batch (P=128) vs the same particles as TWO independent half-batches on two HIP streams
(kernel B of one half overlaps kernel A of the other)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import workloads
from motion_planning_baselines_amd.planners.stomp import STOMP
from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
dev = torch.device('cuda:0')
ta = dict(device=dev, dtype=torch.float32)


def make(P, first):
    wl = workloads.panda_spheres_stomp(P, dev, S=32, first_particle=first)
    prm = wl['params']
    H = prm['n_support_points']
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'], sigma_coll=wl['sigma_coll'],
                                                        tensor_args=ta)], tensor_args=ta)
    return STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
                 initial_particle_means=wl['means0'], tensor_args=ta, noise='philox', seed=0, particle_offset=first, **prm)


K = 400
one = make(128, 0)
one.optimize(opt_iters=500)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); one.optimize(opt_iters=K); torch.cuda.synchronize()
    print(f'one batch  P=128: {(time.perf_counter() - t0) / K * 1e6:.1f} us / iteration', flush=True)
for nsplit in (2, 4):
    parts = [make(128 // nsplit, i * (128 // nsplit)) for i in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    for p, s in zip(parts, streams):
        with torch.cuda.stream(s):
            p.optimize(opt_iters=200)
    torch.cuda.synchronize()
    for chunk in (5, 20):
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(K // chunk):
                for p, s in zip(parts, streams):
                    with torch.cuda.stream(s):
                        p.optimize(opt_iters=chunk)
            torch.cuda.synchronize()
            print(f'{nsplit} streams x P={128 // nsplit} (chunks of {chunk}): {(time.perf_counter() - t0) / K * 1e6:.1f} us / iteration of the whole batch', flush=True)
