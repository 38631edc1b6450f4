"""Host-side overhead of the drop-in classes: optimize(opt_iters=1) called in a Python loop (how the reference's
examples drive the planners) against one call with opt_iters=K."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import workloads
from motion_planning_baselines_amd.planners.stomp import STOMP
from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite

dev = torch.device('cuda:0')
ta = dict(device=dev, dtype=torch.float32)
for P in (4, 128):
    wl = workloads.panda_spheres_stomp(P, dev, S=32)
    prm = wl['params']
    H = prm['n_support_points']
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'], sigma_coll=wl['sigma_coll'],
                                                        tensor_args=ta)], tensor_args=ta)
    pl = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
               initial_particle_means=wl['means0'], tensor_args=ta, noise='philox', seed=0, **prm)
    K = 500
    pl.optimize(opt_iters=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pl.optimize(opt_iters=K); torch.cuda.synchronize(); t_one = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        pl.optimize()
    torch.cuda.synchronize(); t_loop = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        pl._run_optimization(1)
    torch.cuda.synchronize(); t_run = (time.perf_counter() - t0) / K
    print(f'P={P}: one call of {K} iters {t_one*1e6:.1f} us/it; optimize() x{K} {t_loop*1e6:.1f} us/it; '
          f'_run_optimization(1) x{K} {t_run*1e6:.1f} us/it', flush=True)
