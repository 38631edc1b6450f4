"""Where does an MPPI iteration go?  bench.py's `mppi` entry (1 024 point-mass problems, S = 32, T = 64, c = 2) with and without
the collision field, and at other problem counts."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
from motion_planning_baselines_amd.planners.priors.gaussian import const_ctrl_Cov

dev = torch.device('cuda:0')
S, T, c = 32, 64, 2
f = lambda a: torch.as_tensor(a, dtype=torch.float32).contiguous().to(dev)
Cov = const_ctrl_Cov([0.3, 0.3], T, c, dict(device='cpu', dtype=torch.float32))
tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
geom = ops.DeviceGeometry(G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d(), dev)


def bench(NP, with_geom, steps=50):
    gen = torch.Generator().manual_seed(0)
    state0 = f(torch.rand(NP, c, generator=gen) * 0.2 - 0.9)
    goal = f(torch.rand(NP, c, generator=gen) * 0.2 + 0.7)
    mean = torch.zeros(NP, T, c, device=dev)
    controls, states = torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, T, c, device=dev)
    costs, weights = torch.empty(NP, S, device=dev), torch.empty(NP, S, device=dev)

    def run(k):
        mean.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.mppi_step(mean, None, tril, cinv, state0, goal, f([-1., -1.]), f([1., 1.]), torch.ones(T, device=dev),
                      f([1., 1., 1., 100.]), geom if with_geom else None, controls, states, costs, weights, 0.04, k_sigma=1e6, weight=1.0,
                      temp=1.0, step_size=0.7, n_iters=k, seed=3)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    for _ in range(6):
        run(steps)
    t = sorted(run(steps) for _ in range(7))[3]
    return 1e6 * t / steps


for NP in (1, 256, 512, 1024, 4096):
    a, b = bench(NP, True), bench(NP, False)
    print('NP = %5d: %.1f us / iteration with the collision field (%.2f M problem-iterations/s), %.1f without' % (NP, a, NP / a, b))
