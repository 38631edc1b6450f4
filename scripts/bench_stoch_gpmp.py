"""StochGPMP at a C3-like shape (P=128 particles x S=32 samples, H=64, D=7), device Philox noise."""
import gc, os, sys, time
if os.environ.get("MPB_NOGC"): gc.disable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops
from motion_planning_baselines_amd.planners.stoch_gpmp import StochGPMP
dev = torch.device('cuda:0')
ta = dict(device=dev, dtype=torch.float32)
robot, field = G.RobotPanda(), G.env_spheres_3d()
P, S, H, D = int(os.environ.get('SG_P', 128)), 32, 64, 7
g = torch.Generator().manual_seed(0)
qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
s = qmin + (qmax - qmin) * torch.rand(D, generator=g)
e = qmin + (qmax - qmin) * torch.rand(D, generator=g)
a = torch.linspace(0, 1, H).reshape(1, H, 1)
means = torch.cat([(s * (1 - a) + e * a).expand(P, H, D), torch.zeros(P, H, D)], -1).contiguous().to(dev)
pl = StochGPMP(robot=robot, n_dof=D, n_support_points=H, num_particles_per_goal=P, opt_iters=1, dt=5 / 64,
               start_state=s.to(dev), multi_goal_states=e[None].to(dev), initial_particle_means=means, step_size=0.5,
               sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0, sigma_start_sample=1e-3,
               sigma_goal_sample=1e-3, sigma_gp_sample=0.2, num_samples=S, temperature=1.0, collision_fields=[field],
               sigma_start=1e-3, sigma_gp=1.0, sigma_coll=1e-2, sigma_goal_prior=1e-3, tensor_args=ta, noise='philox')
pl.optimize(opt_iters=400); torch.cuda.synchronize()      # long enough for the clocks to settle (bench.py pre-heats likewise)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pl.optimize(opt_iters=100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
t = min(ts)
print(f'StochGPMP P={P} S={S} H={H} D={D}: {t*1e6:.1f} us/iter (best of 5 x 100 iterations; all: ' + ', '.join(f'{v*1e6:.0f}' for v in ts) + ')')
flat = pl.state_samples.reshape(P * S, H, 2 * D)
def tm(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('  sample  %.1f us' % tm(lambda: pl._sample()))
print('  costs   %.1f us' % tm(lambda: ops.stoch_gpmp_costs(flat, pl._particle_means, pl._start, pl._goal, pl.geom, pl.costs, S, pl.sig_cost, pl.sig_sample, pl.dt, pl.temperature)))
print('  update  %.1f us' % tm(lambda: ops.stomp_update(pl._particle_means, pl.state_samples, pl.costs, pl._weights_buf, None, 0.0, 1.0)))
