"""Timings of the other BASELINE configs (parity-test cases, not bench lines): C1, C2, C4, MPPI example."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motion_planning_baselines_amd import geometry as G, ops, workloads
from motion_planning_baselines_amd.planners.stomp import STOMP
from motion_planning_baselines_amd.planners.chomp import CHOMP
from motion_planning_baselines_amd.planners.gpmp2 import GPMP2
from motion_planning_baselines_amd.planners.mppi import MPPI, PointParticleDynamics
from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite

dev = torch.device('cuda:0')
ta = dict(device=dev, dtype=torch.float32)

def wall(fn, n_inner, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) / n_inner

# C1: STOMP point mass, B=16
wl = workloads.pointmass_grid_circles_stomp(dev)
cost = CostComposite(wl['robot'], 64, [CostCollision(wl['robot'], 64, field=wl['field'], sigma_coll=1e-3, tensor_args=ta)], tensor_args=ta)
pl = STOMP(opt_iters=1, start_state=torch.tensor([-0.8, -0.8], device=dev), cost=cost, initial_particle_means=wl['means0'], tensor_args=ta, **wl['params'])
t = wall(lambda: pl.optimize(opt_iters=200), 200)
print(f'C1 STOMP pointmass B=16 H=64 d=4: {t*1e6:.1f} us/iter = {1/t:.0f} it/s (reference CPU in-container: 1624 it/s)')

# C2: CHOMP B=1024
wl = workloads.pointmass_dense_chomp(1024, dev)
cost = CostComposite(wl['robot'], 64, [CostCollision(wl['robot'], 64, field=wl['field'], sigma_coll=1.0, tensor_args=ta)], weights_cost_l=[10.0], tensor_args=ta)
pl = CHOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost, initial_particle_means=wl['means0'], tensor_args=ta, **wl['params'])
t1 = wall(lambda: pl.optimize(opt_iters=1), 1, reps=20)
t = wall(lambda: pl.optimize(opt_iters=500), 500)
print(f'C2 CHOMP B=1024 H=64 D=2 d=4: {t*1e6:.2f} us/iter fused (500 iters/launch) = {1/t:.0f} it/s; single-iteration call {t1*1e6:.1f} us (reference CPU in-container: 119 it/s)')

# C4: GPMP2 B=2048 H=128 D=7
for B, H in ((2048, 128), (256, 128), (2048, 64)):
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
    dt = 5.0 / H
    means0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
    means0[:, 0, 7:] = 0; means0[:, -1, 7:] = 0
    pl = GPMP2(robot=robot, n_dof=7, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt,
               start_state=torch.from_numpy(q[0]).to(dev), multi_goal_states=torch.from_numpy(q[B:B + 1]).to(dev),
               initial_particle_means=means0, solver_params=dict(delta=1e-2, trust_region=True, method='cholesky'),
               collision_fields=[field], tensor_args=ta)
    pl.set_problem_states(torch.from_numpy(q[:B]).to(dev), torch.from_numpy(q[B:]).to(dev))
    c0 = None
    def run():
        pl.optimize(opt_iters=1)
    t = wall(run, 1, reps=5)
    print(f'C4-like GPMP2 B={B} H={H} D=7 (N={14*H}): {t*1e3:.3f} ms/iter = {1/t:.1f} it/s; costs {float(pl.costs.mean()):.4g}; workspace {pl._ws.numel()/1e6:.0f} MB', flush=True)

# MPPI example shape
S, T = 32, 64
system = PointParticleDynamics(rollout_steps=T, control_dim=2, state_dim=2, dt=0.04, discount=1., goal_state=torch.tensor([0.8, 0.8]),
                               ctrl_min=[-100, -100], ctrl_max=[100, 100], c_weights={'pos': 1., 'vel': 1., 'ctrl': 1., 'pos_T': 1000., 'vel_T': 0.}, tensor_args=ta)
pl = MPPI(system, num_ctrl_samples=S, rollout_steps=T, opt_iters=1, control_std=[0.15, 0.15], temp=1., step_size=1., cov_prior_type='const_ctrl', tensor_args=ta)
obs = dict(state=torch.tensor([-0.8, -0.8], device=dev), goal_state=torch.tensor([0.8, 0.8], device=dev))
t = wall(lambda: pl.optimize(opt_iters=100, **obs), 100)
print(f'MPPI S=32 T=64 c=2 (one problem): {t*1e6:.1f} us/iter fused')
