"""Child process of tests/test_gpu_two_ranks.py: one rank of a 2-rank job whose ranks SHARE the one GPU of the box
(process group on gloo, tensors on cuda:0).  Runs the sharded STOMP and GPMP2 paths exactly as an N-GPU job would
(contiguous particle shard, Philox keyed by the global particle id, GPMP2's per-iteration all-reduce of the damping
vector, final gather of the means) and lets rank 0 save the gathered results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(dev):
    """The shared problem set (built identically by the parent for the unsharded run)."""
    from motion_planning_baselines_amd import geometry as G, workloads
    P, S, H, D = 16, 8, 64, 7
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=False)
    Bg, Hg = 6, 32
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    q = workloads.collision_free_configs(robot, field, 2 * Bg, 3, dev)
    dtg = 5.0 / Hg
    x0 = workloads.straight_line_means(q[:Bg], q[Bg:], Hg, dtg, False, dev)
    gen = torch.Generator().manual_seed(1)
    x0[:, 1:-1, :D] += (0.05 * torch.randn(Bg, Hg - 2, D, generator=gen)).to(dev)
    return dict(P=P, S=S, H=H, D=D, wl=wl, Bg=Bg, Hg=Hg, dtg=dtg, robot=robot, field=field, q=q, x0=x0.contiguous())


def run_stomp(pr, dev, lo, hi, iters=3):
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    wl = pr['wl']
    ta = dict(device=dev, dtype=torch.float32)
    cost = CostComposite(wl['robot'], pr['H'], [CostCollision(wl['robot'], pr['H'], field=wl['field'],
                                                              sigma_coll=wl['sigma_coll'], tensor_args=ta)], tensor_args=ta)
    prm = dict(wl['params'])
    prm['num_particles_per_goal'] = hi - lo
    pl = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
               initial_particle_means=wl['means0'][lo:hi].clone(), tensor_args=ta, noise='philox', seed=7,
               particle_offset=lo, **prm)
    pl.optimize(opt_iters=iters)
    return pl._particle_means


def run_gpmp2(pr, dev, lo, hi, group, iters=2):
    from motion_planning_baselines_amd.planners.gpmp2 import GPMP2
    D, q, Bg = pr['D'], pr['q'], pr['Bg']
    pl = GPMP2(robot=pr['robot'], n_dof=D, n_support_points=pr['Hg'], num_particles_per_goal=hi - lo, opt_iters=1,
               dt=pr['dtg'], start_state=torch.from_numpy(q[0]).to(dev), multi_goal_states=torch.from_numpy(q[Bg:Bg + 1]).to(dev),
               initial_particle_means=pr['x0'][lo:hi].clone(), sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0,
               solver_params=dict(delta=1e-2, trust_region=True, method='cholesky'), collision_fields=[pr['field']],
               tensor_args=dict(device=dev, dtype=torch.float32), process_group=group)
    pl.set_problem_states(torch.from_numpy(q[lo:hi]).to(dev), torch.from_numpy(q[Bg + lo:Bg + hi]).to(dev))
    pl.optimize(opt_iters=iters)
    return pl._particle_means


def main():
    out = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    # MPB_DIST_BACKEND=nccl (+ WORLD_SIZE=1, MPB_FORCE_DIST=1): the RCCL rehearsal of tests/test_gpu_rccl_world1.py -- the
    # same code on the backend an N-GPU job uses, every collective executed even though there is one rank
    backend = os.environ.get('MPB_DIST_BACKEND', 'gloo')
    force = os.environ.get('MPB_FORCE_DIST') == '1'
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from motion_planning_baselines_amd import parallel
    pr = problem(dev)
    lo, hi = parallel.shard_range(pr['P'], rank, world)
    m = parallel.gather_means(run_stomp(pr, dev, lo, hi), pr['P'], force=force)
    glo, ghi = parallel.shard_range(pr['Bg'], rank, world)
    x = parallel.gather_means(run_gpmp2(pr, dev, glo, ghi, dist.group.WORLD), pr['Bg'], force=force)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, stomp=m.cpu().numpy(), gpmp2=x.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
