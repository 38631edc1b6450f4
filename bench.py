#!/usr/bin/env python
"""Headline benchmark: STOMP trajectory-update iterations/sec on MI355X (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload ("C3", per GPU): panda_spheres STOMP, P=128 particles x S=32 samples = B=4096 rollouts,
H=64 support points, D=7 (Panda FK + 31 robot collision spheres vs 16 obstacle spheres), reference
example parameters (pos_only=False -> d=14, sigma_coll=1e-3, lr=0.1, T=1), on-device Philox noise.
A "step" is one pass of the planner's loop body (stomp.py:157-160) over the whole batch.  Inputs are
resident in HBM before the timed region.  N>1: every rank runs its own 128 independent start/goal
problems (weak scaling, no data-path collective) and the final means are all-gathered over RCCL inside
the timed region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def stomp_algorithmic_bytes(P, S, H, d):
    """4*[B*H*d (samples written) + 2*P*H*d (means r+w) + 2*B (costs, weights)]."""
    B = P * S
    return 4 * (B * H * d + 2 * P * H * d + 2 * B)


def cpu_baseline(wl, budget_s=12.0, max_iters=8):
    """The oracle restatement of the reference loop (kind "port") on this host's cores, on a bounded
    sample of the same workload: the FULL C3 batch (all P particles x S samples), as many iterations as
    fit the time budget (at least 2 after one warm-up)."""
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    ta = dict(device='cpu', dtype=torch.float32)
    cores = min(os.cpu_count() or 1, 32)     # more intra-op threads than this only adds contention here
    torch.set_num_threads(cores)
    prm = wl['params']
    H, S, d = prm['n_support_points'], prm['num_samples'], wl['means0'].shape[-1]
    P = wl['means0'].shape[0]
    robot, field = make_ref_geometry(wl['robot'], wl['field'], ta)
    R, Sigma, L = O.stomp_constants(H, prm['dt'], prm['sigma_spectral'], ta)
    means = wl['means0'].cpu().clone()
    cost_fn = lambda x: O.collision_cost(x, robot, field, wl['sigma_coll'])
    times = []
    t_start = time.perf_counter()
    for it in range(max_iters + 1):
        t0 = time.perf_counter()
        eps = torch.empty(S, d, P, H).normal_()
        out = O.stomp_iteration(means, eps, L, Sigma, cost_fn, prm['step_size'], prm['temperature'])
        means = out['means']
        if it > 0:
            times.append(time.perf_counter() - t0)
        if len(times) >= 2 and time.perf_counter() - t_start > budget_s:
            break
    times.sort()
    return times[len(times) // 2], cores, len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--particles', type=int, default=128)
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--pos-only', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # one rank per GPU.  MPB_DIST_BACKEND=gloo (test aid) lets several ranks share one GPU to exercise this
    # path on a single-GPU box; the default is RCCL ("nccl") over xGMI
    backend = os.environ.get('MPB_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite

    P, S = args.particles, args.samples
    wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=args.pos_only, first_particle=rank * P)
    prm = wl['params']
    H, d, D = prm['n_support_points'], wl['means0'].shape[-1], 7
    ta = dict(device=dev, dtype=torch.float32)
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'],
                                                        sigma_coll=wl['sigma_coll'], tensor_args=ta)], tensor_args=ta)
    planner = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
                    initial_particle_means=wl['means0'], tensor_args=ta, noise='philox', seed=0,
                    particle_offset=rank * P, **prm)
    gathered = [torch.empty_like(planner._particle_means) for _ in range(world)] if world > 1 else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-heat (untimed set-up, not part of W or K): code objects loaded, clocks ramped, allocator settled
    # before the contract's warm-up and timed steps; the particle means are restored afterwards
    means_init = planner._particle_means.clone()
    planner.optimize(opt_iters=500)
    torch.cuda.synchronize()
    planner._particle_means.copy_(means_init)
    planner.optimize(opt_iters=args.warmup)          # W untimed steps
    if dist is not None:                             # RCCL communicator / xGMI set-up is part of the warm-up
        dist.all_gather(gathered, planner._particle_means)
    barrier()
    t0 = time.perf_counter()
    planner.optimize(opt_iters=args.steps)           # EXACTLY K steps: one C-ABI call, 2K launches
    if dist is not None:
        dist.all_gather(gathered, planner._particle_means)   # final gather of the (P,H,d) means over xGMI
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(planner._particle_means).all()

    # ---- roofline of the dominant kernel (sample+cost), measured live with events on the launch stream
    geom = cost.cost_l[0].device_geometry(dev)
    n_prof = min(args.steps, 50)

    def launch_a(i):
        ops.stomp_sample(planner._particle_means, None, planner.state_particles, planner.scale_tril, S, seed=0,
                         it=10_000 + i, particle_offset=rank * P, geom=geom, costs=planner.costs,
                         k_sigma=cost.cost_l[0].k_sigma, weight=1.0)

    def launch_b():
        ops.stomp_update(planner._particle_means, planner.state_particles, planner.costs, planner._weights_buf,
                         planner.Sigma, planner.lr, planner.temperature)

    # n back-to-back launches of ONE kernel between two events: the event / launch overhead (~4 us when a
    # single launch is bracketed) is amortised, what remains per launch is the kernel plus its ~1 us gap
    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(n_prof):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n_prof

    ka_ms = timed(launch_a)
    kb_ms = timed(lambda i: launch_b())
    # the same two kernels inside the loop they run in (A, B, A, B, ...), each launch with its own pair of HIP events on
    # the dispatch (hipExtLaunchKernelGGL): execution begin -> end without the ~2 us dispatch gap that the back-to-back
    # figure above and rocprofv3's kernel-trace durations (profiles/) both include part of.  Reported next to
    # `kernel_ms`, which stays the conservative back-to-back figure.  The particle means are restored afterwards.
    means_keep = planner._particle_means.clone()
    ka_ev_ms, kb_ev_ms = ops.stomp_step_profile(planner._particle_means, planner.state_particles, planner.costs, planner._weights_buf,
                                          planner.scale_tril, planner.Sigma, geom, S, D, cost.cost_l[0].k_sigma, 1.0,
                                          planner.lr, planner.temperature, n_iters=n_prof, seed=0, iter0=20_000,
                                          particle_offset=rank * P)
    planner._particle_means.copy_(means_keep)
    alg_bytes_a = 4 * (P * S * H * d + P * H * d + P * S)     # kernel A: samples written + means read + costs
    achieved = alg_bytes_a / (ka_ms * 1e-3) / 1e9

    traffic = None
    tfile = os.path.join(ROOT, 'profiles', 'r01_traffic_kernelA.json')
    if os.path.exists(tfile) and P == 128 and S == 32 and not args.pos_only:
        with open(tfile) as fh:
            traffic = json.load(fh).get('hbm_bytes_per_launch')   # rocprofv3 FETCH_SIZE/WRITE_SIZE passes, see the file
    if rank == 0:
        its = world * args.steps / elapsed
        line = {
            'metric': 'stomp_trajectory_update_iters_per_sec',
            'value': its, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'panda_spheres STOMP B=%d (P=%d particles x S=%d samples) H=%d D=%d d=%d per GPU'
                                   % (P * S, P, S, H, D, d),
                       'pos_only': bool(args.pos_only), 'noise': 'device philox', 'sigma_coll': wl['sigma_coll'],
                       'robot_collision_spheres': 31, 'collision_spheres_after_static_pruning': int(geom.host.view('int32')[5]),
                       'obstacle_spheres': 16, 'parallelism': 'particles sharded x%d' % world,
                       'algorithmic_bytes_per_iter': stomp_algorithmic_bytes(P, S, H, d)},
            'roofline': {'bound': 'hbm', 'kernel': 'stomp_sample_cost_h64_kernel<14,true>', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel_ms': ka_ms, 'update_kernel_ms': kb_ms, 'kernel_ms_dispatch_events': ka_ev_ms,
                         'update_kernel_ms_dispatch_events': kb_ev_ms,
                         'algorithmic_bytes_per_launch': alg_bytes_a},
        }
        if not args.no_cpu_baseline and world == 1:
            med, cores, n_it = cpu_baseline(wl)
            line['cpu_baseline'] = {
                'value': 1.0 / med, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
                'sample': 'oracle/planners_ref.py stomp_iteration (PyTorch-CPU restatement of stomp.py:157-160 + build-defined '
                          'FK/SDF) on the full workload (P=%d x S=%d), median of %d iterations = %.3f s, %d intra-op threads'
                          % (P, S, n_it, med, cores)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
