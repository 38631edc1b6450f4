#!/usr/bin/env python
"""Headline benchmark: STOMP trajectory-update iterations/sec on MI355X (BASELINE.json configs[2], "C3").

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Main line (every N): per GPU, panda_spheres STOMP, P=128 particles x S=32 samples = B=4096 rollouts, H=64 support
points, D=7 (Panda FK + 31 robot collision spheres vs 16 obstacle spheres), the reference example's parameters
(pos_only=False -> d=14, sigma_coll=1e-3, lr=0.1, T=1), on-device Philox noise.  A "step" is one pass of the planner's
loop body (stomp.py:157-160) over the whole batch -- all K steps run inside ONE persistent launch (mpb_stomp_run,
csrc/mpb_stomp_fused.hip); inputs are resident in HBM before the timed region.  The K steps
are timed R times (--repeats, default 5), each block bracketed by barrier + synchronize and reduced with MAX over the
ranks; `ms_per_step` / `value` are the MEDIAN block, min / max are reported next to it.  N > 1: every rank runs its own
128 start/goal problems (weak scaling, no data-path collective), the final means are all-gathered over RCCL inside
the timed region.

Further entries of the same JSON line (the other BASELINE configs):
  c5        BASELINE configs[4]'s per-GPU load: 4096 particles (131072 rollouts) per rank, same protocol, at every N
            -- at N = 8 that is the 32768-problem job; compare c5.value across N for its weak scaling;
  c2        pointmass_dense_2d CHOMP B=1024 (N = 1 only);
  c4        panda_spheres GPMP2 B=2048, H=128, D=7 (N = 1 only; structured-FLOP roofline against the fp64 matrix peak,
            CPU baseline on a small batch and extrapolated, as SURVEY 8d prescribes).
One JSON line is printed by rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp32 VALU issue peak: 1024 SIMDs x 2.4 GHz / 2 cycles per wave-instruction (MI355X_MICROARCH.md: v_fma_f32 wave64 =
# 2 cycles on a SIMD-32) = 1228.8 G wave-instructions/s (x 64 lanes x 2 flop = the 157.3 TFLOP/s fp32 vector peak)
VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X spec, v_mfma_f64_16x16x4_f64 at 64 cycles per SIMD


def stomp_algorithmic_bytes(P, S, H, d):
    """4*[B*H*d (samples written) + 2*P*H*d (means r+w) + 2*B (costs, weights)]."""
    B = P * S
    return 4 * (B * H * d + 2 * P * H * d + 2 * B)


def latest_profile(pattern):
    """Newest committed profiles/<pattern> (rocprofv3 summaries are named per round), parsed, or None."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return json.load(fh), os.path.relpath(files[-1], ROOT)


def pmc_freshness(pmc):
    """Are the counters of a committed summary those of the kernel this process runs?  Since round 6 a summary carries the SHA-1
    of the kernel's sources as they were when the passes ran (scripts/pmc_summary.py: `pmc.sources_sha1`); recomputed here from
    the tree.  None: a summary from before round 6 (no fingerprint)."""
    meta = (pmc or {}).get('pmc')
    if not meta:
        return {'pmc_commit': None, 'pmc_matches_sources': None}
    from motion_planning_baselines_amd import build as _build
    try:
        same = _build.sources_sha1(meta['sources']) == meta['sources_sha1']
    except OSError:
        same = False
    return {'pmc_commit': meta.get('commit'), 'pmc_round': meta.get('round'), 'pmc_matches_sources': bool(same)}


def cpu_threads():
    cores = min(os.cpu_count() or 1, 32)     # more intra-op threads than this only adds contention here
    torch.set_num_threads(cores)
    return cores


def cpu_baseline_stomp(wl, L, Sigma, eps_parity, budget_s=10.0, max_iters=8):
    """The oracle restatement of the reference loop (kind "port") on this host's cores, on a bounded sample of the
    same workload: the FULL C3 batch, as many iterations as fit the time budget (at least 2 after one warm-up).  Its first
    len(eps_parity) iterations run on the injected noise the GPU side was fed (`stomp_parity_gpu`); their results are
    returned for the `parity` object of the line."""
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    ta = dict(device='cpu', dtype=torch.float32)
    cores = cpu_threads()
    prm = wl['params']
    H, S, d = prm['n_support_points'], prm['num_samples'], wl['means0'].shape[-1]
    P = wl['means0'].shape[0]
    robot, field = make_ref_geometry(wl['robot'], wl['field'], ta)
    means = wl['means0'].cpu().clone()
    cost_fn = lambda x: O.collision_cost(x, robot, field, wl['sigma_coll'])
    times, kept = [], []
    t_start = time.perf_counter()
    for it in range(max_iters + 1):
        t0 = time.perf_counter()
        eps = eps_parity[it] if it < len(eps_parity) else torch.empty(S, d, P, H).normal_()
        out = O.stomp_iteration(means, eps, L, Sigma, cost_fn, prm['step_size'], prm['temperature'])
        means = out['means']
        if it < len(eps_parity):
            kept.append(out)
        if it > 0:
            times.append(time.perf_counter() - t0)
        if len(times) >= 2 and time.perf_counter() - t_start > budget_s:
            break
    times.sort()
    return times[len(times) // 2], cores, len(times), kept


def stomp_parity_gpu(wl, planner, cost, geom, eps_parity):
    """GPU side of the line's `parity` object: the first len(eps_parity) iterations of the headline workload on INJECTED
    noise (the reference's draw order) through mpb_stomp_run_checked -- the same persistent launch the timed region runs,
    from the same initial means -- returned on the host.  Untimed."""
    from motion_planning_baselines_amd import ops
    dev = wl['means0'].device
    P, H, d = wl['means0'].shape
    S = wl['params']['num_samples']
    ws = ops.stomp_workspace(P, S, H, d, dev)
    path = ops.stomp_run_path(geom, ws, P, S, H, d)
    means = wl['means0'].clone()
    samples, costs, weights = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
    cc = cost.cost_l[0]
    ops.stomp_run(means, eps_parity.to(dev).contiguous(), samples, costs, weights, planner.scale_tril, planner.Sigma, geom, S, 7,
                  cc.k_sigma, 1.0, planner.lr, planner.temperature, ws, n_iters=len(eps_parity))
    torch.cuda.synchronize()
    assert not ops.stomp_run_timed_out(ws)
    return {'means': means.cpu(), 'costs': costs.cpu(), 'path': path}


def stomp_parity(wl, L, Sigma, eps_parity, gpu, kept):
    """`parity`: HIP path vs the oracle after len(eps_parity) free-running iterations on the same injected noise, full
    headline workload.  rel_err_means: max|a-b| / max|b| over the final means; rel_err_means_per_waypoint: worst
    ||a_ph - b_ph||_2 / ||b_ph||_2 over waypoints, position and velocity channels separately (1 % floor on the norm);
    rel_err_costs: the last iteration's costs.  At sigma_coll = 1e-3 the reference's fp32 update is itself only defined up
    to its fp32-vs-fp64 envelope (near-ties of a one-hot softmax), which is computed here by running the oracle in fp64 on
    the same noise and reported next to the errors."""
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    f64 = dict(device='cpu', dtype=torch.float64)
    robot, field = make_ref_geometry(wl['robot'], wl['field'], f64)
    prm = wl['params']
    m64 = wl['means0'].cpu().double()
    for e in eps_parity:
        m64 = O.stomp_iteration(m64, e.double(), L.double(), Sigma.double(), lambda x: O.collision_cost(x, robot, field, wl['sigma_coll']),
                                prm['step_size'], prm['temperature'])['means']
    ref = kept[-1]

    def gmax(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))

    def per_waypoint(a, b, n_pos=7):
        worst = 0.0
        for sl in (slice(0, n_pos), slice(n_pos, a.shape[-1])):
            nb = b[..., sl].double().norm(dim=-1)
            den = nb.clamp_min(1e-2 * float(nb.max()))
            worst = max(worst, float(((a[..., sl].double() - b[..., sl].double()).norm(dim=-1) / den).max()))
        return worst
    return {'iters': len(eps_parity), 'rel_err_means': gmax(gpu['means'], ref['means']),
            'rel_err_means_per_waypoint': per_waypoint(gpu['means'], ref['means']),
            'rel_err_costs': gmax(gpu['costs'], ref['costs']),
            'reference_fp32_vs_fp64_envelope_means': gmax(ref['means'], m64), 'bar': 1e-4,
            'against': 'oracle/planners_ref.py stomp_iteration (fp32, CPU) on the same injected noise, free running from the initial means, '
                       'the whole headline batch; GPU side: mpb_stomp_run_checked, persistent launch (path %d)' % gpu['path']}


def stomp_parity_philox_gpu(wl, planner, cost, geom, n_it=2):
    """GPU side of `parity.philox`: the code path the timed region runs -- device-drawn noise (eps = NULL: Philox4x32-7 +
    Box-Muller, two-component bf16 split, 30-instruction matrix product) -- for the first n_it iterations of the headline
    workload with the planner's own (seed, iter0 = 0, particle_offset), through mpb_stomp_run_checked; the normals the
    kernels drew are fetched through the test aid mpb_debug_stomp_normals_h and (a) injected through the eps argument (the
    three-component instantiation every golden test runs): the outputs must be the same BITS; (b) handed to the oracle
    (stomp_parity_philox).  Untimed."""
    from motion_planning_baselines_amd import ops
    dev = wl['means0'].device
    P, H, d = wl['means0'].shape
    S = wl['params']['num_samples']
    ws = ops.stomp_workspace(P, S, H, d, dev)
    cc = cost.cost_l[0]
    out = [torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)]

    def run(n, eps):
        means = wl['means0'].clone()
        ops.stomp_run(means, eps, *out, planner.scale_tril, planner.Sigma, geom, S, 7, cc.k_sigma, 1.0, planner.lr, planner.temperature,
                      ws, n_iters=n, seed=planner.seed, iter0=0, particle_offset=planner.particle_offset)
        torch.cuda.synchronize()
        assert not ops.stomp_run_timed_out(ws)
        return [means.clone()] + [t.clone() for t in out]
    nrm = ops.debug_stomp_normals(P, S, d, n_it, dev, seed=planner.seed, iter0=0, particle_offset=planner.particle_offset, H=H)
    eps = nrm[..., :H].permute(0, 2, 3, 1, 4).contiguous()            # the reference's draw order (n_it, S, d, P, H)
    first = run(1, None)                                              # samples of iteration 0: means0 + L @ normals[0]
    noise = torch.einsum('hk,sdpk->pshd', planner.scale_tril.double(), eps[0].double())
    noise[:, :, 0, :] = 0
    noise[:, :, -1, :] = 0                                            # stomp.py:105-106
    want = wl['means0'].double().unsqueeze(1) + noise
    e_samples = float((first[1].double() - want).abs().max() / want.abs().max())
    drawn, injected = run(n_it, None), run(n_it, eps)
    same = all(torch.equal(a, b) for a, b in zip(drawn, injected))
    return {'means': drawn[0].cpu(), 'costs': drawn[2].cpu(), 'eps': eps.cpu(), 'samples_vs_L_normals': e_samples,
            'injected_same_bits': bool(same), 'path': ops.stomp_run_path(geom, ws, P, S, H, d)}


def stomp_parity_philox(wl, L, Sigma, gpu):
    """`parity.philox`: the device-noise path against the oracle on the normals the kernels drew (free running from the
    initial means, the whole headline batch), next to the reference's own fp32-vs-fp64 envelope on the same normals."""
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    prm = wl['params']
    outs = {}
    for dtype in (torch.float32, torch.float64):
        robot, field = make_ref_geometry(wl['robot'], wl['field'], dict(device='cpu', dtype=dtype))
        m = wl['means0'].cpu().to(dtype)
        for e in gpu['eps']:
            out = O.stomp_iteration(m, e.to(dtype), L.to(dtype), Sigma.to(dtype), lambda x: O.collision_cost(x, robot, field, wl['sigma_coll']),
                                    prm['step_size'], prm['temperature'])
            m = out['means']
        outs[dtype] = out
    ref, ref64 = outs[torch.float32], outs[torch.float64]
    gmax = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))
    return {'iters': len(gpu['eps']), 'samples_vs_means_plus_L_normals_fp64': gpu['samples_vs_L_normals'],
            'by_particle': parity_by_particle(gpu['means'], ref['means'], ref64['means']),
            'rel_err_means': gmax(gpu['means'], ref['means']), 'rel_err_means_per_waypoint': _per_waypoint(gpu['means'], ref['means']),
            'rel_err_costs': gmax(gpu['costs'], ref['costs']),
            'reference_fp32_vs_fp64_envelope_means': gmax(ref['means'], ref64['means']),
            'injected_normals_same_bits': gpu['injected_same_bits'], 'bar': 1e-4,
            'against': 'the TIMED instantiation (eps = NULL: device Philox draw, two-component bf16 split, 30-MFMA product; path %d): its '
                       'drawn normals (mpb_debug_stomp_normals_h) fed to oracle/planners_ref.py stomp_iteration, free running; '
                       '`injected_normals_same_bits`: the same normals through the eps argument (three-component instantiation) give '
                       'bit-identical means / samples / costs / weights' % gpu['path']}


def parity_by_particle(got, ref32, ref64, n_pos=7, bar=1e-4):
    """Final means (P,H,d) of a K-iteration free run judged PER PARTICLE against the oracle's fp32 run of the same noise
    (VERDICT r05 item 1): error of a particle in both norms of the parity tests -- max|a - b| over the particle / max|b| over
    the batch, and the worst waypoint ||a_ph - b_ph||_2 / max(||b_ph||_2, 1 % of the largest waypoint norm), position and
    velocity channels separately.  `frac_strict` = share of particles within `bar` in BOTH norms; `worst_rest` = the largest
    global-norm error among the others, to be read against `envelope` = the reference's OWN fp32-vs-fp64 spread on the same noise
    (at sigma_coll = 1e-3 the softmax is one-hot: a near-tie that fp32 and fp64 resolve differently sends a particle down another
    sample's path, in the reference itself); `reference_frac_strict` = the share of particles on which the reference's fp32 and
    fp64 runs agree within `bar` -- the yardstick for `frac_strict`."""
    a, b, c = got.detach().cpu().double(), ref32.detach().cpu().double(), ref64.detach().cpu().double()

    def norms(x, y):
        g = (x - y).abs().amax(dim=(1, 2)) / y.abs().max().clamp_min(1e-12)
        w = torch.zeros(x.shape[0], dtype=torch.float64)
        for sl in (slice(0, n_pos), slice(n_pos, x.shape[-1])):
            if sl.start >= x.shape[-1]:
                continue
            nb = y[..., sl].norm(dim=-1)
            den = nb.clamp_min(1e-2 * float(nb.max().clamp_min(1e-12)))
            w = torch.maximum(w, ((x[..., sl] - y[..., sl]).norm(dim=-1) / den).amax(dim=1))
        return g, w
    g, w = norms(a, b)
    strict = (g < bar) & (w < bar)
    rg, rw = norms(b, c)
    ref_strict = (rg < bar) & (rw < bar)
    return {'particles': int(a.shape[0]), 'bar': bar, 'frac_strict': float(strict.double().mean()),
            'strict_worst_global': float(g[strict].max()) if bool(strict.any()) else None,
            'strict_worst_per_waypoint': float(w[strict].max()) if bool(strict.any()) else None,
            'worst_rest': float(g[~strict].max()) if bool((~strict).any()) else 0.0,
            'envelope': float(rg.max()), 'reference_frac_strict': float(ref_strict.double().mean()),
            'strict_in_both': float((strict & ref_strict).double().mean())}


def _per_waypoint(a, b, n_pos=7):
    worst = 0.0
    for sl in (slice(0, n_pos), slice(n_pos, a.shape[-1])):
        if sl.start >= a.shape[-1]:
            continue
        nb = b[..., sl].double().norm(dim=-1)
        den = nb.clamp_min(1e-2 * float(nb.max()))
        worst = max(worst, float(((a[..., sl].double() - b[..., sl].double()).norm(dim=-1) / den).max()))
    return worst


def cpu_baseline_sample(wl, L, Sigma, n_particles, budget_s=6.0):
    """The oracle's STOMP iteration on the FIRST n_particles particles of a workload (x its S samples), median seconds per
    iteration over as many iterations as fit the budget (at least 2 after one warm-up)."""
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    ta = dict(device='cpu', dtype=torch.float32)
    cores = cpu_threads()
    prm = wl['params']
    H, S, d = prm['n_support_points'], prm['num_samples'], wl['means0'].shape[-1]
    robot, field = make_ref_geometry(wl['robot'], wl['field'], ta)
    means = wl['means0'][:n_particles].cpu().clone()
    ts, t_start = [], time.perf_counter()
    for it in range(9):
        t0 = time.perf_counter()
        means = O.stomp_iteration(means, torch.empty(S, d, n_particles, H).normal_(), L, Sigma,
                                  lambda x: O.collision_cost(x, robot, field, wl['sigma_coll']), prm['step_size'], prm['temperature'])['means']
        if it > 0:
            ts.append(time.perf_counter() - t0)
        if len(ts) >= 2 and time.perf_counter() - t_start > budget_s:
            break
    ts.sort()
    return ts[len(ts) // 2], cores, len(ts)


class Clock:
    """R timed blocks of one callable, each bracketed by barrier + synchronize, MAX over the ranks.

    The closing bracket first spins on a stream query (host polling, no sleep) and THEN calls the barrier and
    torch.cuda.synchronize(), which by then return at once: the timed region still ends when every kernel of the block has
    finished and synchronize() has returned, but the runtime's blocking wait is taken out of it -- on this pool
    hipDeviceSynchronize wakes up 15-25 us late for about half the launches of one particular duration (~360 us, i.e.
    exactly 20 C3 steps; scripts/sync_latency.py: K = 19 and K = 21 are not affected), which is host scheduling, not
    planner time."""

    def __init__(self, dist, dev):
        self.dist, self.dev = dist, dev
        self._flag = torch.zeros(1, device=dev) if dist is not None else None

    def barrier(self, spin=False):
        """Every rank has finished everything it enqueued.  Across ranks the barrier is an all_reduce of one element on
        the process group's backend (on "nccl" that is what ProcessGroupNCCL.barrier itself runs before it blocks the host
        in a device synchronize); here the host spins on an event behind it instead of sleeping, then calls
        torch.cuda.synchronize(), which returns at once."""
        if self.dist is not None:
            self.dist.all_reduce(self._flag)
        if spin or self.dist is not None:
            self._spin()
        torch.cuda.synchronize()

    @staticmethod
    def _spin():
        """Host polling until the current stream has drained (hipStreamQuery: no marker packet behind the kernel -- an event
        recorded for the purpose is one more packet the queue processes after the kernel's end, ~2-3 us in the timed region)."""
        s = torch.cuda.current_stream()
        while not s.query():
            pass

    def wait(self):
        """This rank has finished everything it enqueued (host spin on the stream, then synchronize)."""
        self._spin()
        torch.cuda.synchronize()

    def blocks(self, fn, repeats, before=None):
        """`fn` returns True when the LAST thing it enqueued is a collective that completes on no rank before every rank has
        enqueued it behind its own steps (the final all-gather of the means): that collective IS the closing barrier of the
        block -- no rank leaves it before all ranks have finished their K steps -- and the bracket is closed by waiting for
        it; a second collective behind it would only add its own latency to the timed region.  Otherwise (single process,
        or no such collective) the explicit barrier closes the block."""
        out = []
        for _ in range(repeats):
            if before is not None:
                before()                 # untimed: every block starts from the same state
            self.barrier()
            t0 = time.perf_counter()
            closed = fn()
            if closed and self.dist is not None:
                self.wait()
            else:
                self.barrier(spin=True)
            el = time.perf_counter() - t0
            if self.dist is not None:
                t = torch.tensor([el], device=self.dev, dtype=torch.float64)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                el = float(t.item())
            out.append(el)
        return out


def spread(blocks, steps):
    b = sorted(blocks)
    med = b[len(b) // 2]
    return med, {'n': len(b), 'ms_per_step_median': 1e3 * med / steps, 'ms_per_step_min': 1e3 * b[0] / steps,
                 'ms_per_step_max': 1e3 * b[-1] / steps}


def make_stomp(P, S, dev, rank, pos_only=False, H=64):
    from motion_planning_baselines_amd import workloads
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=pos_only, first_particle=rank * P)
    prm = wl['params']
    H = prm['n_support_points']
    ta = dict(device=dev, dtype=torch.float32)
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'],
                                                        sigma_coll=wl['sigma_coll'], tensor_args=ta)], tensor_args=ta)
    planner = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
                    initial_particle_means=wl['means0'], tensor_args=ta, noise='philox', seed=0,
                    particle_offset=rank * P, check='deferred', **prm)   # (the timed region synchronises itself; loss checked after it)
    return wl, cost, planner


def STOMP_two_kernel(wl, cost, dev, rank, P):
    """The same planner on the two-kernels-per-iteration path (persistent=False), for the comparison entry."""
    from motion_planning_baselines_amd.planners.stomp import STOMP
    ta = dict(device=dev, dtype=torch.float32)
    return STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
                 initial_particle_means=wl['means0'], tensor_args=ta, noise='philox', seed=0,
                 particle_offset=rank * P, persistent=False, check='deferred', **wl['params'])


def bench_seeded(wl, cost, dev, rank, P, steps):
    """`seeded`: the IDENTICAL-SEED modes of the headline workload.  north_star judges parity "on identical
    seeds": STOMP(noise='torch_cpu') draws every iteration's (S,d,P,H) standard normals with the torch CPU generator exactly as
    the reference does (stomp.py:97-108 through MultivariateNormal), so torch.manual_seed(s) reproduces a reference run; 'torch'
    is the same call sequence on the device generator.  The main line times device Philox noise; this entry is what the seeded
    modes cost and where the time goes."""
    from motion_planning_baselines_amd.planners.stomp import STOMP
    ta = dict(device=dev, dtype=torch.float32)
    out = {'workload': "the main line's (C3), K = %d iterations per optimize() call, check='sync'" % steps}
    for mode in ('torch_cpu', 'torch'):
        pl = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost, initial_particle_means=wl['means0'],
                   tensor_args=ta, noise=mode, seed=0, particle_offset=rank * P, check='sync', **wl['params'])
        torch.manual_seed(0)
        pl.optimize(opt_iters=2)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            pl._particle_means.copy_(wl['means0'])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl.optimize(opt_iters=steps)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
        t = sorted(ts)[1]
        ent = {'value': 1.0 / t, 'unit': 'iters/s', 'ms_per_step': 1e3 * t, 'blocks': 3}
        if mode == 'torch_cpu':
            ring = pl._eps_ring
            hs = []
            for _ in range(3):
                t0 = time.perf_counter()
                ring['host'][0].normal_()
                hs.append(time.perf_counter() - t0)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize()
            e0.record()
            ring['dev'][0].copy_(ring['host'][0], non_blocking=True)
            e1.record()
            from motion_planning_baselines_amd import ops
            geom = cost.cost_l[0].device_geometry(dev)
            cc = cost.cost_l[0]
            ops.stomp_run(pl._particle_means, ring['dev'][0], pl.state_particles, pl.costs, pl._weights_buf, pl.scale_tril, pl.Sigma, geom,
                          pl.num_samples, pl.n_dof, cc.k_sigma, 1.0, pl.lr, pl.temperature, pl._run_ws, n_iters=1, seed=0, iter0=0,
                          status=pl._status)
            e2.record()
            torch.cuda.synchronize()
            ent.update({'host_draw_ms': 1e3 * sorted(hs)[1], 'h2d_ms': e0.elapsed_time(e1), 'kernel_ms_one_iteration_launch': e1.elapsed_time(e2),
                        'bytes_per_iteration_over_pcie': int(ring['host'][0].numel() * 4), 'host_generator_threads': 1,
                        'note': 'two-stage pipeline: the host draws block k + 1 (torch CPU generator: one Mersenne-Twister stream, serial '
                                'by construction -- the same cost the reference\'s own CPU loop pays per iteration) into pinned memory while the '
                                'GPU copies and consumes block k (async H2D + one n_iters = 1 persistent launch, injected-noise instantiation): '
                                'ms_per_step ~ host_draw_ms.  PCIe-inclusive by nature; never the main line\'s `value`'})
        else:
            ent['note'] = ('per iteration one device normal_() kernel into a chunk buffer (the reference\'s call sequence on the device '
                           'generator), one persistent launch per chunk of <= 16 iterations (injected-noise instantiation)')
        out[mode] = ent
        del pl
    return out


def bench_generic_model(wl, dev, rank, P, S, steps):
    """`generic_model`: the main line's workload with the robot taken from the geometry buffer's TABLES (pack_geometry(use_model=False):
    the table-driven chain walk any serial chain gets) instead of the compile-time Panda model whose DH entries and collision spheres
    are literals of the instruction stream (csrc/mpb_model_panda.h, geometry flag bits 0-7) -- what a second robot costs on the same
    persistent launch.  Same bits as the model walk (tests/test_gpu_stomp_fused.py)."""
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    ta = dict(device=dev, dtype=torch.float32)
    H = wl['params']['n_support_points']
    cc = CostCollision(wl['robot'], H, field=wl['field'], sigma_coll=wl['sigma_coll'], tensor_args=ta)
    cc._geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev, use_model=False)
    assert (cc._geom.flags & 0xFF) == 0
    cost = CostComposite(wl['robot'], H, [cc], tensor_args=ta)
    pl = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost, initial_particle_means=wl['means0'],
               tensor_args=ta, noise='philox', seed=0, particle_offset=rank * P, check='deferred', **wl['params'])
    for _ in range(10):
        pl._particle_means.copy_(wl['means0'])
        pl.optimize(opt_iters=steps)
        torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        pl._particle_means.copy_(wl['means0'])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.optimize(opt_iters=steps)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    assert not pl.persistent_timed_out()
    t = sorted(ts)[len(ts) // 2]
    return {'value': steps / t, 'unit': 'iters/s', 'ms_per_step': 1e3 * t / steps, 'blocks': len(ts), 'path': int(pl.run_path()),
            'note': 'table-driven robot (geometry flags model id 0) on the persistent launch, K = %d steps per block from the initial means, '
                    'median of %d blocks bracketed by synchronize; the main line is the compile-time Panda model' % (steps, len(ts))}


def bench_crowded(dev, rank, P, S, steps):
    """`crowded`: the main line's shape (P x S rollouts, H = 64, d = 14, Panda) in a scene of 200 obstacle spheres + 32 boxes -- beyond
    the compact broad-phase grid (63 spheres, three candidates per cell), which until round 5 sent such a scene to the two-kernel
    EXHAUSTIVE walk.  Round 6: geometry version 7 (list grid: any number of candidates per cell, boxes culled like spheres) on the
    persistent launch (stomp_fused_hx_kernel<..., LIST>); the two-kernel exhaustive path next to it.  Same bits (tests)."""
    from motion_planning_baselines_amd import workloads
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    wl = workloads.panda_crowded_stomp(P, dev, S=S)
    H = wl['params']['n_support_points']
    ta = dict(device=dev, dtype=torch.float32)
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'], sigma_coll=wl['sigma_coll'], tensor_args=ta)],
                         tensor_args=ta)
    out = {'workload': 'panda STOMP P=%d x S=%d H=%d d=14, 200 obstacle spheres + 32 boxes (geometry version 7: list grid)' % (P, S, H)}
    for name, persistent in (('persistent', True), ('two_kernel_exhaustive', False)):
        pl = STOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost, initial_particle_means=wl['means0'],
                   tensor_args=ta, noise='philox', seed=0, particle_offset=rank * P, check='deferred', persistent=persistent, **wl['params'])
        for _ in range(4):
            pl._particle_means.copy_(wl['means0'])
            pl.optimize(opt_iters=steps)
            torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            pl._particle_means.copy_(wl['means0'])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl.optimize(opt_iters=steps)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        out[name] = {'value': steps / t, 'unit': 'iters/s', 'ms_per_step': 1e3 * t / steps, 'path': int(pl.run_path())}
        assert not pl.persistent_timed_out()
    geom = cost.cost_l[0].device_geometry(dev)
    gi = geom.host.view('int32')
    out['grid'] = {'version': int(gi[1]), 'cells': [int(gi[17]), int(gi[18]), int(gi[19])], 'candidate_bytes': int(4 * (gi[13] - gi[16] - (gi[26] + 1023) // 1024 * 1024))}
    return out


def run_stomp(planner, clock, dist, world, steps, warmup, repeats, preheat, cold=False):
    """W untimed steps, then R blocks of EXACTLY `steps` steps (one C-ABI call = 2K launches) + the final gather."""
    # the final gather's destination: one flat (world * P, H, d) tensor (all_gather_into_tensor: one RCCL kernel, no
    # per-rank copy-out kernels)
    gathered = (torch.empty((world * planner._particle_means.shape[0],) + tuple(planner._particle_means.shape[1:]),
                            device=planner._particle_means.device) if dist is not None else None)
    means_init = planner._particle_means.clone()
    run_stomp.cold_block_s = None
    if cold and dist is None:
        # `cold`: what `--preheat 0` would time as its FIRST block -- W warm-up steps (the first call into the library: workspace,
        # plan, code object), then one K-step block from the initial means, bracketed like the timed blocks.  Untimed blocks of
        # the protocol (pre-heat) come only after it.
        planner.optimize(opt_iters=warmup)
        planner._particle_means.copy_(means_init)
        clock.barrier()
        t0 = time.perf_counter()
        planner.optimize(opt_iters=steps)
        clock.barrier(spin=True)
        run_stomp.cold_block_s = time.perf_counter() - t0
        planner._particle_means.copy_(means_init)
    if preheat:
        # device pre-heat (untimed set-up, not part of W or K): `preheat` untimed blocks of the very shape that is timed below
        # (K steps from the initial means, synchronize).  The chip is power-managed: right after ONE long launch (round 2's
        # pre-heat: 500 iterations = 8.5 ms of dense VALU work) the clocks sit at their throttled level and the K-step
        # blocks that follow speed up block after block (432 -> 378 us over forty 20-step blocks on one box,
        # DESIGN section 6); what is timed here is the steady state of the protocol itself -- bursts of K steps with a
        # synchronize between them.
        for _ in range(preheat):
            planner._particle_means.copy_(means_init)
            planner.optimize(opt_iters=steps)
            torch.cuda.synchronize()
        planner._particle_means.copy_(means_init)
    planner.optimize(opt_iters=warmup)
    def gather():
        if dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(gathered, planner._particle_means)
        else:                                        # (gloo rehearsals on a one-GPU box)
            dist.all_gather(list(gathered.chunk(world)), planner._particle_means)
    if dist is not None:                             # RCCL communicator / xGMI set-up is part of the warm-up
        gather()

    def block():
        planner.optimize(opt_iters=steps)
        if dist is not None:
            gather()                                 # final gather of the (P,H,d) means over xGMI
        return dist is not None                      # (the all-gather closes the block: Clock.blocks)
    spans = []       # the timed launches' own duration on the device's clock (the kernel stamps its pinned status block: free)

    def reset():
        if planner._status is not None and planner._status.device_span_ms() is not None:
            spans.append(planner._status.device_span_ms())
        planner._particle_means.copy_(means_init)
    blocks = clock.blocks(block, repeats, before=reset)
    reset()
    run_stomp.device_span_ms = sorted(spans[-repeats:]) if len(spans) >= repeats else None
    assert torch.isfinite(planner._particle_means).all()
    assert not planner.persistent_timed_out(), 'a workgroup of the persistent STOMP kernel gave up waiting for its partner'
    # the kernel's own duration: the SAME K-step launch, from the same state, R more times with a HIP event pair recorded on
    # the dispatch itself (mpb_stomp_run_timed: hipExtLaunchKernelGGL -- kernel begin / end, what rocprofv3 --kernel-trace
    # reports; events recorded around the launch on the stream add the marker packets' own ~20 us).  Kept out of the timed
    # blocks above.
    ev_ms = []
    for _ in range(repeats):
        reset()
        clock.barrier()
        ms = planner.optimize_timed(steps)
        if ms is None:       # not on the persistent path: plain stream events around the call
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            planner.optimize(opt_iters=steps)
            e1.record()
            clock.barrier(spin=True)
            ms = e0.elapsed_time(e1)
        ev_ms.append(ms)
    run_stomp.launch_ms = sorted(ev_ms)
    return blocks


def bench_c2(dev, steps, with_cpu=True):
    """BASELINE configs[1]: pointmass_dense_2d CHOMP, B=1024, H=64, D=2 (d=4): the whole loop is ONE launch."""
    from motion_planning_baselines_amd import workloads
    from motion_planning_baselines_amd.planners.chomp import CHOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    B, H = 1024, 64
    ta = dict(device=dev, dtype=torch.float32)
    wl = workloads.pointmass_dense_chomp(B, dev)
    cost = CostComposite(wl['robot'], H, [CostCollision(wl['robot'], H, field=wl['field'], sigma_coll=wl['sigma_coll'],
                                                        tensor_args=ta)], weights_cost_l=[wl['weight']], tensor_args=ta)
    pl = CHOMP(opt_iters=1, start_state=torch.from_numpy(wl['starts'][0]).to(dev), cost=cost,
               initial_particle_means=wl['means0'], tensor_args=ta, **wl['params'])
    pl.optimize(opt_iters=steps)
    torch.cuda.synchronize()
    blocks = []
    for _ in range(5):
        pl.reset(initial_particle_means=wl['means0'])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.optimize(opt_iters=steps)
        torch.cuda.synchronize()
        blocks.append(time.perf_counter() - t0)
    med, sp = spread(blocks, steps)
    d = wl['means0'].shape[-1]
    alg_call = 4 * 2 * B * H * d                                   # the state is read and written once per CALL
    out = {
        'workload': 'pointmass_dense_2d CHOMP B=%d H=%d D=2 d=%d, %d iterations per launch' % (B, H, d, steps),
        'metric': 'chomp_trajectory_update_iters_per_sec', 'value': steps / med, 'unit': 'iters/s',
        'ms_per_step': 1e3 * med / steps, 'repeats': sp, 'dtype': 'f32',
        'roofline': {'bound': 'valu/latency', 'note': 'one workgroup per particle keeps its trajectory in registers for the whole '
                     'loop: HBM sees the state once per CALL, so the per-iteration HBM figure is not a bound',
                     'algorithmic_bytes_per_call': alg_call, 'algorithmic_bytes_per_iter_survey': alg_call,
                     'hbm_frac_if_streamed_every_iter': alg_call / (med / steps) / 1e9 / HBM_PEAK_GBS}}
    pmc, pmc_file = latest_profile('r*_pmc_chomp.json')
    if pmc and pmc.get('SQ_INSTS_VALU_per_wave_iteration') and B == 1024:
        ginstr = pmc['SQ_INSTS_VALU_per_wave_iteration'] * pmc['waves_per_launch'] / (med / steps) / 1e9
        out['roofline'].update({'bound': 'valu', 'achieved': ginstr, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instr/s',
                                'frac': ginstr / VALU_PEAK_GINSTR, 'pmc_source': pmc_file, **pmc_freshness(pmc),
                                'valu_instructions_per_wave_iteration': pmc['SQ_INSTS_VALU_per_wave_iteration']})
    if with_cpu:
        # CPU: the oracle's autograd restatement of chomp.py:134-149 on the full batch
        from oracle import planners_ref as O
        from oracle.geometry_ref import make_ref_geometry
        cores = cpu_threads()
        cta = dict(device='cpu', dtype=torch.float32)
        robot, field = make_ref_geometry(wl['robot'], wl['field'], cta)
        R = O.chomp_precision(H, wl['params']['dt'], cta)
        m = wl['means0'].cpu().clone()
        ts = []
        for it in range(13):
            t0 = time.perf_counter()
            m = O.chomp_iteration(m, R, lambda x: O.collision_cost(x, robot, field, wl['sigma_coll'], wl['weight']),
                                  wl['params']['weight_prior_cost'], wl['params']['step_size'], wl['params']['grad_clip'])['means']
            if it >= 3:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        cpu_med = ts[len(ts) // 2]
        out['cpu_baseline'] = {'value': 1.0 / cpu_med, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
                               'sample': 'oracle chomp_iteration (autograd restatement of chomp.py:134-149) on the full batch, '
                                         'median of %d iterations = %.4f s' % (len(ts), cpu_med)}
    return out


def bench_c4(dev, steps, with_cpu=True):
    """BASELINE configs[3]: panda_spheres GPMP2, B=2048, H=128, D=7 (fp64 block-tridiagonal solve)."""
    from motion_planning_baselines_amd import geometry as G, workloads
    from motion_planning_baselines_amd.planners.gpmp2 import GPMP2
    B, H, D = 2048, 128, 7
    ta = dict(device=dev, dtype=torch.float32)
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
    dt = 5.0 / H
    means0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
    means0[:, 0, D:] = 0
    means0[:, -1, D:] = 0
    sig = dict(sigma_start=1e-5, sigma_gp=1e-2, sigma_goal_prior=1e-5, sigma_coll=1e-5)
    pl = GPMP2(robot=robot, n_dof=D, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt,
               start_state=torch.from_numpy(q[0]).to(dev), multi_goal_states=torch.from_numpy(q[B:B + 1]).to(dev),
               initial_particle_means=means0, solver_params=dict(delta=1e-2, trust_region=True, method='cholesky'),
               collision_fields=[field], tensor_args=ta, **sig)
    pl.set_problem_states(torch.from_numpy(q[:B]).to(dev), torch.from_numpy(q[B:]).to(dev))
    pl.optimize(opt_iters=3)
    torch.cuda.synchronize()

    def timed_blocks(k, n_blocks=5):
        """every block = the FIRST k iterations from the initial straight lines (round 6: the low-rank form's time depends on how
        many waypoints are inside the hinge margin, which falls as the trajectories leave the obstacles; until round 5 the blocks
        continued from wherever the previous block had left the particles, which the block elimination's time did not notice)"""
        out = []
        for _ in range(n_blocks):
            pl._particle_means.copy_(means0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl.optimize(opt_iters=k)
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t0)
        return out
    blocks = timed_blocks(steps)
    med, sp = spread(blocks, steps)
    assert torch.isfinite(pl._particle_means).all()
    first_ms = 1e3 * sorted(timed_blocks(1))[2]
    # the block elimination of rounds 1-5 on the same blocks (MPB_GPMP2_FORM is read per call)
    lr_form = os.environ.get('MPB_GPMP2_FORM', '') != 'block' and os.environ.get('MPB_GPMP2_SM') is None and (H - 1) <= 127
    block_ms = None
    if lr_form:
        os.environ['MPB_GPMP2_FORM'] = 'block'
        try:
            pl.optimize(opt_iters=2)
            block_ms = 1e3 * spread(timed_blocks(steps, 3), steps)[0] / steps
        finally:
            del os.environ['MPB_GPMP2_FORM']
    # structured flops: per particle (H-1) block steps x (potrf 14^3/3 + trsm 14^3 + syrk/gemm 2*14^3)
    n = 2 * D
    flop_iter = B * (H - 1) * (n ** 3 / 3 + n ** 3 + 2 * n ** 3)
    tflops = flop_iter / (med / steps) / 1e12
    out = {
        'workload': 'panda_spheres GPMP2 B=%d H=%d D=%d (N=%d unknowns per particle), trust region, fp64 solve' % (B, H, D, n * H),
        'metric': 'gpmp2_trajectory_update_iters_per_sec', 'value': steps / med, 'unit': 'iters/s',
        'ms_per_step': 1e3 * med / steps, 'repeats': sp, 'dtype': 'f64',
        'protocol': 'every block = the first %d iterations from the initial straight lines (reset before each block)' % steps,
        'ms_first_iteration': first_ms,
        'form': ('low-rank (csrc/mpb_gpmp2_lr.hip): A0 = priors + GP blocks + damping shared by all particles -- its cyclic-reduction '
                 'coefficients computed once per iteration, A0^-1 g by parallel cyclic reduction (a wave per chain) -- + one dense SPD '
                 'solve over the ACTIVE collision rows per particle, largest systems first') if lr_form
                else 'block elimination (csrc/mpb_gpmp2.hip)',
        'block_form_ms_per_step': block_ms,
        'roofline': {'bound': 'mfma', 'achieved': tflops, 'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': tflops / FP64_MATRIX_PEAK_TFLOPS, 'structured_flop_per_iter': flop_iter,
                     'note': 'SURVEY 8(d)\'s structured count of the block-tridiagonal elimination (14 x 14 blocks, 127 block steps per particle) '
                             'over the iteration\'s time, against the fp64 matrix peak -- the figure rounds 1-5 report.  The low-rank form of '
                             'round 6 does NOT execute those flops: it solves the same system with ~0.2 MFLOP per particle (the cyclic-reduction solves) plus '
                             'n_a^3 / 3 for its n_a active rows; the iteration is bound by the latency of the largest capacitance system (DESIGN 6)'}}
    pmc, pmc_file = latest_profile('r*_pmc_solve.json')
    if pmc and pmc.get('SQ_INSTS_VALU_per_wave') and B == 2048 and not lr_form:
        # what actually binds the solve kernel (88 % of the iteration): fp64 VALU issue (4.7 cycles per wave-instruction,
        # profiles/r01_microbench_valu.txt) and the W_t workspace stream; the whole iteration's time is used (conservative)
        it_s = med / steps
        ginstr = pmc['SQ_INSTS_VALU_per_wave'] * pmc['waves_per_launch'] / it_s / 1e9
        peak64 = 1024 * 2.4 / 4.7
        peak64_spec = 1024 * 2.4 / 4.0          # the spec rate: 78.6 TFLOP/s fp64 vector / (64 lanes x 2 flop) = 614.4 G wave-instr/s
        out['roofline']['valu_f64'] = {'achieved': ginstr, 'peak': peak64, 'unit': 'G wave-instr/s', 'frac': ginstr / peak64,
                                       'peak_spec': peak64_spec, 'frac_of_spec': ginstr / peak64_spec,
                                       'peak_note': '`peak` = the microbenched 4.7 cycles per fp64 wave-instruction (profiles/r01_microbench_valu.txt), '
                                                    '`peak_spec` = the 4 cycles of the data sheet',
                                       'valu_instructions_per_wave': pmc['SQ_INSTS_VALU_per_wave'], 'pmc_source': pmc_file, **pmc_freshness(pmc)}
        cls = {c: pmc.get('SQ_INSTS_VALU_%s_per_wave' % c) for c in ('ADD_F64', 'MUL_F64', 'FMA_F64', 'TRANS_F64', 'INT32', 'INT64', 'CVT')}
        if all(v is not None for v in cls.values()) and pmc.get('SQ_INSTS_MFMA_per_wave') is not None:
            # the fp64 pipe of a SIMD by instruction class (architectural cycles per wave-instruction: fp64 add / mul / fma 4,
            # v_rcp_f64 16, convert and 64-bit integer 4, v_mfma_f64_16x16x4_f64 64; matrix and vector instructions do not run
            # together: profiles/r03_microbench_overlap.txt).  low: 32-bit integer and everything else (select, compare, move,
            # lane operations) at 2 -- a strict lower bound; mid: those at 3 and 3.5.  The solve kernel's share of the iteration's
            # time is used (conservative: the whole iteration)
            other = pmc['SQ_INSTS_VALU_per_wave'] - sum(cls.values())
            fixed = (4.0 * (cls['ADD_F64'] + cls['MUL_F64'] + cls['FMA_F64']) + 16.0 * cls['TRANS_F64'] + 4.0 * (cls['INT64'] + cls['CVT'])
                     + 64.0 * pmc['SQ_INSTS_MFMA_per_wave'])
            low, mid = fixed + 2.0 * (cls['INT32'] + other), fixed + 3.0 * cls['INT32'] + 3.5 * other
            per_s = pmc['waves_per_launch'] / it_s / 1e9 / (1024 * 2.4)
            out['roofline']['fp64_pipe_by_class'] = {'frac_low': low * per_s, 'frac_mid': mid * per_s, 'cycles_per_wave_low': low,
                                                     'cycles_per_wave_mid': mid, 'instructions_per_wave': dict(cls, OTHER=other),
                                                     'mfma_valu_coexec_cycles_per_launch': pmc.get('SQ_VALU_MFMA_COEXEC_CYCLES_per_wave')}
        if pmc.get('FETCH_SIZE_KB_raw_per_launch') and pmc.get('WRITE_SIZE_KB_raw_per_launch'):
            traffic = (2 * pmc['FETCH_SIZE_KB_raw_per_launch'] + pmc['WRITE_SIZE_KB_raw_per_launch']) * 1024
            out['roofline']['hbm'] = {'achieved': traffic / it_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                      'frac': traffic / it_s / 1e9 / HBM_PEAK_GBS, 'traffic': traffic,
                                      'note': 'counter traffic of the solve kernel (FETCH_SIZE doubled per the guide + WRITE_SIZE): the W_t records written by the elimination and read back by the substitution'}
    if with_cpu:
        # CPU: the oracle's DENSE restatement (gpmp2.py:308-368, :451-452) on a small batch; the dense system of the
        # full batch needs > 150 GB, so the figure is extrapolated linearly in B and says so
        from oracle import planners_ref as O
        from oracle.geometry_ref import make_ref_geometry
        cores = cpu_threads()
        f64 = dict(device='cpu', dtype=torch.float64)
        rrobot, rfield = make_ref_geometry(robot, field, f64)
        Bc = 32          # BASELINE.md section 3 / SURVEY 8(d): timed at B = 32, scaled x64 (the dense system: ~3.5 GB at this size)
        x0 = means0[:Bc].cpu().double()
        start = torch.cat([torch.from_numpy(q[0]).double(), torch.zeros(D, dtype=torch.float64)])
        goal = torch.cat([torch.from_numpy(q[B]).double(), torch.zeros(D, dtype=torch.float64)])
        ts = []
        for it in range(2):
            t0 = time.perf_counter()
            O.gpmp2_iteration(x0, rrobot, rfield, start, goal, D=D, dt=dt, sigma_start=1e-5, sigma_gp=1e-2, sigma_goal=1e-5,
                              sigma_coll=1e-5, delta=1e-2, trust_region=True, step_size=1.0, tensor_args=f64)
            if it >= 1:
                ts.append(time.perf_counter() - t0)
        cpu_t = min(ts) * (B / Bc)
        out['cpu_baseline'] = {'value': 1.0 / cpu_t, 'unit': 'iters/s', 'cores': cores, 'kind': 'port', 'extrapolated': True,
                               'sample': 'oracle gpmp2_iteration (dense fp64 restatement) at B=%d: %.2f s per iteration, scaled x%d '
                                         'to B=%d (the dense system of the full batch does not fit in memory)' % (Bc, min(ts), B // Bc, B)}
    return out


def bench_h128(dev, steps, with_cpu=True):
    """STOMP at H = 128 (two 64-waypoint chunks per rollout), P = 128, S = 32, d = 14: the generalised persistent kernel
    (csrc/mpb_stomp_fused_hx.hip) next to the two-kernel path."""
    from motion_planning_baselines_amd import ops
    wl, cost, pl = make_stomp(128, 32, dev, 0, H=128)
    path = pl.run_path()
    m0 = pl._particle_means.clone()
    for _ in range(12):     # (the CPU baselines before this entry leave the chip idle: untimed calls of the timed shape first,
        pl.optimize(opt_iters=steps)   # as for the main line -- the first launches after an idle gap run ~10 % slow)
        torch.cuda.synchronize()

    def timed(p, k):
        ts = []
        for _ in range(5):
            p._particle_means.copy_(m0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p.optimize(opt_iters=k)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[2]
    t = timed(pl, steps)
    two = STOMP_two_kernel(wl, cost, dev, 0, 128)
    two.optimize(opt_iters=steps)
    t2 = timed(two, steps)
    P, S, H, d = 128, 32, 128, wl['means0'].shape[-1]
    it_s = t / steps
    alg = 4 * (P * S * H * d + 2 * P * S)                  # per iteration of the persistent launch: samples, costs, weights written
    roof = {'bound': 'hbm', 'achieved': alg / it_s / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg / it_s / 1e9 / HBM_PEAK_GBS,
            'algorithmic_bytes_per_launch': alg, 'traffic': None}
    pmc, pmc_file = latest_profile('r*_pmc_stomp_h128.json')
    if pmc and pmc.get('SQ_INSTS_VALU_per_wave_iteration') and pmc.get('waves_per_launch'):
        gi = pmc['SQ_INSTS_VALU_per_wave_iteration'] * pmc['waves_per_launch'] / it_s / 1e9
        roof = {'bound': 'valu', 'achieved': gi, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instr/s', 'frac': gi / VALU_PEAK_GINSTR,
                'valu_instructions_per_wave_iteration': pmc['SQ_INSTS_VALU_per_wave_iteration'], 'waves_per_launch': pmc['waves_per_launch'],
                'mfma_instructions_per_wave_iteration': pmc.get('SQ_INSTS_MFMA_per_wave_iteration'),
                'kernel': pmc.get('kernel'), 'pmc_source': pmc_file, **pmc_freshness(pmc), 'traffic': pmc.get('hbm_bytes_per_iteration'), 'hbm': roof}
    out = {'workload': 'panda_spheres STOMP B=4096 (P=128 x S=32) H=128 D=7 d=14, %d iterations per call' % steps,
           'metric': 'stomp_trajectory_update_iters_per_sec', 'value': steps / t, 'unit': 'iters/s', 'ms_per_step': 1e3 * t / steps,
           'path': {ops.STOMP_PATH_TWO_KERNEL: 'two-kernel', ops.STOMP_PATH_PERSISTENT_EXCHANGE: 'persistent (exchange)',
                    ops.STOMP_PATH_PERSISTENT: 'persistent'}[path],
           'two_kernel_path_ms_per_step': 1e3 * t2 / steps, 'dtype': 'f32', 'roofline': roof}
    if with_cpu:
        tc, cores, nit = cpu_baseline_sample(wl, pl.scale_tril.cpu(), pl.Sigma.cpu(), P)
        out['cpu_baseline'] = {'value': 1.0 / tc, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
                               'sample': 'oracle stomp_iteration on the full workload (P=%d x S=%d, H=%d), median of %d iterations = %.3f s'
                                         % (P, S, H, nit, tc)}
    return out


def bench_mppi(dev, steps, NP=1024, with_cpu=True):
    """MPPI on NP independent point-mass problems (S = 32 control samples, T = 64 steps, c = 2; the reference example's
    shape, examples/pointmass_grid_circles_2d_MPPI.py:58-67) with the collision shift: one workgroup per problem."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.planners.priors.gaussian import const_ctrl_Cov
    S, T, c = 32, 64, 2
    f = lambda a: torch.as_tensor(a, dtype=torch.float32).contiguous().to(dev)
    Cov = const_ctrl_Cov([0.3, 0.3], T, c, dict(device='cpu', dtype=torch.float32))            # (T,T,c)
    tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
    cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
    gen = torch.Generator().manual_seed(0)
    state0 = f(torch.rand(NP, c, generator=gen) * 0.2 - 0.9)
    goal = f(torch.rand(NP, c, generator=gen) * 0.2 + 0.7)
    geom = ops.DeviceGeometry(G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d(), dev)
    mean = torch.zeros(NP, T, c, device=dev)
    controls, states = torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, T, c, device=dev)
    costs, weights = torch.empty(NP, S, device=dev), torch.empty(NP, S, device=dev)
    # MPPI._save_best runs in every iteration of the reference's loop (mppi.py:148, :164-168): the entry tracks the best sample too
    # (until round 5 it passed no best buffers, which skips that phase)
    track_best = os.environ.get('MPB_MPPI_BENCH_NO_BEST') is None
    best_cost = torch.empty(NP, device=dev) if track_best else None
    best_states = torch.zeros(NP, T, c, device=dev) if track_best else None

    def run(k):
        mean.zero_()
        if track_best:
            best_cost.fill_(3.0e38)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.mppi_step(mean, None, tril, cinv, state0, goal, f([-1., -1.]), f([1., 1.]), torch.ones(T, device=dev),
                      f([1., 1., 1., 100.]), geom, controls, states, costs, weights, 0.04, k_sigma=1e6, weight=1.0, temp=1.0,
                      step_size=0.7, n_iters=k, seed=3, best_cost=best_cost, best_states=best_states)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    for _ in range(6):      # (untimed calls of the timed shape first: see bench_h128)
        run(steps)
    t = sorted(run(steps) for _ in range(5))[2]
    alg = 4 * (2 * S * T * c + 2 * T * c + 2 * S)          # bytes per problem and iteration
    roof = {'bound': 'hbm (nominal)', 'achieved': alg * NP * steps / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': alg * NP * steps / t / 1e9 / HBM_PEAK_GBS,
            'note': 'controls + states written per iteration; the kernel keeps a problem in one workgroup (wave = sample)'}
    pmc, pmc_file = latest_profile('r*_mppi_pmc_mppi.json')
    if pmc and pmc.get('SQ_INSTS_VALU_per_wave_iteration') and NP == 1024:
        # what binds it: fp32 VALU issue (instruction count of the committed rocprofv3 PMC passes of this very entry,
        # scripts/prof_mppi.py, x the waves of a launch / the time measured here)
        ginstr = pmc['SQ_INSTS_VALU_per_wave_iteration'] * pmc['waves_per_launch'] / (t / steps) / 1e9
        roof = {'bound': 'valu', 'achieved': ginstr, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instr/s', 'frac': ginstr / VALU_PEAK_GINSTR,
                'valu_instructions_per_wave_iteration': pmc['SQ_INSTS_VALU_per_wave_iteration'], 'pmc_source': pmc_file, **pmc_freshness(pmc),
                'lds_bank_conflict_frac_of_lds_cycles': (pmc['SQ_LDS_BANK_CONFLICT_per_wave_iteration'] / pmc['SQ_LDS_IDX_ACTIVE_per_wave_iteration']
                                                         if pmc.get('SQ_LDS_IDX_ACTIVE_per_wave_iteration') else None),
                'scratch_bytes_per_lane': pmc.get('scratch_bytes'), 'hbm_nominal': roof}
    out = {'workload': 'MPPI point mass, %d problems x S=%d samples x T=%d steps x c=%d, %d iterations per launch' % (NP, S, T, c, steps),
           'metric': 'mppi_problem_iterations_per_sec', 'value': NP * steps / t, 'unit': 'problem-iters/s',
           'ms_per_step': 1e3 * t / steps, 'us_per_problem_iteration': 1e6 * t / steps / NP, 'dtype': 'f32',
           'save_best': bool(track_best), 'roofline': roof}
    if with_cpu:
        # CPU: the oracle's sequential rollout (mppi.py:131-209, point.py:102-226, quirk Q6: two passes per iteration, the first
        # for the collision shift) on the first 32 problems, one after the other as the reference class would run them
        from oracle import planners_ref as O
        from oracle.geometry_ref import make_ref_geometry
        cores = cpu_threads()
        cta = dict(device='cpu', dtype=torch.float32)
        rr, rf = make_ref_geometry(G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d(), cta)
        trc, cic, s0c, glc = tril.cpu(), cinv.cpu(), state0.cpu(), goal.cpu()
        cmin, cmax, disc = torch.tensor([-1., -1.]), torch.tensor([1., 1.]), torch.ones(T)
        cw = dict(pos=1.0, vel=1.0, ctrl=1.0, pos_T=100.0)
        n_cpu, t0 = 32, None
        for p in range(n_cpu + 2):
            if p == 2:
                t0 = time.perf_counter()             # (two untimed problems first)
            e = torch.randn(c, S, T)
            m = torch.zeros(T, c)
            pre = O.mppi_iteration(m, e, trc, cic, s0c[p], glc[p], 0.04, cmin, cmax, cw, disc, 1.0, 0.7, c)
            q = pre['states'][:, 1:, :2]
            shift = 1e6 * float(rf.compute_cost(q, rr.fk_map_collision(q)).sum())
            O.mppi_iteration(m, e, trc, cic, s0c[p], glc[p], 0.04, cmin, cmax, cw, disc, 1.0, 0.7, c, shift_cost=shift)
        tc = (time.perf_counter() - t0) / n_cpu
        out['cpu_baseline'] = {'value': 1.0 / tc, 'unit': 'problem-iters/s', 'cores': cores, 'kind': 'port',
                               'sample': 'oracle mppi_iteration (sequential rollout + collision shift) on %d problems x S=%d, one iteration '
                                         'each, one problem after the other: %.4f s per problem-iteration' % (n_cpu, S, tc)}
    return out


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching the GPU (no HIP call, no torch.cuda call): the KFD topology nodes with
    SIMDs (CPU nodes have simd_count 0), cut down by a *_VISIBLE_DEVICES list if one is set.  None when the topology cannot be read."""
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None and v.strip() != '':
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) started bare, the way the driver starts the N = 1 line: this process has not touched
    the GPU (no torch.cuda call, no library load) and never will -- it starts the N ranks as FRESH child processes through
    `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1), relays what they print (rank 0's JSON line)
    and returns the launcher's exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    backend = os.environ.get('MPB_DIST_BACKEND', 'nccl')
    have = visible_gpu_count()
    if backend == 'nccl' and have is not None and have < n:
        # (RCCL: one rank per GPU.  Said here, before any rank starts, rather than as N tracebacks from set_device)
        print('bench.py: --gpus %d but this node shows %d GPU(s) (KFD topology%s); run with --gpus <= %d, or with MPB_DIST_BACKEND=gloo '
              'to rehearse several ranks on one GPU' % (n, have, ', *_VISIBLE_DEVICES applied' if any(
                  os.environ.get(v) for v in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')) else '', max(have, 1)),
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=9)
    ap.add_argument('--preheat', type=int, default=-1, help='untimed K-step blocks before the warm-up (default: by K)')
    ap.add_argument('--particles', type=int, default=128)
    ap.add_argument('--samples', type=int, default=32)
    ap.add_argument('--pos-only', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='main line only (profiling runs)')
    ap.add_argument('--main-only', action='store_true', help='(rocprofv3 stats pass) no side measurements either: the only launches of the '
                    'persistent kernel are the pre-heat, warm-up and timed K-step launches, so the trace\'s average is the timed launch')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))     # (nothing has touched the GPU yet: the ranks are fresh child processes)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, which launches its own N ranks, '
                 'or under torch.distributed.run with --nproc-per-node N)' % (args.gpus, world))
    # one rank per GPU.  MPB_DIST_BACKEND=gloo (test aid) lets several ranks share one GPU to exercise this
    # path on a single-GPU box; the default is RCCL ("nccl") over xGMI
    backend = os.environ.get('MPB_DIST_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    # MPB_FORCE_DIST=1: take the distributed path at world size 1 too (RCCL rehearsal on a one-GPU box: process group on
    # "nccl", barrier, all-gather of the means and the MAX all-reduce of the clock all execute)
    forced = os.environ.get('MPB_FORCE_DIST') == '1'
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    clock = Clock(dist, dev)

    P, S = args.particles, args.samples
    wl, cost, planner = make_stomp(P, S, dev, rank, args.pos_only)
    prm = wl['params']
    H, d, D = prm['n_support_points'], wl['means0'].shape[-1], 7
    preheat = args.preheat if args.preheat >= 0 else max(8, min(60, 1200 // max(args.steps, 1)))
    blocks = run_stomp(planner, clock, dist, world, args.steps, args.warmup, args.repeats, preheat=preheat, cold=not args.main_only)
    cold_block_s = run_stomp.cold_block_s
    elapsed, sp = spread(blocks, args.steps)
    timed_launch_ms = run_stomp.launch_ms[len(run_stomp.launch_ms) // 2]    # median over the R event-timed launches
    span = run_stomp.device_span_ms
    span_ms = span[len(span) // 2] if span else None                        # median over the R TIMED launches themselves

    # ---- the dominant kernel, measured live with events on the launch stream.  The whole loop is ONE launch of the
    # persistent kernel (csrc/mpb_stomp_fused.hip): its duration / K is the per-iteration kernel time; a launch of
    # 2K iterations minus a launch of K removes the fixed part (launch + constants into LDS)
    geom = cost.cost_l[0].device_geometry(dev)
    n_prof = max(args.steps, 50)

    def launch_ms(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            planner._particle_means.copy_(means_init)
            torch.cuda.synchronize()
            e0.record()
            planner.optimize(opt_iters=k)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    means_init = wl['means0'].clone()
    # the launch of the timed region: K iterations from the initial means.  (Iterations get cheaper as the trajectories
    # leave the obstacles -- fewer broad-phase candidates: the K iterations after these run ~10 % faster; `later_ms`.)
    # one iteration of the timed launch, its fixed part included: the device-clock span of the timed launches themselves
    # where the kernel reports it (persistent path), else the event figure
    k_ms = (span_ms if span_ms else timed_launch_ms) / args.steps
    later_ms = launch_fixed_ms = two_ms = None
    if not args.main_only:
        t2a, t2b = launch_ms(n_prof), launch_ms(2 * n_prof)
        later_ms = (t2b - t2a) / n_prof                        # one of iterations n_prof .. 2 n_prof
        l1, l2 = launch_ms(1), launch_ms(2)
        launch_fixed_ms = max(2 * l1 - l2, 0.0)                # launch + constants into LDS + first draws: t(1) - (t(2) - t(1))
        # the two-kernel path (mpb_stomp_step: sample + cost kernel, update kernel per iteration) on the same problem
        two = STOMP_two_kernel(wl, cost, dev, rank, P)
        two.optimize(opt_iters=200)
        torch.cuda.synchronize()
        tw = []
        for _ in range(5):      # median of 5: a Python-driven loop of short launches occasionally catches a ~70 ms device stall
            t0 = time.perf_counter()
            two.optimize(opt_iters=args.steps)
            torch.cuda.synchronize()
            tw.append(time.perf_counter() - t0)
        two_ms = 1e3 * sorted(tw)[2] / args.steps
        del two
    # algorithmic bytes of one iteration of the persistent kernel: samples written, costs + weights written; the means
    # and every constant stay in LDS (SURVEY 8d's formula additionally counts the means read + written per iteration)
    alg_bytes_k = 4 * (P * S * H * d + 2 * P * S)
    hbm_gbs = alg_bytes_k / (k_ms * 1e-3) / 1e9
    # instruction counts and HBM-side bytes: rocprofv3 PMC passes of this same workload, committed under profiles/
    # (counters cannot be read from inside the process being profiled)
    pmc, pmc_file = latest_profile('r*_pmc_stomp.json')
    c3_shape = P == 128 and S == 32 and not args.pos_only
    valu = pmc.get('SQ_INSTS_VALU_per_wave_iteration') if (pmc and c3_shape) else None
    traffic = pmc.get('hbm_bytes_per_iteration') if (pmc and c3_shape) else None
    if valu:
        ginstr = valu * (P * S) / (k_ms * 1e-3) / 1e9
        roof = {'bound': 'valu', 'achieved': ginstr, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instr/s',
                'frac': ginstr / VALU_PEAK_GINSTR, 'valu_instructions_per_wave_iteration': valu, 'pmc_source': pmc_file,
                **pmc_freshness(pmc),
                'frac_note': 'vector instructions ISSUED per second over the issue peak (one wave-instruction per 2 cycles per SIMD): it falls when a '
                             'round removes instructions faster than time (0.49 in round 4 at 2 045 instructions and 13.8 us per iteration of the timed launch; %.2f now at %d); ' % (ginstr / VALU_PEAK_GINSTR, round(valu)) +
                             '`hbm.frac` is the algorithmic figure.  By phase (profiles/r05_stamps_fused.txt, diagnostic build): the collision-cost phase '
                             '-- 54 % of an iteration -- issues one instruction per 2.43 cycles per SIMD, the rate scripts/microbench_rates.hip measures '
                             'for its opcode mix; draw + noise product 19 % (vector and matrix work of a SIMD in turn); ~14 % synchronisation / latency'}
    else:   # no committed counter summary for this shape: the nominal bound
        roof = {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / HBM_PEAK_GBS}
    mfma = pmc.get('SQ_INSTS_MFMA_per_wave_iteration') if (pmc and c3_shape) else None
    if valu and mfma is not None:
        # v_mfma_f32_16x16x4_f32 does not run UNDER fp32 VALU work on gfx950 (profiles/r03_microbench_overlap.txt: an MFMA wave
        # and an FMA wave on one SIMD take the SUM of their times, and so does one interleaved stream): the matrix product of
        # the noise occupies the same fp32 pipe for 32 cycles per instruction.  Pipe cycles per wave-iteration =
        # 2 x VALU + 32 x MFMA (every VALU instruction counted at the full rate although v_min / v_max / v_cvt / 3-source v_fma / ... take 4 cycles and v_sqrt / v_sin 8,
        # profiles/r03_microbench_rates.txt: a lower bound of the occupancy)
        # round 4: the noise product runs as v_mfma_f32_16x16x32_bf16 (16 cycles each, and -- unlike the fp32 MFMA -- beside
        # vector work: SQ_VALU_MFMA_COEXEC_CYCLES > 0); the matrix pipe's busy cycles are taken from the counter itself
        busy = pmc.get('SQ_VALU_MFMA_BUSY_CYCLES_per_launch')
        mfma_cyc = (busy / pmc['waves_per_launch'] / pmc['iterations_per_launch']) if busy else 32.0 * mfma
        cyc = 2.0 * valu + mfma_cyc
        pipe = cyc * (P * S) / (k_ms * 1e-3) / 1e9
        roof['fp32_pipe'] = {'achieved': pipe, 'peak': 1024 * 2.4, 'unit': 'G SIMD-cycles/s', 'frac': pipe / (1024 * 2.4),
                             'mfma_instructions_per_wave_iteration': mfma, 'mfma_busy_cycles_per_wave_iteration': mfma_cyc,
                             'mfma_valu_coexec_cycles_per_launch': pmc.get('SQ_VALU_MFMA_COEXEC_CYCLES_per_launch'),
                             'note': 'fp32 VALU issue (2 cycles per wave-instruction) + the matrix pipe\'s busy cycles (SQ_VALU_MFMA_BUSY_CYCLES: '
                                     '36 bf16 MFMAs at 16 cycles + 4 fp32 MFMAs at 32 per wave-iteration; an upper bound of their cost to the vector '
                                     'pipe since round 4: the bf16 ones partly run beside vector work, see mfma_valu_coexec_cycles); peak = 1024 '
                                     'SIMDs x 2.4 GHz; `frac` above counts the VALU instructions alone, as in rounds 1-2'}
    cls = {c: pmc.get('SQ_INSTS_VALU_%s_per_wave_iteration' % c) for c in ('ADD_F32', 'MUL_F32', 'FMA_F32', 'TRANS_F32', 'INT32', 'INT64', 'CVT')} \
        if (pmc and c3_shape) else {}
    if valu and mfma is not None and all(v is not None for v in cls.values()) and cls:
        # the same pipe with the instruction classes of the PMC passes priced by profiles/r03_microbench_rates.txt (architectural
        # cycles per wave-instruction: add / mul 2, transcendental 8, convert 4, 64-bit integer multiply 4, matrix 32):
        #   low : every class whose members issue at EITHER rate (fma: 2 with two distinct VGPR sources, 4 with three; 32-bit
        #         integer: add / and / xor 2, shifts / bfe / 24-bit mad 4; the rest -- min, max, select, compare, move, lane ops)
        #         at 2: a strict lower bound of the pipe occupancy;
        #   mid : those classes at 3 (fma, int32) and 3.5 (the rest, mostly 4-cycle instructions).
        other = valu - sum(cls.values())
        fixed = 2.0 * (cls['ADD_F32'] + cls['MUL_F32']) + 8.0 * cls['TRANS_F32'] + 4.0 * (cls['INT64'] + cls['CVT']) + mfma_cyc
        low = fixed + 2.0 * (cls['FMA_F32'] + cls['INT32'] + other)
        mid = fixed + 3.0 * (cls['FMA_F32'] + cls['INT32']) + 3.5 * other
        per_s = (P * S) / (k_ms * 1e-3) / 1e9 / (1024 * 2.4)
        roof['fp32_pipe_by_class'] = {'frac_low': low * per_s, 'frac_mid': mid * per_s, 'cycles_per_wave_iteration_low': low,
                                      'cycles_per_wave_iteration_mid': mid, 'instructions_per_wave_iteration': dict(cls, OTHER=other),
                                      'mfma_valu_coexec_cycles_per_launch': pmc.get('SQ_VALU_MFMA_COEXEC_CYCLES_per_launch'),
                                      'note': 'class counters SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / INT32 / INT64 / CVT of the PMC passes, '
                                              'priced per profiles/r03_microbench_rates.txt; peak = 1024 SIMDs x 2.4 GHz; the hardware '
                                              'counts 0 cycles in which a matrix and a vector instruction executed together'}
    if cls and all(v is not None for v in cls.values()) and valu:
        # arithmetic actually done, next to the issue fraction: fp32 flops of the counted classes (fma = 2)
        flop_wave_it = 64.0 * (2.0 * cls['FMA_F32'] + cls['ADD_F32'] + cls['MUL_F32'] + cls['TRANS_F32'])
        tf = flop_wave_it * (P * S) / (k_ms * 1e-3) / 1e12
        roof['fp32_flops'] = {'achieved': tf, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': tf / 157.3,
                              'note': 'SQ_INSTS_VALU_{FMA x 2, ADD, MUL, TRANS}_F32 x 64 lanes per wave-iteration over the kernel time; the '
                                      'rest of the issued instructions are integer / convert / move / select / lane traffic'}
    roof.update({'kernel': 'stomp_fused_kernel<%d, model> (persistent: one launch = all iterations)' % d, 'traffic': traffic,
                 'hbm': {'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / HBM_PEAK_GBS,
                         'algorithmic_bytes_per_launch': alg_bytes_k, 'note': 'per iteration of the persistent launch'},
                 'kernel_ms': k_ms, 'kernel_ms_note': 'duration of the K-iteration persistent launches OF THE TIMED BLOCKS on the device\'s '
                 'real-time counter (first unit started -> last workgroup left; the kernel stamps its pinned status block, no cost), '
                 'median over the R blocks, divided by K; agrees with rocprofv3 --kernel-trace (profiles/)',
                 'kernel_ms_hip_events': timed_launch_ms / args.steps,
                 'kernel_ms_hip_events_note': 'HIP event pair recorded on the dispatch (hipExtLaunchKernelGGL, launch stream) of the same '
                 'launch repeated R times right after the timed blocks; those launches run ~10 % slower than the timed ones (rocprofv3 '
                 'shows the kernels themselves longer: the profiled dispatch and the blocking wait behind it change the clock state)',
                 'launch_fixed_ms': launch_fixed_ms,
                 'kernel_ms_iterations_%d_to_%d' % (n_prof, 2 * n_prof): later_ms,
                 'two_kernel_path_ms_per_step': two_ms, 'two_kernel_path_iters_per_sec': (1e3 / two_ms) if two_ms else None})

    # ---- `k1`: the reference examples' calling pattern -- ONE iteration per optimize() call, each call waited for
    # (check='sync', the planner's default; examples/pointmass_grid_circles_2d_STOMP.py:104-106: `for i in range(opt_iters):
    # trajs = planner.optimize(opt_iters=1)`): every iteration pays the launch, the constants into LDS and the host side of a call
    k1 = None
    if not args.main_only and world == 1:
        n_k1 = 150
        planner.check = 'sync'
        for _ in range(20):
            planner.optimize(opt_iters=1)
        ts = []
        for _ in range(7):       # median of 7 short batches: a Python-driven loop of short launches occasionally catches a ~70 ms device stall
            planner._particle_means.copy_(means_init)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_k1):
                planner.optimize(opt_iters=1)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / n_k1)
        planner.check = 'deferred'
        t1 = sorted(ts)[len(ts) // 2]
        k1 = {'value': 1.0 / t1, 'unit': 'iters/s', 'ms_per_call': 1e3 * t1, 'ms_per_call_min': 1e3 * min(ts), 'ms_per_call_max': 1e3 * max(ts),
              'calls': n_k1, 'batches': len(ts), 'check': 'sync',
              'clears_10k_target': bool(1.0 / t1 >= 1e4),
              'note': 'one STOMP iteration per optimize() call, every call waited for before the next (the loop of the reference\'s '
                      'examples, examples/pointmass_grid_circles_2d_STOMP.py:104-106): ms_per_call = one steady iteration + '
                      '`roofline.launch_fixed_ms` (launch + constants into LDS + first draw, t(1) - (t(2) - t(1)) on stream events) + the host '
                      'side of a call (Python + ctypes + the wait); `value` of the main line amortises that fixed part over K steps'}

    # ---- parity of the headline workload against the oracle (GPU side here, untimed; the oracle side rides on the CPU baseline)
    par_gpu = eps_parity = par_philox_gpu = None
    consts = (planner.scale_tril.cpu(), planner.Sigma.cpu())
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.main_only:
        eps_parity = torch.randn(2, S, d, P, H, generator=torch.Generator().manual_seed(1234))
        par_gpu = stomp_parity_gpu(wl, planner, cost, geom, eps_parity)
        par_philox_gpu = stomp_parity_philox_gpu(wl, planner, cost, geom, n_it=max(2, min(args.steps, 20)))
    seeded = generic = crowded = None
    if rank == 0 and world == 1 and not args.main_only and not args.no_other_configs:
        seeded = bench_seeded(wl, cost, dev, rank, P, args.steps)
        generic = bench_generic_model(wl, dev, rank, P, S, args.steps)
        crowded = bench_crowded(dev, rank, P, S, args.steps)

    # ---- BASELINE configs[4]'s per-GPU load with the same protocol (every N)
    c5 = None
    if not args.no_other_configs and not args.main_only:
        del planner
        torch.cuda.empty_cache()
        P5 = 4096
        k5 = max(1, min(args.steps, 50))
        wl5, cost5, pl5 = make_stomp(P5, S, dev, rank, args.pos_only)
        b5 = run_stomp(pl5, clock, dist, world, k5, max(1, min(args.warmup, 5)), args.repeats, preheat=2)
        el5, sp5 = spread(b5, k5)
        c5 = {'workload': 'panda_spheres STOMP, %d particles x S=%d = %d rollouts per GPU, %d particles in the job '
                          '(BASELINE configs[4] is this load on 8 GPUs = 32768 problems)' % (P5, S, P5 * S, world * P5),
              'metric': 'stomp_trajectory_update_iters_per_sec', 'value': world * k5 / el5, 'unit': 'iters/s',
              'steps': k5, 'ms_per_step': 1e3 * el5 / k5, 'repeats': sp5, 'scaling': 'weak',
              'particle_updates_per_sec': world * P5 * k5 / el5, 'rollouts_per_sec': world * P5 * S * k5 / el5,
              'hbm_frac': stomp_algorithmic_bytes(P5, S, H, d) * k5 / el5 / 1e9 / HBM_PEAK_GBS}
        # its roofline: the two-batch instantiation stomp_fused_kernel<d, model, 2> has its own PMC passes (scripts/profile_c5.sh)
        it5 = el5 / k5
        alg5 = 4 * (P5 * S * H * d + 2 * P5 * S)          # per iteration of the persistent launch (means / constants stay in LDS)
        roof5 = {'bound': 'hbm', 'achieved': alg5 / it5 / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg5 / it5 / 1e9 / HBM_PEAK_GBS,
                 'algorithmic_bytes_per_launch': alg5, 'traffic': None}
        pmc5, pmc5_file = latest_profile('r*_pmc_stomp_c5.json')
        if pmc5 and pmc5.get('SQ_INSTS_VALU_per_wave_iteration') and pmc5.get('waves_per_launch') and not args.pos_only:
            # waves of a launch x iterations: a wave of the two-batch layout runs TWO rollouts per iteration
            gi5 = pmc5['SQ_INSTS_VALU_per_wave_iteration'] * pmc5['waves_per_launch'] / it5 / 1e9
            roof5 = {'bound': 'valu', 'achieved': gi5, 'peak': VALU_PEAK_GINSTR, 'unit': 'G wave-instr/s', 'frac': gi5 / VALU_PEAK_GINSTR,
                     'valu_instructions_per_wave_iteration': pmc5['SQ_INSTS_VALU_per_wave_iteration'], 'waves_per_launch': pmc5['waves_per_launch'],
                     'kernel': pmc5.get('kernel'), 'pmc_source': pmc5_file, **pmc_freshness(pmc5), 'traffic': pmc5.get('hbm_bytes_per_iteration'), 'hbm': roof5}
        c5['roofline'] = roof5
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            n5 = 128
            t5, cores5, nit5 = cpu_baseline_sample(wl5, pl5.scale_tril.cpu(), pl5.Sigma.cpu(), n5)
            c5['cpu_baseline'] = {'value': 1.0 / (t5 * (P5 / n5)), 'unit': 'iters/s', 'cores': cores5, 'kind': 'port', 'extrapolated': True,
                                  'sample': 'oracle stomp_iteration on the first %d of the %d particles (x S=%d), median of %d iterations = '
                                            '%.3f s, scaled x%d (particles are independent: the oracle\'s cost is linear in P)'
                                            % (n5, P5, S, nit5, t5, P5 // n5)}
        del pl5, cost5, wl5
        torch.cuda.empty_cache()

    if rank == 0:
        its = world * args.steps / elapsed
        line = {
            'metric': 'stomp_trajectory_update_iters_per_sec',
            'value': its, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'repeats': sp, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'panda_spheres STOMP B=%d (P=%d particles x S=%d samples) H=%d D=%d d=%d per GPU'
                                   % (P * S, P, S, H, D, d),
                       'pos_only': bool(args.pos_only), 'noise': 'device philox', 'sigma_coll': wl['sigma_coll'],
                       'robot_collision_spheres': 31, 'collision_spheres_after_static_pruning': int(geom.host.view('int32')[5]),
                       'obstacle_spheres': 16, 'parallelism': 'particles sharded x%d' % world,
                       'algorithmic_bytes_per_iter': stomp_algorithmic_bytes(P, S, H, d)},
            'roofline': roof,
            'preheat': {'untimed_blocks': preheat, 'steps_per_block': args.steps,
                        'note': 'untimed K-step blocks of the timed shape (synchronize between them) before the W warm-up steps: the '
                                'chip is power-managed, the timed blocks measure the steady state of the protocol itself (DESIGN 6.2)'},
            'scaling_note': 'weak scaling, 128 particles per GPU on this line at every N; its N-GPU ratio is capped at ~0.92 by the one '
                            'fixed-cost collective (the final all-gather, ~26 us) inside a ~0.36 ms block at --steps 20.  The entry that '
                            'answers ">= 6x at 8 GPUs" is `c5` (BASELINE configs[4]: 4096 particles per GPU, 25 ms blocks): compare c5.value across N',
        }
        if dist is not None:
            line['dist'] = {'backend': dist.get_backend(), 'world': world, 'rccl_ranks_seen': dist.get_world_size(), 'forced_at_world_1': bool(forced and world == 1),
                            'collectives_in_timed_region': 'ONE: the final all_gather_into_tensor of the (P,H,d) means, which is also the closing barrier of the '
                            'block (it completes on no rank before every rank has contributed, i.e. finished its K steps; the host spins on an '
                            'event behind it, then synchronizes); the opening barrier (all_reduce of one element) and the all_reduce(MAX) of '
                            'the clock are outside'}
        if cold_block_s is not None:
            line['cold'] = {'ms_per_step': 1e3 * cold_block_s / args.steps, 'value': args.steps / cold_block_s, 'unit': 'iters/s',
                            'note': 'the FIRST timed K-step block of the process with no pre-heat (what `--preheat 0` times first): only '
                                    'the W warm-up steps precede it; the main line is the steady state after `preheat.untimed_blocks` such blocks'}
        if k1 is not None:
            line['k1'] = k1
        if seeded is not None:
            line['seeded'] = seeded
        if generic is not None:
            line['generic_model'] = generic
        if crowded is not None:
            line['crowded'] = crowded
        if c5 is not None:
            line['c5'] = c5
        if world == 1 and not args.no_cpu_baseline:
            med, cores, n_it, kept = cpu_baseline_stomp(wl, consts[0], consts[1], eps_parity if eps_parity is not None else [])
            if par_gpu is not None:
                line['parity'] = stomp_parity(wl, consts[0], consts[1], eps_parity, par_gpu, kept)
            if par_philox_gpu is not None:
                line.setdefault('parity', {})['philox'] = stomp_parity_philox(wl, consts[0], consts[1], par_philox_gpu)
            line['cpu_baseline'] = {
                'value': 1.0 / med, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
                'sample': 'oracle/planners_ref.py stomp_iteration (PyTorch-CPU restatement of stomp.py:157-160 + build-defined '
                          'FK/SDF) on the full workload (P=%d x S=%d), median of %d iterations = %.3f s, %d intra-op threads'
                          % (P, S, n_it, med, cores)}
        if world == 1 and not args.no_other_configs and not args.main_only:
            line['c2'] = bench_c2(dev, 500, with_cpu=not args.no_cpu_baseline)
            line['c4'] = bench_c4(dev, 10, with_cpu=not args.no_cpu_baseline)
            line['h128'] = bench_h128(dev, 50, with_cpu=not args.no_cpu_baseline)
            line['mppi'] = bench_mppi(dev, 50, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
