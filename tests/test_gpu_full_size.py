"""GPU, BASELINE full sizes: size-independent properties of the HIP path (the oracle cannot run these
sizes in seconds): sharding invariance, fused == split, consistency between kernels, idempotence,
normalisation, monotone behaviour."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _c3(dev, P=128, S=32, pos_only=False):
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    wl = workloads.panda_spheres_stomp(P, dev, S=S, pos_only=pos_only)
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(64, wl['params']['dt'], 0.5, cpu)
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    return wl, Sigma, L, geom


def test_stomp_c3_sharded_equals_unsharded(gpu_device):
    """C3/C5: particles [0,128) in one call == two shards of 64 with particle_offset (device Philox noise):
    the result must not depend on how the problems are split over GPUs."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    P, S, H = 128, 32, 64
    wl, Sigma, L, geom = _c3(dev, P, S)
    d = wl['means0'].shape[-1]
    mk = lambda p: (torch.empty(p, S, H, d, device=dev), torch.empty(p, S, device=dev), torch.empty(p, S, device=dev))
    full = wl['means0'].clone()
    s, c, w = mk(P)
    ops.stomp_step(full, None, s, c, w, L, Sigma, geom, S, 7, 1.0, 1.0, 0.1, 1.0, n_iters=5, seed=11)
    halves = []
    for lo in (0, 64):
        m = wl['means0'][lo:lo + 64].clone()
        s2, c2, w2 = mk(64)
        ops.stomp_step(m, None, s2, c2, w2, L, Sigma, geom, S, 7, 1.0, 1.0, 0.1, 1.0, n_iters=5, seed=11,
                       particle_offset=lo)
        halves.append((m, s2, c2, w2))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([h[0] for h in halves]), full)
    assert torch.equal(torch.cat([h[1] for h in halves]), s)
    assert torch.equal(torch.cat([h[2] for h in halves]), c)
    assert torch.allclose(w.sum(1), torch.ones(P, device=dev), atol=1e-5)
    assert torch.isfinite(full).all()


@pytest.mark.parametrize('pos_only', [False, True])
def test_stomp_c3_kernels_consistent(gpu_device, pos_only):
    """At B=4096: the fused cost equals the stand-alone cost kernel on the samples it wrote; the fused step
    equals sample -> update; the noise rows 0 and H-1 are exactly zero; a zero-lr step is idempotent."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    P, S, H = 128, 32, 64
    wl, Sigma, L, geom = _c3(dev, P, S, pos_only)
    d = wl['means0'].shape[-1]
    means = wl['means0'].clone()
    samples = torch.empty(P, S, H, d, device=dev)
    costs = torch.empty(P, S, device=dev)
    weights = torch.empty(P, S, device=dev)
    ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, seed=5)
    alone = ops.cost_collision_eval(samples.flatten(0, 1), geom, 1e6).reshape(P, S)
    m2 = wl['means0'].clone()
    s2, w2 = torch.empty_like(samples), torch.empty_like(weights)
    ops.stomp_sample(m2, None, s2, L, S, seed=5, it=0)
    ops.stomp_update(m2, s2, alone, w2, Sigma, 0.1, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(alone, costs)
    assert torch.equal(s2, samples) and torch.equal(m2, means) and torch.equal(w2, weights)
    assert torch.equal(samples[:, :, 0], wl['means0'][:, None, 0].expand(-1, S, -1))
    assert torch.equal(samples[:, :, -1], wl['means0'][:, None, -1].expand(-1, S, -1))
    m3 = means.clone()
    ops.stomp_step(m3, None, samples, costs, weights, L, Sigma, geom, S, 7, 1e6, 1.0, 0.0, 1.0, seed=6)
    assert torch.equal(m3, means)


@pytest.mark.parametrize('keep_all', [False, True])
def test_panda_model_path_equals_generic_walk(gpu_device, keep_all):
    """The compile-time Panda model (csrc/mpb_model_panda.h: unrolled chain, folded transforms) and the generic
    table-driven chain walk return the same bits: fused STOMP cost and stand-alone cost on 131072 waypoints incl.
    far-out-of-range angles, with the statically pruned link table and with the full one (frame-1 group live)."""
    import numpy as np
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    P, S, H = 16, 32, 64
    wl, Sigma, L, _ = _c3(dev, P, S)
    gm = ops.DeviceGeometry(wl['robot'], wl['field'], dev, keep_all_links=keep_all)
    gg = ops.DeviceGeometry(wl['robot'], wl['field'], dev, keep_all_links=keep_all, use_model=False)
    assert int(gm.host.view(np.int32)[29]) == 1 and int(gg.host.view(np.int32)[29]) == 0
    n_kept = int(gm.host.view(np.int32)[5])
    assert int(gm.host.view(np.uint32)[30]) == ((1 << 31) - 1 if keep_all else ((1 << 31) - 1) & ~((1 << (31 - n_kept)) - 1))
    d = wl['means0'].shape[-1]
    out = []
    for geom in (gm, gg):
        means = wl['means0'].clone()
        samples = torch.empty(P, S, H, d, device=dev)
        costs = torch.empty(P, S, device=dev)
        ops.stomp_sample(means, None, samples, L, S, seed=3, it=1, geom=geom, costs=costs, k_sigma=1e6)
        g = torch.Generator().manual_seed(0)
        x = ((torch.rand(2048, 64, 14, generator=g) * 2 - 1) * 6.0).to(dev)
        alone, pw = ops.cost_collision_eval(x, geom, 1.0, per_waypoint=True)
        torch.cuda.synchronize()
        out.append((samples, costs, alone, pw))
    assert float(out[0][1].max()) > 0 and float(out[0][2].max()) > 0
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize('keep_all', [False, True])
def test_panda_model_gradient_walk_equals_generic_walk(gpu_device, keep_all):
    """The compile-time Panda model of the GRADIENT evaluators (waypoint_cost_grid_grad_model: stand-alone cost + gradient,
    GPMP2's linearisation incl. interpolated points, CHOMP's loop) and the table-driven walk return the same bits:
    131072 waypoints incl. far-out-of-range angles, pruned and full link tables."""
    import numpy as np
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    wl, _, _, _ = _c3(dev, 8, 4)
    gm = ops.DeviceGeometry(wl['robot'], wl['field'], dev, keep_all_links=keep_all)
    gg = ops.DeviceGeometry(wl['robot'], wl['field'], dev, keep_all_links=keep_all, use_model=False)
    assert (gm.flags & 0xFF) == 1 and (gg.flags & 0xFF) == 0
    g = torch.Generator().manual_seed(1)
    x = ((torch.rand(2048, 64, 14, generator=g) * 2 - 1) * 6.0).to(dev)
    x[:1024] *= 0.4                                          # half of them near the workspace centre: many contacts
    outs = []
    for geom in (gm, gg):
        c, gr = ops.cost_collision_grad(x, geom, 1.0)
        rows = ops.gpmp2_collision_rows(x[:256], geom, n_interp=0)
        rows_i = ops.gpmp2_collision_rows(x[:64], geom, n_interp=2)
        torch.cuda.synchronize()
        outs.append((c, gr, rows, rows_i))
    assert float(outs[0][0].max()) > 0 and float(outs[0][1].abs().max()) > 0
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # CHOMP on the Panda: five iterations of the one-launch loop with either geometry
    from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
    R = chomp_precision_matrix(dt=5 / 64, n_support_points=64, tensor_args=dict(device='cpu', dtype=torch.float32)).to(dev).contiguous()
    res = []
    for geom in (gm, gg):
        m = (x[:512] * 0.3).contiguous().clone()
        ops.chomp_step(m, R, geom, 7, 1.0, 10.0, 1e-6, 0.05, 0.05, n_iters=5)
        torch.cuda.synchronize()
        res.append(m)
    assert torch.isfinite(res[0]).all() and torch.equal(res[0], res[1])


def test_collision_cost_grid_equals_exhaustive_at_scale(gpu_device):
    """Broad-phase grid (cost-only path) == exhaustive obstacle loop (gradient path's cost output), bit for
    bit, on 4096 x 64 random Panda configurations incl. far-out-of-workspace angles."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    wl, _, _, geom = _c3(dev, 8, 4)
    g = torch.Generator().manual_seed(0)
    x = ((torch.rand(4096, 64, 14, generator=g) * 2 - 1) * 6.0).to(dev)
    a = ops.cost_collision_eval(x, geom, 1.0)
    b, _ = ops.cost_collision_grad(x, geom, 1.0)
    torch.cuda.synchronize()
    assert float(a.max()) > 0
    assert torch.equal(a, b)


def test_chomp_c2_quad_kernel_equals_lean_kernel(gpu_device):
    """C2 runs on the four-lanes-per-waypoint kernel (obstacles in registers, nearest combined over the quad by (signed
    distance, obstacle index)); with the geometry's flag bit 9 cleared the same call takes the one-lane-per-waypoint
    kernel with its exhaustive loop: same arithmetic per obstacle, same first-minimum rule."""
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
    dev = gpu_device
    wl = workloads.pointmass_dense_chomp(1024, dev)
    R = chomp_precision_matrix(wl['params']['dt'], 64, dict(device='cpu', dtype=torch.float32)).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    assert geom.flags & 0x200
    kw = dict(D=2, k_sigma=1.0, weight=10.0, w_prior=1e-4, lr=0.05, grad_clip=0.05)
    a, b = wl['means0'].clone(), wl['means0'].clone()
    ca, cb = torch.empty(1024, device=dev), torch.empty(1024, device=dev)
    ops.chomp_step(a, R, geom, n_iters=1, costs_out=ca, **kw)
    geom.flags &= ~0x200
    ops.chomp_step(b, R, geom, n_iters=1, costs_out=cb, **kw)
    torch.cuda.synchronize()
    assert float(ca.max()) > 0 and not torch.equal(a, wl['means0'])
    assert torch.equal(ca, cb) and torch.equal(a, b)


def test_chomp_c2_fused_equals_stepwise(gpu_device):
    """C2 (B=1024): 50 iterations inside one launch == 50 single-iteration calls, bit for bit; endpoints fixed."""
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
    dev = gpu_device
    wl = workloads.pointmass_dense_chomp(1024, dev)
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    R = chomp_precision_matrix(dt=0.04, n_support_points=64, tensor_args=dict(device='cpu', dtype=torch.float32)).to(dev)
    kw = dict(D=2, k_sigma=1.0, weight=10.0, w_prior=1e-4, lr=0.05, grad_clip=0.05)
    a = wl['means0'].clone()
    ops.chomp_step(a, R, geom, n_iters=50, **kw)
    b = wl['means0'].clone()
    for _ in range(50):
        ops.chomp_step(b, R, geom, n_iters=1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(a[:, 0], wl['means0'][:, 0]) and torch.equal(a[:, -1], wl['means0'][:, -1])
    assert torch.isfinite(a).all()
    # sharding with the global batch size (quirk Q3) reproduces the unsharded result
    c = wl['means0'][:512].clone()
    ops.chomp_step(c, R, geom, n_iters=50, B_global=1024, **kw)
    assert torch.equal(c, a[:512])


@pytest.mark.parametrize('B', [256, 2048])
def test_gpmp2_c4_solve_properties(gpu_device, B):
    """C4 (H=128, D=7) at its full batch B=2048 (352 MB workspace) and at B=256: the Gauss-Newton step must (i) be finite, (ii) leave trajectories that
    already satisfy all factors unchanged (fixed point: straight line, no collision, exact start/goal),
    (iii) reduce the cost b^T K b on colliding problems, (iv) split entry points == single call."""
    from motion_planning_baselines_amd import geometry as G, ops, workloads
    dev = gpu_device
    H, D = 128, 7
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    geom = ops.DeviceGeometry(robot, field, dev)
    q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
    dt = 5.0 / H
    x0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
    z = torch.zeros(B, D, device=dev)
    start = torch.cat([torch.from_numpy(q[:B]).to(dev), z], -1).contiguous()
    goal = torch.cat([torch.from_numpy(q[B:]).to(dev), z], -1).contiguous()
    sig = (1e-5, 1e-2, 1e-5, 1e-5)
    ws = ops.gpmp2_workspace(B, H, D, dev)
    c0, c1 = torch.empty(B, device=dev), torch.empty(B, device=dev)
    x = x0.clone()
    ops.gpmp2_step(x, start, goal, geom, ws, sig, dt, 1e-2, True, 1.0, costs_out=c0)
    ops.gpmp2_step(x, start, goal, geom, ws, sig, dt, 1e-2, True, 1.0, costs_out=c1)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all() and torch.isfinite(c0).all()
    colliding = c0 > 1.0
    assert int(colliding.sum()) > 10
    assert float((c1[colliding] < c0[colliding]).float().mean()) > 0.9
    # without obstacles the problem is exactly quadratic: one (barely damped) Gauss-Newton step lands on the
    # minimiser -- the cost collapses and a second step no longer moves the trajectory
    far = G.CollisionField(spheres=[[50.0, 50.0, 50.0, 0.1]], margin=0.05)
    geom_far = ops.DeviceGeometry(robot, far, dev)
    xl = x0.clone()
    q0, q1, q2 = (torch.empty(B, device=dev) for _ in range(3))
    ops.gpmp2_step(xl, start, goal, geom_far, ws, sig, dt, 1e-6, False, 1.0, costs_out=q0)
    x1 = xl.clone()
    ops.gpmp2_step(xl, start, goal, geom_far, ws, sig, dt, 1e-6, False, 1.0, costs_out=q1)
    ops.gpmp2_step(xl, start, goal, geom_far, ws, sig, dt, 1e-6, False, 1.0, costs_out=q2)
    torch.cuda.synchronize()
    assert float((q1 / q0).max()) < 1e-3, float((q1 / q0).max())
    assert float((xl - x1).abs().max()) < 1e-3
    assert float((x1[:, 0, :D] - x0[:, 0, :D]).abs().max()) < 1e-4      # tight start / goal priors
    assert float((x1[:, -1, :D] - x0[:, -1, :D]).abs().max()) < 1e-4
    # split == single call
    xa, xb = x0.clone(), x0.clone()
    ops.gpmp2_step(xa, start, goal, geom, ws, sig, dt, 1e-2, True, 1.0)
    ops.gpmp2_linearize(xb, geom, ws)
    dsum = ops.gpmp2_diag(ws, B, H, D, sig, dt)
    ops.gpmp2_solve(xb, start, goal, dsum / B, ws, sig, dt, 1e-2, True, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb)


def test_stomp_c5_per_gpu_load(gpu_device):
    """C5's per-GPU load: 4096 particles x 32 samples = 131 072 rollouts per iteration (470 MB of samples).  With the
    device Philox stream keyed by the global particle id, the first 128 particles of the big launch must equal a
    128-particle launch bit for bit (nothing depends on the grid size or on which block a particle lands in), the
    weights are normalised and the costs equal the stand-alone cost kernel on the samples the big launch wrote."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    P, S, H = 4096, 32, 64
    wl, Sigma, L, geom = _c3(dev, P, S)
    d = wl['means0'].shape[-1]
    mk = lambda p: (torch.empty(p, S, H, d, device=dev), torch.empty(p, S, device=dev), torch.empty(p, S, device=dev))
    big = wl['means0'].clone()
    s, c, w = mk(P)
    ops.stomp_step(big, None, s, c, w, L, Sigma, geom, S, 7, 1.0, 1.0, 0.1, 1.0, n_iters=3, seed=5)
    small = wl['means0'][:128].clone()
    s2, c2, w2 = mk(128)
    ops.stomp_step(small, None, s2, c2, w2, L, Sigma, geom, S, 7, 1.0, 1.0, 0.1, 1.0, n_iters=3, seed=5)
    tail = wl['means0'][P - 64:].clone()
    s3, c3, w3 = mk(64)
    ops.stomp_step(tail, None, s3, c3, w3, L, Sigma, geom, S, 7, 1.0, 1.0, 0.1, 1.0, n_iters=3, seed=5, particle_offset=P - 64)
    torch.cuda.synchronize()
    assert torch.isfinite(big).all()
    assert torch.equal(big[:128], small) and torch.equal(s[:128], s2) and torch.equal(c[:128], c2) and torch.equal(w[:128], w2)
    assert torch.equal(big[P - 64:], tail) and torch.equal(s[P - 64:], s3)
    assert torch.allclose(w.sum(1), torch.ones(P, device=dev), atol=1e-5)
    cc = ops.cost_collision_eval(s.flatten(0, 1), geom, 1.0).reshape(P, S)
    assert torch.equal(cc, c)


def test_stomp_c5_persistent_launch_matches_c3_layout(gpu_device):
    """The persistent kernel at C5's per-GPU load (4096 particles: the launcher takes the one-workgroup-per-particle, two-batch
    layout) against 128-particle launches of the same particles (two workgroups per particle exchanging their partials):
    means, samples, costs and weights of those particles equal bit for bit after three iterations -- neither the layout, nor
    the grid size, nor the block a particle lands in changes a result."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    P, S, H = 4096, 32, 64
    wl, Sigma, L, geom = _c3(dev, P, S)
    d = wl['means0'].shape[-1]
    mk = lambda p: (torch.empty(p, S, H, d, device=dev), torch.empty(p, S, device=dev), torch.empty(p, S, device=dev))
    args = (L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1e5)
    big = wl['means0'].clone()
    s, c, w = mk(P)
    ws = ops.stomp_workspace(P, S, H, d, dev)
    ops.stomp_run(big, None, s, c, w, *args, ws, n_iters=3, seed=5)
    torch.cuda.synchronize()
    assert not ops.stomp_run_timed_out(ws)
    assert torch.isfinite(big).all() and float(c.max()) > 0
    assert torch.allclose(w.sum(1), torch.ones(P, device=dev), atol=1e-5)
    ws2 = ops.stomp_workspace(128, S, H, d, dev)
    for lo in (0, 1920, P - 128):
        small = wl['means0'][lo:lo + 128].clone()
        s2, c2, w2 = mk(128)
        ops.stomp_run(small, None, s2, c2, w2, *args, ws2, n_iters=3, seed=5, particle_offset=lo)
        torch.cuda.synchronize()
        assert not ops.stomp_run_timed_out(ws2)
        assert torch.equal(big[lo:lo + 128], small) and torch.equal(s[lo:lo + 128], s2), lo
        assert torch.equal(c[lo:lo + 128], c2) and torch.equal(w[lo:lo + 128], w2), lo


def test_static_link_pruning_changes_nothing(gpu_device):
    """The packed link table without the collision spheres that can never reach an obstacle (pack_geometry's static broad
    phase: 31 -> 28 for the C3 scene) gives bit-identical costs, per-waypoint costs and gradients at the C3 batch size."""
    from motion_planning_baselines_amd import geometry as G, ops, workloads
    dev = gpu_device
    robot, field = G.RobotPanda(), G.env_spheres_3d(seed=0)
    full = ops.DeviceGeometry(robot, field, dev, keep_all_links=True)
    pruned = ops.DeviceGeometry(robot, field, dev)
    assert int(full.host.view(np.int32)[5]) == 31 and int(pruned.host.view(np.int32)[5]) == 28
    g = torch.Generator().manual_seed(4)
    lo, hi = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
    q = (lo + (hi - lo) * torch.rand(4096, 64, 7, generator=g) + 0.5 * torch.randn(4096, 64, 7, generator=g)).to(dev)
    x = torch.cat([q, torch.zeros_like(q)], -1).contiguous()
    c0, pw0 = ops.cost_collision_eval(x, full, 1e6, per_waypoint=True)
    c1, pw1 = ops.cost_collision_eval(x, pruned, 1e6, per_waypoint=True)
    _, g0 = ops.cost_collision_grad(x, full, 1e6)
    _, g1 = ops.cost_collision_grad(x, pruned, 1e6)
    torch.cuda.synchronize()
    assert float(c0.max()) > 0
    assert torch.equal(c0, c1) and torch.equal(pw0, pw1) and torch.equal(g0, g1)


def test_bench_line_contract(gpu_device):
    """bench.py prints ONE JSON line with the driver's fields, consistent with each other (child process: bench.py owns
    its process group / device set-up)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '3',
                          '--no-cpu-baseline'], check=True, capture_output=True, text=True, timeout=600, cwd=root).stdout
    lines = [l for l in out.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    assert j['metric'] == 'stomp_trajectory_update_iters_per_sec' and j['unit'] == 'iters/s'
    assert j['n_gpus'] == 1 and j['steps'] == 20 and j['warmup'] == 3 and j['higher_is_better'] is True
    assert j['scaling'] == 'weak' and j['vs_baseline'] is None and j['dtype'] == 'f32' and j['data'] == 'synthetic'
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert abs(j['value'] * j['ms_per_step'] * 1e-3 - 1.0) < 1e-6              # value = steps / elapsed on one GPU
    r = j['roofline']
    # the binding roofline is the fp32 VALU issue rate when a committed rocprofv3 counter summary of this workload
    # exists (profiles/r*_pmc_stomp.json: instructions per wave and iteration of the persistent kernel), the nominal HBM bound of SURVEY 8d otherwise; the
    # HBM figure is always reported next to it
    assert r['bound'] in ('valu', 'hbm')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    if r['bound'] == 'valu':
        assert r['unit'] == 'G wave-instr/s' and abs(r['peak'] - 1228.8) < 1e-6
        assert abs(r['achieved'] - r['valu_instructions_per_wave_iteration'] * 4096 / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    h = r['hbm']
    assert h['unit'] == 'GB/s' and h['peak'] == 8000.0 and abs(h['frac'] - h['achieved'] / h['peak']) < 1e-9
    assert abs(h['achieved'] - h['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * h['achieved']
    sp = j['repeats']
    assert sp['n'] == 9 and sp['ms_per_step_min'] <= sp['ms_per_step_median'] <= sp['ms_per_step_max']
    assert abs(sp['ms_per_step_median'] - j['ms_per_step']) < 1e-9
    # the other BASELINE configs ride on the same line: C5's per-GPU load, C2 (CHOMP), C4 (GPMP2 at B = 2048)
    for k in ('c5', 'c2', 'c4', 'h128'):
        assert j[k]['value'] > 0 and j[k]['unit'] == 'iters/s' and 'workload' in j[k], k
    assert j['h128']['path'].startswith('persistent') and 'H=128' in j['h128']['workload']
    assert j['mppi']['value'] > 0 and j['mppi']['unit'] == 'problem-iters/s'
    assert r['kernel_ms'] <= j['ms_per_step']              # device-clock span of the timed launches themselves
    assert '4096 particles' in j['c5']['workload'] and 'B=1024' in j['c2']['workload'] and 'B=2048' in j['c4']['workload']
    assert j['c4']['roofline']['bound'] == 'mfma' and j['c4']['dtype'] == 'f64'
    # timings are reported, not asserted against a bar: a rare ~70 ms device stall on this pool would fail it
    assert r['kernel_ms'] > 0 and r['two_kernel_path_iters_per_sec'] > 0 and j['value'] > 0
    assert r['kernel_ms'] <= 1.25 * j['ms_per_step']        # the kernel's iteration cannot be slower than the step it is most of


def test_gpmp2_low_rank_form_equals_block_elimination_at_c4(gpu_device, monkeypatch):
    """Round 6: the low-rank form of the GPMP2 solve (mpb_gpmp2_lr.hip: A0 = priors + GP blocks + damping shared by all particles
    and factored once, one dense solve per particle of the size of its ACTIVE collision rows) against the block elimination of
    rounds 1-5 on the same linearisation, C4's shape and sigmas, 256 particles whose active sets run from none to ~100 rows, with
    and without the trust region: the steps agree to the solvers' fp64 rounding, far below the fp32 storage of x."""
    from motion_planning_baselines_amd import geometry as G, ops, workloads
    dev = gpu_device
    B, H, D = 256, 128, 7
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    geom = ops.DeviceGeometry(robot, field, dev)
    q = workloads.collision_free_configs(robot, field, 2 * B, 23, dev)
    dt = 5.0 / H
    x0 = workloads.straight_line_means(q[:B], q[B:], H, dt, False, dev)
    z = torch.zeros(B, D, device=dev)
    start = torch.cat([torch.from_numpy(q[:B]).to(dev), z], -1).contiguous()
    goal = torch.cat([torch.from_numpy(q[B:]).to(dev), z], -1).contiguous()
    sig = (1e-5, 1e-2, 1e-5, 1e-5)
    ws = ops.gpmp2_workspace(B, H, D, dev)
    rows = ops.gpmp2_collision_rows(x0, geom)[0]
    n_act = (rows[..., :D].abs().sum(-1) > 0).sum(1)
    assert int(n_act.max()) > 60 and int(n_act.min()) == 0, (int(n_act.min()), int(n_act.max()))
    for trust in (True, False):
        out = {}
        for form in ('lr', 'block'):
            monkeypatch.setenv('MPB_GPMP2_FORM', form)
            x, c = x0.clone(), torch.empty(B, device=dev)
            ops.gpmp2_step(x, start, goal, geom, ws, sig, dt, 1e-2, trust, 1.0, costs_out=c)
            torch.cuda.synchronize()
            out[form] = (x.double() - x0.double(), c)
        monkeypatch.delenv('MPB_GPMP2_FORM')
        d_lr, d_bl = out['lr'][0], out['block'][0]
        per_particle = (d_lr - d_bl).abs().amax(dim=(1, 2)) / d_bl.abs().amax(dim=(1, 2)).clamp_min(1e-12)
        # x is stored in fp32: two correct fp64 steps can round differently in the last place of x (6e-8 of |x| ~ 3)
        ulp = 2.0 ** -23 * float(x0.abs().max())
        assert float((d_lr - d_bl).abs().max()) <= 2.5 * ulp, (trust, float((d_lr - d_bl).abs().max()), ulp)
        print('trust %d: low-rank vs block elimination, step rel diff per particle: max %.2e median %.2e' %
              (trust, float(per_particle.max()), float(per_particle.median())))
        np.testing.assert_allclose(out['lr'][1].cpu().numpy(), out['block'][1].cpu().numpy(), rtol=1e-5)
