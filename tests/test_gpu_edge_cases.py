"""GPU: edge cases of the HIP path against the oracle on small seeded inputs -- ragged / odd sizes, generic-H
and generic-d kernels, obstacle sets that defeat the broad phase (crowded cells, > 63 spheres), boxes with a
chain robot, 3-D point robot, empty batches, argument validation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _geoms():
    from motion_planning_baselines_amd import geometry as G
    rng = np.random.RandomState(1)
    crowded = np.concatenate([rng.uniform(-0.15, 0.15, (12, 3)) + [0.4, 0.1, 0.4], rng.uniform(0.03, 0.06, (12, 1))], 1)
    many = np.concatenate([rng.uniform(-0.9, 0.9, (100, 3)), rng.uniform(0.02, 0.05, (100, 1))], 1)
    boxes3 = np.concatenate([rng.uniform(-0.6, 0.6, (5, 3)), rng.uniform(0.05, 0.15, (5, 3))], 1)
    return {
        'panda_crowded': (G.RobotPanda(), G.CollisionField(spheres=crowded, margin=0.05)),          # grid cells overflow
        'panda_many': (G.RobotPanda(), G.CollisionField(spheres=many, margin=0.03)),                # > 63 spheres: no grid
        'panda_boxes': (G.RobotPanda(), G.CollisionField(spheres=crowded[:3], boxes=boxes3, margin=0.05)),
        'panda_boxes_only': (G.RobotPanda(), G.CollisionField(boxes=boxes3, margin=0.05)),
        'point3d': (G.RobotPointMass(3, radius=0.05), G.env_spheres_3d(seed=2)),
        'point2d_boxes': (G.RobotPointMass(2, radius=0.02), G.env_dense_2d(seed=5)),
    }


def _trajs(robot, B, H, d, seed):
    g = torch.Generator().manual_seed(seed)
    D = robot.q_dim
    lo, hi = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
    a = lo + (hi - lo) * torch.rand(B, 1, D, generator=g)
    b = lo + (hi - lo) * torch.rand(B, 1, D, generator=g)
    t = torch.linspace(0, 1, H).reshape(1, H, 1)
    q = a * (1 - t) + b * t + 0.03 * torch.randn(B, H, D, generator=g)
    return (torch.cat([q, torch.randn(B, H, d - D, generator=g)], -1) if d > D else q).contiguous()


@pytest.mark.parametrize('name', ['panda_crowded', 'panda_many', 'panda_boxes', 'panda_boxes_only', 'point3d', 'point2d_boxes'])
@pytest.mark.parametrize('H', [64, 37])
def test_cost_and_grad_vs_oracle(gpu_device, name, H):
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    robot, field = _geoms()[name]
    rr, rf = make_ref_geometry(robot, field)
    D = robot.q_dim
    x = _trajs(robot, 21, H, 2 * D, 3).requires_grad_(True)
    ref = O.collision_cost(x, rr, rf, 0.5, weight=2.0)
    ref.sum().backward()
    geom = ops.DeviceGeometry(robot, field, gpu_device)
    out = ops.cost_collision_eval(x.detach().to(gpu_device), geom, 4.0, weight=2.0)
    out2, grad = ops.cost_collision_grad(x.detach().to(gpu_device), geom, 4.0, weight=2.0)
    torch.cuda.synchronize()
    assert float(ref.detach().max()) > 0, 'inputs must collide'
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=3e-5, atol=1e-5)
    assert torch.equal(out, out2)              # broad-phase path == exhaustive path, bit for bit
    diff = (grad.cpu() - x.grad).abs()
    tol = 2e-4 * x.grad.abs().max() + 2e-4 * x.grad.abs()
    assert float((diff > tol).float().mean()) < 3e-3


@pytest.mark.parametrize('name,P,S,H,pos_only', [
    ('panda_crowded', 3, 5, 64, True),       # ragged: P*S not a multiple of the 4 rollouts per block, fast path d=7
    ('panda_boxes', 2, 8, 64, False),        # d = 14
    ('point3d', 5, 3, 64, False),            # d = 6
    ('point3d', 4, 7, 64, True),             # d = 3
    ('panda_many', 2, 4, 80, True),          # chunked kernel: two chunks, ragged tail, odd d (scalar stores: 80*7 % 4 = 0 -> vector)
    ('point2d_boxes', 3, 9, 100, False),     # chunked kernel, H > 64, d = 4
    ('panda_crowded', 2, 5, 48, False),      # chunked kernel, one padded chunk (H < 64), d = 14
    ('panda_boxes', 2, 3, 128, False),       # two full chunks
    ('point3d', 3, 4, 150, True),            # three chunks of the M = 4 instance, d = 3, 150*3 % 4 != 0 -> scalar stores
    ('point2d_boxes', 2, 4, 256, True),      # maximum horizon, d = 2
    ('point3d', 2, 6, 61, True),             # odd horizon below one chunk, 61*3 % 4 != 0
])
def test_stomp_iteration_vs_oracle(gpu_device, name, P, S, H, pos_only):
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    robot, field = _geoms()[name]
    rr, rf = make_ref_geometry(robot, field)
    D = robot.q_dim
    d = D if pos_only else 2 * D
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, 0.05, 1.0, cpu)
    Sigma, L = torch.inverse(R).contiguous(), precision_to_scale_tril(R).contiguous()
    means0 = _trajs(robot, P, H, d, 5)
    g = torch.Generator().manual_seed(9)
    eps = torch.randn(2, S, d, P, H, generator=g)
    sigma = 0.3
    ref = means0.clone()
    for it in range(2):
        out = O.stomp_iteration(ref, eps[it], L, Sigma, lambda x: O.collision_cost(x, rr, rf, sigma), 0.2, 0.7)
        ref = out['means']
    geom = ops.DeviceGeometry(robot, field, dev)
    means = means0.clone().to(dev)
    samples = torch.empty(P, S, H, d, device=dev)
    costs = torch.empty(P, S, device=dev)
    weights = torch.empty(P, S, device=dev)
    ops.stomp_step(means, eps.to(dev), samples, costs, weights, L.to(dev), Sigma.to(dev), geom, S, D, 1.0 / sigma ** 2, 1.0,
                   0.2, 0.7, n_iters=2)
    torch.cuda.synchronize()
    assert rel_err(samples, out['samples']) < 1e-4
    np.testing.assert_allclose(costs.cpu().numpy(), out['costs'].numpy(), rtol=1e-4, atol=1e-4)
    assert rel_err(means, ref) < 1e-4


def test_empty_and_invalid_arguments(gpu_device):
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd._lib import MPBError
    dev = gpu_device
    robot, field = _geoms()['point3d']
    geom = ops.DeviceGeometry(robot, field, dev)
    # empty batches are accepted and do nothing
    out = ops.cost_collision_eval(torch.empty(0, 64, 3, device=dev), geom, 1.0)
    assert out.shape == (0,)
    H, d, S = 64, 3, 4
    L = torch.eye(H, device=dev)
    m = torch.empty(0, H, d, device=dev)
    ops.stomp_step(m, None, torch.empty(0, S, H, d, device=dev), torch.empty(0, S, device=dev), torch.empty(0, S, device=dev),
                   L, L, geom, S, 3, 1.0, 1.0, 0.1, 1.0)
    # wrong device / dtype / shape / contiguity are rejected on the host before any launch
    x = torch.zeros(2, H, d, device=dev)
    with pytest.raises(ValueError):
        ops.cost_collision_eval(x.cpu(), geom, 1.0)
    with pytest.raises(ValueError):
        ops.cost_collision_eval(x.double(), geom, 1.0)
    with pytest.raises(ValueError):
        ops.cost_collision_eval(x.transpose(0, 1), geom, 1.0)
    with pytest.raises(ValueError):
        ops.stomp_step(x, None, torch.empty(2, S, H, d, device=dev), torch.empty(2, S + 1, device=dev),
                       torch.empty(2, S, device=dev), L, L, geom, S, 3, 1.0, 1.0, 0.1, 1.0)
    # the C-ABI itself rejects out-of-range shapes and bad temperatures
    with pytest.raises(MPBError):
        ops.stomp_step(x, None, torch.empty(2, S, H, d, device=dev), torch.empty(2, S, device=dev),
                       torch.empty(2, S, device=dev), L, L, geom, S, 3, 1.0, 1.0, 0.1, 0.0)
    big = torch.zeros(1, 300, d, device=dev)
    with pytest.raises(MPBError):
        ops.stomp_sample(big, None, torch.empty(1, S, 300, d, device=dev), torch.eye(300, device=dev), S)
    torch.cuda.synchronize()


@pytest.mark.parametrize('NP,S,Tn,c,with_geom', [
    (3, 5, 48, 2, True),       # fewer samples than waves, horizon shorter than a wave
    (2, 33, 64, 2, True),      # samples not a multiple of the 16 waves
    (2, 100, 100, 3, False),   # ragged horizon across two 64-lane chunks, c = 3
    (1, 20, 200, 2, False),    # scale_tril does not fit in LDS: global-memory path, four chunks
    (4, 16, 64, 1, False),     # one control dimension
    (2, 24, 96, 4, False),     # maximum control dimension
    (2, 1, 64, 2, True),       # ONE sample: a one-wave workgroup (save-best and the softmax share wave 0)
    (3, 2, 64, 2, True),       # two samples: the smallest workgroup with a softmax wave of its own
])
def test_mppi_shapes_vs_oracle(gpu_device, NP, S, Tn, c, with_geom):
    """MPPI kernel (wave = sample, lane = time step) on ragged shapes against the oracle's sequential rollout.  Every case
    tracks the best sample (mppi.py:164-168) as MPPI.optimize does."""
    from motion_planning_baselines_amd import geometry as G, ops
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    gen = torch.Generator().manual_seed(NP * 1000 + S + Tn + c)
    ta = dict(device='cpu', dtype=torch.float32)
    dt, temp, step = 0.05, 0.7, 0.8
    Cov = O.mppi_covariance([0.2 + 0.05 * i for i in range(c)], Tn, c, 'const_ctrl', ta)          # (T,T,c)
    Cov = Cov + 0.05 * torch.eye(Tn).unsqueeze(-1)
    tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous()
    cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous()
    mean0 = 0.3 * torch.randn(NP, Tn, c, generator=gen)
    state0 = 0.5 * torch.randn(NP, c, generator=gen)
    goal = 0.5 * torch.randn(NP, c, generator=gen)
    cmin, cmax = torch.full((c,), -0.6), torch.full((c,), 0.5)                                     # clamping is active
    disc = 0.99 ** torch.arange(Tn, dtype=torch.float32)
    cw = dict(pos=1.0, vel=0.5, ctrl=0.1, pos_T=20.0)
    n_it = 2
    eps = torch.randn(n_it, NP, c, S, Tn, generator=gen)
    geom = rr = rf = None
    if with_geom:
        robot, field = G.RobotPointMass(2, radius=0.02), G.env_dense_2d()
        geom = ops.DeviceGeometry(robot, field, dev)
        rr, rf = make_ref_geometry(robot, field, ta)
    f = lambda t: t.contiguous().to(dev)
    mean = f(mean0.clone())
    controls, states = torch.empty(NP, S, Tn, c, device=dev), torch.empty(NP, S, Tn, c, device=dev)
    costs, weights = torch.empty(NP, S, device=dev), torch.empty(NP, S, device=dev)
    best_cost, best_states = torch.full((NP,), 3.0e38, device=dev), torch.zeros(NP, Tn, c, device=dev)
    ops.mppi_step(mean, f(eps), f(tril), f(cinv), f(state0), f(goal), f(cmin), f(cmax), f(disc),
                  f(torch.tensor([cw['pos'], cw['vel'], cw['ctrl'], cw['pos_T']])), geom, controls, states, costs, weights,
                  dt, k_sigma=4.0, weight=1.5, temp=temp, step_size=step, n_iters=n_it, best_cost=best_cost, best_states=best_states)
    torch.cuda.synchronize()
    for p in range(NP):
        m = mean0[p].clone()
        best = (np.inf, None)
        for it in range(n_it):
            shift = 0.0
            if with_geom:      # quirk Q6: the summed collision cost of ALL samples shifts every sample's cost
                pre = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c)
                q = pre['states'][:, 1:, :2]
                shift = 1.5 * 4.0 * float(rf.compute_cost(q, rr.fk_map_collision(q)).sum())
            out = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c,
                                   shift_cost=shift)
            m = out['mean']
            cst = out['costs'].reshape(-1)
            if float(cst.min()) < best[0]:
                best = (float(cst.min()), out['states'][int(cst.argmin())])
        np.testing.assert_allclose(float(best_cost[p]), best[0], rtol=2e-4)
        if S <= 2:           # (with more samples a near-tie may pick another winner in fp32: the cost above is the bar there)
            assert rel_err(best_states[p], best[1]) < 1e-4
        assert rel_err(controls[p], out['controls']) < 1e-4
        assert rel_err(states[p], out['states']) < 1e-4
        np.testing.assert_allclose(costs[p].cpu().numpy(), out['costs'].reshape(-1).numpy(), rtol=2e-4)
        np.testing.assert_allclose(weights[p].cpu().numpy(), out['weights'].reshape(-1).numpy(), rtol=5e-2, atol=1e-5)
        assert rel_err(mean[p], m) < 2e-4


def test_new_entry_points_empty_and_invalid(gpu_device):
    """Empty batches are accepted, wrong shapes / devices rejected on the host, for the round-1 late additions."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd._lib import MPBError
    dev = gpu_device
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    geom = ops.DeviceGeometry(robot, field, dev, keep_all_links=True)
    with pytest.raises(ValueError):
        ops.fk_collision_points(torch.zeros(1, 8, 7, device=dev), ops.DeviceGeometry(robot, field, dev))   # pruned link table
    assert ops.fk_collision_points(torch.empty(0, 8, 7, device=dev), geom).shape == (0, 8, 31, 3)
    assert ops.field_cost_points(torch.empty(0, 8, 31, 3, device=dev), geom).shape == (0, 8)
    assert ops.gp_factor_error(torch.empty(0, 8, 14, device=dev), 7, 0.1).shape == (0, 7, 14)
    assert ops.traj_resample(torch.empty(0, 5, 7, device=dev), torch.empty(0, dtype=torch.int32, device=dev), 16, 0.1).shape == (0, 16, 14)
    with pytest.raises(ValueError):
        ops.field_cost_points(torch.zeros(2, 8, 30, 3, device=dev), geom)            # wrong number of collision spheres
    with pytest.raises(ValueError):
        ops.fk_collision_points(torch.zeros(2, 8, 7), geom)                           # CPU tensor
    with pytest.raises(ValueError):
        ops.gp_factor_error(torch.zeros(2, 8, 13, device=dev), 7, 0.1)                # d != 2D
    with pytest.raises(MPBError):
        ops.traj_resample(torch.zeros(2, 5, 7, device=dev), torch.full((2,), 5, dtype=torch.int32, device=dev), 1, 0.1)   # H < 2
    from motion_planning_baselines_amd import _lib
    import ctypes
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    out = torch.empty(2, 200, 4, device=dev)
    rc = _lib.lib().mpb_gp_prior_sample_dense(p(out), p(torch.zeros(1, 200, 4, dtype=torch.float64, device=dev)), ctypes.c_void_p(0),
                                              p(torch.eye(400, dtype=torch.float64, device=dev)), 1, 2, 200, 2, 0, ctypes.c_void_p(0))
    assert rc == 2 and b'H > 128' in _lib.lib().mpb_last_error()          # MPB_E_UNSUPPORTED, nothing launched
    torch.cuda.synchronize()


_MPPI_WAVES_CHILD = r"""
import sys, numpy as np, torch
from motion_planning_baselines_amd import geometry as G, ops
from motion_planning_baselines_amd.planners.priors.gaussian import const_ctrl_Cov
out = sys.argv[1]
dev = torch.device('cuda:0')
NP, S, T, c = 3, 40, 64, 2
f = lambda a: torch.as_tensor(a, dtype=torch.float32).contiguous().to(dev)
Cov = const_ctrl_Cov([0.3, 0.3], T, c, dict(device='cpu', dtype=torch.float32))
tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous().to(dev)
gen = torch.Generator().manual_seed(5)
state0 = f(torch.rand(NP, c, generator=gen) * 0.2 - 0.9)
goal = f(torch.rand(NP, c, generator=gen) * 0.2 + 0.7)
geom = ops.DeviceGeometry(G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d(), dev)
mean = torch.zeros(NP, T, c, device=dev)
controls, states = torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, T, c, device=dev)
costs, weights = torch.empty(NP, S, device=dev), torch.empty(NP, S, device=dev)
best_cost, best_states = torch.full((NP,), 3.0e38, device=dev), torch.zeros(NP, T, c, device=dev)
ops.mppi_step(mean, None, tril, cinv, state0, goal, f([-1., -1.]), f([1., 1.]), torch.ones(T, device=dev),
              f([1., 1., 1., 100.]), geom, controls, states, costs, weights, 0.04, k_sigma=1e3, weight=1.0, temp=1.0,
              step_size=0.7, n_iters=3, seed=3, best_cost=best_cost, best_states=best_states)
torch.cuda.synchronize()
np.savez(out, mean=mean.cpu().numpy(), controls=controls.cpu().numpy(), states=states.cpu().numpy(),
         costs=costs.cpu().numpy(), weights=weights.cpu().numpy(), best_cost=best_cost.cpu().numpy(),
         best_states=best_states.cpu().numpy())
"""


def test_mppi_same_bits_on_8_and_16_waves(gpu_device, tmp_path):
    """mpb_mppi_step gives a problem 16 waves, or 8 when there are at least two problems per CU (two workgroups resident
    per CU).  The choice must not show in the results: every per-sample quantity is computed by one wave, and the one
    cross-sample sum (quirk Q6's collision scalar) is taken in an order that depends on S alone.  MPB_MPPI_WAVES forces the
    workgroup size (read once per process: one child each)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = {}
    for nw in (16, 8, 5, 1):          # 1: a one-wave workgroup runs save-best and the softmax on the same wave
        out = str(tmp_path / f'nw{nw}.npz')
        env = dict(os.environ, MPB_MPPI_WAVES=str(nw), PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
        r = subprocess.run([sys.executable, '-c', _MPPI_WAVES_CHILD, out], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[nw] = np.load(out)
    for k in ('controls', 'states', 'costs', 'weights', 'mean', 'best_cost', 'best_states'):
        assert res[16][k].tobytes() == res[8][k].tobytes() == res[5][k].tobytes() == res[1][k].tobytes(), k
    assert float(res[16]['best_cost'].max()) < 1e30
    assert np.isfinite(res[16]['mean']).all() and float(np.abs(res[16]['mean']).max()) > 0


def test_mppi_grid_collision_equals_exhaustive(gpu_device, tmp_path):
    """With ONE grid-backed collision field the MPPI kernel looks a waypoint's candidates up in the broad-phase grid (staged
    in LDS) instead of walking every obstacle (round 4: 68 -> 46 us per iteration at 1 024 problems).  The grid only culls:
    same bits.  MPB_MPPI_NO_GRID=1 keeps the exhaustive loop (read once per process: one child each)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = {}
    for no_grid in (0, 1):
        out = str(tmp_path / f'g{no_grid}.npz')
        env = dict(os.environ, MPB_MPPI_NO_GRID=str(no_grid), PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
        r = subprocess.run([sys.executable, '-c', _MPPI_WAVES_CHILD, out], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[no_grid] = np.load(out)
    for k in ('controls', 'states', 'costs', 'weights', 'mean'):
        assert res[0][k].tobytes() == res[1][k].tobytes(), k
    assert float(res[0]['costs'].min()) > 0


def test_point_entry_points_edge_cases(gpu_device):
    """mpb_point_dynamics / mpb_point_traj_cost (round 4): empty batches are no-ops, one-step horizons and three-dimensional
    states work, wrong devices / shapes / strides are refused with an exception (never a silent CPU path)."""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    f = lambda a: torch.as_tensor(a, dtype=torch.float32).contiguous().to(dev)
    lo, hi = f([-1., -1., -1.]), f([1., 1., 1.])
    assert ops.point_dynamics(torch.empty(0, 1, 3, device=dev), torch.empty(0, 1, 3, device=dev), lo, hi, 0.1).shape == (0, 1, 3)
    x = f(np.arange(12).reshape(2, 2, 3) * 0.1)
    u = f(np.full((2, 2, 3), 2.0))
    np.testing.assert_allclose(ops.point_dynamics(x, u, lo, hi, 0.5).cpu().numpy(), x.cpu().numpy() + 0.5, rtol=1e-6)
    with pytest.raises(ValueError):
        ops.point_dynamics(x.cpu(), u, lo, hi, 0.5)
    with pytest.raises(ValueError):
        ops.point_dynamics(x.transpose(0, 1), u.transpose(0, 1), lo, hi, 0.5)          # not contiguous
    # one step, three state dimensions, a two-dimensional control
    X, U = f(np.ones((1, 4, 3))), f(np.full((1, 4, 2), 2.0))
    c = ops.point_traj_cost(X, U, f([0., 0., 0.]), f([0.5]), 2.0, 7.0, 3.0, 10.0, energy=1.5)
    want = 0.5 * (2.0 * 3.0 + 3.0 * 8.0) + 10.0 * 3.0 * 0.5 + 1.5
    np.testing.assert_allclose(c.cpu().numpy(), np.full(4, want, np.float32), rtol=1e-6)
    assert ops.point_traj_cost(torch.empty(3, 0, 2, device=dev), torch.empty(3, 0, 2, device=dev), f([0., 0.]), f([1., 1., 1.]),
                               1., 1., 1., 1.).shape == (0,)
    with pytest.raises(ValueError):
        ops.point_traj_cost(X, U, f([0., 0.]), f([0.5]), 1., 1., 1., 1.)                # goal of the wrong length


def _rewrite_grid_words(host, fn):
    """A copy of a packed geometry buffer with every (non-overflow) word of its broad-phase grid rewritten by fn(slots, n_sph)
    -> slots (lists of obstacle indices; pack_geometry header: [6] n_spheres, [16] off_grid, [26] n_cells)."""
    out = host.copy()
    w = out.view(np.uint32)
    ns, off, n = int(w[6]), int(w[16]), int(w[26])
    for i in range(off, off + n):
        word = int(w[i])
        if word == 0xFFFFFFFE:
            continue
        slots = [(word >> (8 * k)) & 0xFF for k in range(4)]
        slots = fn([b for b in slots if b < ns], ns)
        slots = (list(slots) + [ns] * 4)[:4]
        w[i] = slots[0] | (slots[1] << 8) | (slots[2] << 16) | (slots[3] << 24)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('pos_only', [False, True])
def test_grid_words_in_any_order_and_with_extra_candidates(gpu_device, pos_only):
    """The kernels that stage the grid as offset words (round 4: csrc/mpb_geom.h grid_offset_word -- the persistent STOMP
    kernels, MPPI) re-encode every cell while they copy it.  The encoding must not depend on what the host happened to
    produce: candidates listed in DESCENDING order (obstacle 0 then sits in a later slot, whose offset 0 means "unused" in
    the new form), and cells padded with obstacles that are no candidates at all (harmless by construction -- and a cell with
    four entries takes the exhaustive path) give the same bits as the host's own grid, on the persistent and on the
    two-kernel path."""
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    dev = gpu_device
    P, S, H = 5, 32, 64
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=pos_only)
    d = wl['means0'].shape[-1]
    R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, dict(device='cpu', dtype=torch.float32))
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    base = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    n_used = []

    def descending(slots, ns):
        n_used.append(len(slots))
        return sorted(slots, reverse=True)

    def padded(slots, ns):
        extra = [o for o in range(ns) if o not in slots]
        return list(slots) + extra[:max(0, (len(slots) + 1 + len(slots) % 2) - len(slots))]      # one or two non-candidates more

    variants = [base, ops.DeviceGeometry.from_packed(_rewrite_grid_words(base.host, descending), dev),
                ops.DeviceGeometry.from_packed(_rewrite_grid_words(base.host, padded), dev)]
    assert max(n_used) >= 2                                      # (the scene has cells with several candidates)
    res = []
    for geom in variants:
        for ws in (ops.stomp_workspace(P, S, H, d, dev), None):  # persistent / two-kernel
            m = wl['means0'].clone()
            s, c, w = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
            ops.stomp_run(m, None, s, c, w, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1.0, ws, n_iters=3, seed=11)
            torch.cuda.synchronize()
            if ws is not None:
                assert not ops.stomp_run_timed_out(ws)
            res.append((ws is not None, m, s, c))
    for persistent in (True, False):
        ref = next(r for r in res if r[0] == persistent)
        for r in res:
            if r[0] == persistent:
                assert torch.equal(r[3], ref[3]) and torch.equal(r[2], ref[2]) and torch.equal(r[1], ref[1])
    assert float(res[0][3].max()) > 0
