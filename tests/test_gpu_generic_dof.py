"""GPU: serial chains with D = 1, 4, 5, 6, 8, 9 and 12 (MPB_MAX_DOF since round 6: 8 until then; beyond 8 joints STOMP runs position only --
its rollout tile holds 16 channels -- and GPMP2 takes the low-rank form, whose chains have no block-size limit) degrees of freedom -- the dof counts the kernels have no
compile-time instance for (the reference's robots are D = 2, 3, 7): collision cost / gradient, STOMP (d = 2D up to 16 =
MPB_MAX_D, chunked MFMA kernel), CHOMP (general kernel) and GPMP2 (run-time block size; D = 8 fills the 16 x 16 tile
with no padding) against the oracle on small seeded inputs."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (a, d, alpha) rows in the modified-DH convention of geometry._mdh: a Panda-like arm with an eighth joint
_MDH = [(0.0, 0.333, 0.0), (0.0, 0.0, -math.pi / 2), (0.0, 0.316, math.pi / 2), (0.0825, 0.0, math.pi / 2),
        (-0.0825, 0.384, -math.pi / 2), (0.0, 0.0, math.pi / 2), (0.088, 0.0, math.pi / 2), (0.0, 0.12, -math.pi / 2),
        # round 6: four more joints (MPB_MAX_DOF = 12)
        (0.05, 0.0, math.pi / 2), (0.0, 0.15, -math.pi / 2), (0.04, 0.0, math.pi / 2), (0.0, 0.10, -math.pi / 2)]


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def make_arm(D):
    from motion_planning_baselines_amd import geometry as G
    tfs = np.stack([G._mdh(alpha, a, d) for (a, d, alpha) in _MDH[:D]])
    # two collision spheres per link frame 1..D and one on the tool frame D+1
    frames, offs, rad = [], [], []
    for f in range(1, D + 1):
        frames += [f, f]
        offs += [(0.0, 0.0, -0.05), (0.02, -0.03, 0.04)]
        rad += [0.07, 0.05]
    frames.append(D + 1)
    offs.append((0.0, 0.0, 0.05))
    rad.append(0.05)
    return G.RobotSerialChain(tfs, frames, offs, rad, q_min=[-2.5] * D, q_max=[2.5] * D)


def make_field(seed=4):
    from motion_planning_baselines_amd import geometry as G
    rng = np.random.RandomState(seed)
    sph = np.concatenate([rng.uniform(-0.7, 0.7, (14, 3)) + [0.0, 0.0, 0.4], rng.uniform(0.08, 0.18, (14, 1))], 1)
    return G.CollisionField(spheres=sph, margin=0.06)


def trajs(D, B, H, d, seed):
    g = torch.Generator().manual_seed(seed)
    a = -2.0 + 4.0 * torch.rand(B, 1, D, generator=g)
    b = -2.0 + 4.0 * torch.rand(B, 1, D, generator=g)
    t = torch.linspace(0, 1, H).reshape(1, H, 1)
    q = a * (1 - t) + b * t + 0.03 * torch.randn(B, H, D, generator=g)
    return (torch.cat([q, 0.3 * torch.randn(B, H, d - D, generator=g)], -1) if d > D else q).contiguous()


@pytest.mark.parametrize('D', [1, 4, 5, 6, 8, 9, 12])
def test_cost_and_grad(gpu_device, D):
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    robot, field = make_arm(D), make_field()
    rr, rf = make_ref_geometry(robot, field)
    x = trajs(D, 13, 64, 2 * D, 3).requires_grad_(True)
    ref = O.collision_cost(x, rr, rf, 0.5, weight=2.0)
    ref.sum().backward()
    assert float(ref.detach().max()) > 0, 'inputs must collide'
    geom = ops.DeviceGeometry(robot, field, gpu_device)
    out, grad = ops.cost_collision_grad(x.detach().to(gpu_device), geom, 4.0, weight=2.0)
    out2 = ops.cost_collision_eval(x.detach().to(gpu_device), geom, 4.0, weight=2.0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=3e-5, atol=1e-5)
    assert torch.equal(out, out2)
    diff = (grad.cpu() - x.grad).abs()
    tol = 2e-4 * x.grad.abs().max() + 2e-4 * x.grad.abs()
    assert float((diff > tol).float().mean()) < 3e-3


@pytest.mark.parametrize('D,pos_only', [(1, False), (4, False), (4, True), (5, False), (6, True), (8, False), (8, True), (9, True), (12, True)])
def test_stomp(gpu_device, D, pos_only):
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    robot, field = make_arm(D), make_field()
    rr, rf = make_ref_geometry(robot, field)
    P, S, H = 3, 6, 64
    d = D if pos_only else 2 * D
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, 0.05, 1.0, cpu)
    Sigma, L = torch.inverse(R).contiguous(), precision_to_scale_tril(R).contiguous()
    means0 = trajs(D, P, H, d, 5)
    eps = torch.randn(2, S, d, P, H, generator=torch.Generator().manual_seed(9))
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


@pytest.mark.parametrize('D', [1, 4, 5, 6, 8, 9, 12])
def test_gpmp2(gpu_device, D):
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    robot, field = make_arm(D), make_field()
    B, H, dt = 4, 12, 0.1
    x0 = trajs(D, B, H, 2 * D, 7)
    start, goal = x0[:, 0].clone(), x0[:, -1].clone()
    start[:, D:] = 0
    goal[:, D:] = 0
    sig = (1e-3, 0.5, 1e-3, 5e-2)
    geom = ops.DeviceGeometry(robot, field, dev)
    f64 = dict(device='cpu', dtype=torch.float64)
    rr, rf = make_ref_geometry(robot, field, f64)
    for trust in (False, True):
        x = x0.clone().to(dev)
        costs = torch.empty(B, device=dev)
        ws = ops.gpmp2_workspace(B, H, D, dev)
        ops.gpmp2_step(x, start.to(dev), goal.to(dev), geom, ws, sig, dt, 1e-2, trust, 0.5, costs_out=costs)
        torch.cuda.synchronize()
        ref = O.gpmp2_iteration(x0.double(), rr, rf, start.double(), goal.double(), D=D, dt=dt, sigma_start=sig[0],
                                sigma_gp=sig[1], sigma_goal=sig[2], sigma_coll=sig[3], delta=1e-2, trust_region=trust,
                                step_size=0.5, tensor_args=f64)
        assert float(ref['dtheta'].abs().max()) > 1e-3
        step_err = float((x.cpu().double() - x0.double() - 0.5 * ref['dtheta']).abs().max() / (0.5 * ref['dtheta']).abs().max())
        assert rel_err(x, ref['means'].float()) < 1e-5, (D, trust)
        assert step_err < 2e-3, (D, trust, step_err)
        np.testing.assert_allclose(costs.cpu().numpy(), ref['costs'].numpy(), rtol=2e-3)


@pytest.mark.parametrize('D', [1, 4, 6, 8, 9, 12])
def test_chomp(gpu_device, D):
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd.planners.chomp import CHOMP
    from motion_planning_baselines_amd.planners.costs.cost_functions import CostCollision, CostComposite
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    robot, field = make_arm(D), make_field()
    rr, rf = make_ref_geometry(robot, field)
    B, H, dt = 5, 64, 0.05
    ta = dict(device=dev, dtype=torch.float32)
    x0 = trajs(D, B, H, 2 * D, 11)
    cost = CostComposite(robot, H, [CostCollision(robot, H, field=field, sigma_coll=0.5, tensor_args=ta)], tensor_args=ta)
    pl = CHOMP(n_dof=D, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt, start_state=x0[0, 0, :D].to(dev),
               cost=cost, initial_particle_means=x0.to(dev), step_size=0.02, grad_clip=0.1, weight_prior_cost=1e-4,
               pos_only=False, tensor_args=ta)
    pl.optimize(opt_iters=3)
    torch.cuda.synchronize()
    ref = x0.clone()
    Rm = O.chomp_precision(H, dt, dict(device='cpu', dtype=torch.float32))
    for _ in range(3):
        ref = O.chomp_iteration(ref, Rm, lambda x: O.collision_cost(x, rr, rf, 0.5), 1e-4, 0.02, 0.1)['means']
    assert rel_err(pl._particle_means, ref) < 1e-4
