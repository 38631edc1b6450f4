"""GPU: the trajectory-only cost classes (CostGP, CostGPTrajectory, position-only wrapper,
CostSmoothnessCHOMP, CostJointLimits, CostGoalPrior, CostGoal) and CostComposite through
mpb_cost_terms_eval, against goldens produced by the reference's own classes
(tests/golden/make_goldens.py gen_cost_terms) and against the oracle on other inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, product_geometry_from_golden, ref_geometry_from_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RTOL = 1e-4          # north_star tolerance (fp32, relative)


def _robot(name, dt):
    from motion_planning_baselines_amd import geometry as G
    r = G.RobotPointMass(2, radius=0.01) if 'pm2d' in name else G.RobotPanda()
    r.dt = dt
    return r


def _bar(g, key):
    """max(1e-4, 2 x the reference's own fp32-vs-fp64 spread) for a golden quantity."""
    a, b = g[key + '_f32'].astype(np.float64), g[key + '_f64']
    env = float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
    return max(RTOL, 2 * env)


def _close(got, want, bar):
    got = got.detach().cpu().double().numpy()
    err = float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300))
    assert err < bar, (err, bar)


@pytest.mark.parametrize('name', ['cost_terms_pm2d', 'cost_terms_panda'])
def test_cost_classes_vs_reference_golden(gpu_device, name):
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    g = load_golden(name)
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    robot = _robot(name, dt)
    assert np.array_equal(robot.q_min_np, g['q_min']) and np.array_equal(robot.q_max_np, g['q_max'])
    x = T(g['trajs']).to(dev)
    sig = dict(sigma_start=float(g['sigma_start']), sigma_gp=float(g['sigma_gp']))
    gp = C.CostGP(robot, H, T(g['start']).to(dev), dt, sig, tensor_args=ta)
    _close(gp(x), g['gp_f64'], _bar(g, 'gp'))
    _close(C.CostGPTrajectory(robot, H, dt, sigma_gp=sig['sigma_gp'], tensor_args=ta)(x), g['gptraj_f64'], _bar(g, 'gptraj'))
    _close(C.CostGPTrajectoryPositionOnlyWrapper(robot, H, dt, sigma_gp=sig['sigma_gp'], tensor_args=ta)(
        x[..., :D].contiguous()), g['gptraj_posonly_f64'], _bar(g, 'gptraj_posonly'))
    # the reference's stub returns per-column values (B, d); the build's cost is their sum
    _close(C.CostSmoothnessCHOMP(robot, H, tensor_args=ta)(x), g['smooth_f64'].sum(-1), RTOL)
    jl = C.CostJointLimits(robot, H, eps=float(g['jl_eps']), tensor_args=ta)(x)
    assert jl.ndim == 0 and jl.dtype == torch.float32                  # batch-global scalar, as in the reference
    _close(jl, g['jlim_f64'], _bar(g, 'jlim'))
    gpr = C.CostGoalPrior(robot, H, multi_goal_states=T(g['goals']).to(dev), num_particles_per_goal=int(g['npg']),
                          num_samples=int(g['S']), sigma_goal_prior=float(g['sigma_goal_prior']), tensor_args=ta)
    _close(gpr(x), g['goalprior_f64'], _bar(g, 'goalprior'))
    # 4-D input (N, B, H, d) is flattened like get_q_pos_vel_and_fk_map does (cost_functions.py:42-48)
    x4 = x.reshape(2, -1, H, 2 * D)
    _close(C.CostGPTrajectory(robot, H, dt, sigma_gp=sig['sigma_gp'], tensor_args=ta)(x4), g['gptraj_f64'], _bar(g, 'gptraj'))
    # dense linear systems of the GP / goal factors reproduce the costs: b^T K b == cost
    A, b, K = gp.get_linear_system(x)
    _close((b.transpose(1, 2) @ K @ b).reshape(-1), g['gp_f64'], 1e-3)
    A, b, K = gpr.get_linear_system(x[:int(g['G']) * int(g['npg'])])
    assert A.shape == (int(g['G']) * int(g['npg']), 2 * D, 2 * D * H)


@pytest.mark.parametrize('name', ['cost_terms_pm2d', 'cost_terms_panda'])
def test_composite_one_pass_equals_sum_of_members(gpu_device, name):
    """CostComposite.eval (cost_functions.py:70-87): weighted sum; all trajectory terms in one launch, the
    joint-limit scalar broadcast to every trajectory, collision on the interpolated trajectories if given."""
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    g = load_golden(name)
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    robot = _robot(name, dt)
    field = G.env_grid_circles_2d() if D == 2 else G.env_spheres_3d()
    x = T(g['trajs']).to(dev)
    members = [
        C.CostCollision(robot, H, field=field, sigma_coll=0.1, tensor_args=ta),
        C.CostGPTrajectory(robot, H, dt, sigma_gp=float(g['sigma_gp']), tensor_args=ta),
        C.CostJointLimits(robot, H, eps=float(g['jl_eps']), tensor_args=ta),
        C.CostGoalPrior(robot, H, multi_goal_states=T(g['goals']).to(dev), num_particles_per_goal=int(g['npg']),
                        num_samples=int(g['S']), sigma_goal_prior=float(g['sigma_goal_prior']), tensor_args=ta),
        C.CostSmoothnessCHOMP(robot, H, tensor_args=ta),
    ]
    w = [1.0, 0.5, 3.0, 2.0, 1e-3]
    comp = C.CostComposite(robot, H, members, weights_cost_l=w, tensor_args=ta)
    coll, groups, other = comp.device_plan(dev)
    assert len(coll) == 1 and len(groups) == 1 and not other           # four trajectory terms, one launch
    total = comp(x)
    # oracle: the same sum in fp64
    ta64 = dict(device='cpu', dtype=torch.float64)
    rr, rf = make_ref_geometry(robot, field, ta64)
    x64 = T(g['trajs']).to(**ta64)
    want = (w[0] * O.collision_cost(x64, rr, rf, 0.1)
            + w[1] * O.cost_gp_trajectory_eval(x64, D, dt, float(g['sigma_gp']), ta64)
            + w[2] * O.cost_joint_limits_eval(x64, D, T(g['q_min']).double(), T(g['q_max']).double(), float(g['jl_eps']))
            + w[3] * O.cost_goal_prior_multi_eval(x64, T(g['goals']).double(), int(g['npg']) * int(g['S']),
                                                  float(g['sigma_goal_prior']))
            + w[4] * O.cost_smoothness_chomp_eval(x64, dt, ta64)[0])
    _close(total, want.numpy(), RTOL)
    # member list form
    costs, weights = comp(x, return_invidual_costs_and_weights=True)
    assert len(costs) == 5 and weights == w
    acc = sum(wi * ci for wi, ci in zip(w, costs))
    _close(acc, want.numpy(), RTOL)
    # two members of the same kind cannot share a launch: they are accumulated by a second one
    comp2 = C.CostComposite(robot, H, [members[1], members[1]], weights_cost_l=[1.0, 2.0], tensor_args=ta)
    assert len(comp2.device_plan(dev)[1]) == 2
    _close(comp2(x), 3.0 * O.cost_gp_trajectory_eval(x64, D, dt, float(g['sigma_gp']), ta64).numpy(), RTOL)
    # interpolated trajectories feed the collision member only (cost_functions.py:77-84)
    xi = torch.cat([x, x[:, -1:]], 1).contiguous()                      # (B, H+1, d) stand-in
    tot_i = comp(x, trajs_interpolated=xi)
    want_i = want - w[0] * O.collision_cost(x64, rr, rf, 0.1) + w[0] * O.collision_cost(
        torch.cat([x64, x64[:, -1:]], 1), rr, rf, 0.1)
    _close(tot_i, want_i.numpy(), RTOL)


def test_cost_goal_last_waypoint(gpu_device):
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_dense_2d()
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(32, 16, 4, generator=gen) * 2 - 1).to(dev)
    cg = C.CostGoal(robot, 16, field=field, sigma_goal=0.5, tensor_args=ta)
    rr, rf = make_ref_geometry(robot, field, dict(device='cpu', dtype=torch.float64))
    q = x[:, -1:, :2].cpu().double()
    want = rf.compute_cost(q, rr.fk_map_collision(q)).reshape(32, 1).sum(1) / 0.5 ** 2
    _close(cg(x), want.numpy(), RTOL)
    assert C.CostGoal(robot, 16, field=None, sigma_goal=0.5, tensor_args=ta)(x) == 0
    A, b, K = cg.get_linear_system(x)
    assert A.shape == (32, 1, 64) and b.shape == (32, 1, 1) and K.shape == (32, 1, 1)


def test_stomp_with_gp_and_collision_composite(gpu_device):
    """STOMP on CostComposite([CostCollision, CostGPTrajectory]) runs sample -> terms -> update on the device
    and follows the oracle's iteration with the same injected noise."""
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    from motion_planning_baselines_amd.planners.stomp import STOMP
    from oracle import planners_ref as O
    g = load_golden('stomp_pm2d_benign')
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    robot, field = product_geometry_from_golden(g)
    rr, rf = ref_geometry_from_golden(g, dtype=torch.float64)
    P, S, H, D, dt = int(g['P']), int(g['S']), int(g['H']), int(g['D']), float(g['dt'])
    sigma_coll, sigma_gp, w_gp = float(g['sigma_coll']), 2.0, 0.05
    comp = C.CostComposite(robot, H, [C.CostCollision(robot, H, field=field, sigma_coll=sigma_coll, tensor_args=ta),
                                      C.CostGPTrajectory(robot, H, dt, sigma_gp=sigma_gp, tensor_args=ta)],
                           weights_cost_l=[1.0, w_gp], tensor_args=ta)
    assert C.device_plan(comp, dev) is not None
    pl = STOMP(n_dof=D, n_support_points=H, num_particles_per_goal=P, num_samples=S, opt_iters=1, dt=dt,
               start_state=T(g['start']).to(dev), cost=comp, initial_particle_means=T(g['means0']).to(dev),
               temperature=float(g['temperature']), step_size=float(g['lr']), sigma_spectral=float(g['sigma_spectral']),
               pos_only=bool(g['pos_only']), tensor_args=ta, noise='torch_cpu')
    pl.Sigma = T(g['Sigma']).to(dev).contiguous()
    pl.scale_tril = T(g['L']).to(dev).contiguous()
    ta64 = dict(device='cpu', dtype=torch.float64)
    L64, Sig64 = T(g['L']).double(), T(g['Sigma']).double()
    means = T(g['means0']).double()
    cost_fn = lambda xx: (O.collision_cost(xx, rr, rf, sigma_coll)
                          + w_gp * O.cost_gp_trajectory_eval(xx, D, dt, sigma_gp, ta64))
    torch.manual_seed(11)
    eps_all = [torch.empty(S, 2 * D, P, H).normal_() for _ in range(4)]
    torch.manual_seed(11)
    for it in range(4):
        pl.optimize()
        out = O.stomp_iteration(means, eps_all[it].double(), L64, Sig64, cost_fn, float(g['lr']), float(g['temperature']))
        means = out['means']
        _close(pl.costs.reshape(-1), out['costs'].reshape(-1).numpy(), 1e-3 if it else RTOL)
    _close(pl._particle_means, means.numpy(), 1e-3)


def test_cost_terms_error_behaviour(gpu_device):
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd._lib import MPBError
    dev = gpu_device
    x = torch.zeros(4, 16, 4, device=dev)
    with pytest.raises(MPBError):
        ops.cost_terms_eval(x, 3, dt=0.1, k_gp=1.0, terms={'gp'})               # d != 2 * n_dof
    with pytest.raises(MPBError):
        ops.cost_terms_eval(x, 2, dt=0.0, k_gp=1.0, terms={'gp'})               # dt
    with pytest.raises(ValueError):
        ops.cost_terms_eval(x, 2, dt=0.1, terms={'nope'})
    with pytest.raises(ValueError):
        ops.cost_terms_eval(x.cpu(), 2, dt=0.1, k_gp=1.0, terms={'gp'})         # no CPU path
    with pytest.raises(ValueError):
        ops.cost_terms_eval(x, 2, k_goal=1.0, goal_states=torch.zeros(1, 4, device=dev), trajs_per_goal=2, terms={'goal'})
    out, _ = ops.cost_terms_eval(torch.zeros(0, 16, 4, device=dev), 2, dt=0.1, k_gp=1.0, terms={'gp'})
    assert out.shape == (0,)
    # long horizon (several waypoints per lane) against the oracle
    from oracle import planners_ref as O
    gen = torch.Generator().manual_seed(3)
    xl = torch.randn(5, 200, 6, generator=gen)
    want = O.cost_gp_trajectory_eval(xl.double(), 3, 0.05, 1.5, dict(device='cpu', dtype=torch.float64))
    got, _ = ops.cost_terms_eval(xl.to(dev), 3, dt=0.05, k_gp=1 / 1.5 ** 2, terms={'gp'})
    _close(got, want.numpy(), RTOL)


def test_traj_interpolate_and_finite_difference(gpu_device):
    """The two trajectory utilities either side of the loop (both build-defined: torch_robotics is external)."""
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    dev = gpu_device
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(7, 33, 5, generator=gen)
    for n in (0, 1, 3, 8):
        got = ops.traj_interpolate(x.to(dev), n).cpu()
        want = O.interpolate_trajs(x, n) if n else x
        assert got.shape == ((7, 32 * (n + 1) + 1, 5))
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
        assert torch.equal(got[:, ::n + 1], x)                           # support points are kept bit for bit
    pos = torch.randn(9, 70, 7, generator=gen)
    got = ops.traj_finite_difference(pos.to(dev), 0.05).cpu()
    want = torch.cat((pos, O.finite_difference_central(pos, 0.05)), -1)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-5)
    assert torch.equal(got[..., :7], pos)
    assert ops.traj_interpolate(torch.zeros(0, 4, 3, device=dev), 2).shape == (0, 10, 3)


def test_gpmp2_interpolated_jacobian_large(gpu_device):
    """H > 64 (carry of the segment gradients across 64-waypoint chunks): the workspace Jacobian against the
    oracle's autograd through the interpolated trajectory."""
    from motion_planning_baselines_amd import geometry as G, ops
    from oracle.geometry_ref import make_ref_geometry
    from oracle import planners_ref as O
    dev = gpu_device
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_dense_2d()
    geom = ops.DeviceGeometry(robot, field, dev)
    B, H, D, n = 3, 150, 2, 2
    gen = torch.Generator().manual_seed(1)
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    p0, p1 = torch.rand(B, 1, D, generator=gen) * 1.6 - 0.8, torch.rand(B, 1, D, generator=gen) * 1.6 - 0.8
    pos = p0 * (1 - a) + p1 * a + 0.01 * torch.randn(B, H, D, generator=gen)
    x = torch.cat([pos, torch.zeros(B, H, D)], -1).contiguous()
    ws = ops.gpmp2_workspace(B, H, D, dev)
    ops.gpmp2_linearize(x.to(dev), geom, ws, n_interp=n)
    torch.cuda.synchronize()
    jac = ws[:B * H * (D + 1) * 4].view(torch.float32).reshape(B, H, D + 1).cpu()     # workspace head: (B,H,D+1) fp32
    ta = dict(device='cpu', dtype=torch.float64)
    rr, rf = make_ref_geometry(robot, field, ta)
    xg = x.double().requires_grad_(True)
    xi = O.interpolate_trajs(xg, n)
    qi = rr.get_position(xi)
    err = rf.compute_cost(qi[:, 1:], rr.fk_map_collision(qi)[:, 1:])
    want = -torch.autograd.grad(err.sum(), xg)[0][:, 1:, :D]
    assert float(want.abs().max()) > 0.1                                              # collisions are active
    _close(jac[:, 1:, :D], want.numpy(), RTOL)
    q = rr.get_position(x.double())
    c = rf.compute_cost(q[:, 1:], rr.fk_map_collision(q)[:, 1:]).reshape(B, H - 1)
    _close(jac[:, 1:, D], c.numpy(), RTOL)


def test_hybrid_planner_warm_start(gpu_device):
    """HybridPlanner (hybrid_planner.py:10-89) around a duck-typed sample-based planner: ragged polylines and a
    missing path -> one mpb_traj_resample launch -> GPMP2 on the GPU."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.planners.gpmp2 import GPMP2
    from motion_planning_baselines_amd.planners.hybrid_planner import HybridPlanner
    from oracle import planners_ref as O
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_dense_2d()
    start, goal = torch.tensor([-0.9, -0.9]), torch.tensor([0.9, 0.9])
    gen = torch.Generator().manual_seed(2)

    class FakeRRT:
        start_state_pos, goal_state_pos = start, goal

        def __init__(self):
            self.paths = []
            for L in (2, 5, 70, 131):                 # incl. more than one 64-waypoint chunk
                mid = torch.rand(L - 2, 2, generator=gen) * 1.6 - 0.8
                self.paths.append(torch.cat([start[None], mid, goal[None]]))
            self.paths.append(None)                   # no solution found (hybrid_planner.py:47-51)
            self.paths.append(torch.cat([start[None], start[None], goal[None], goal[None]]))   # zero-length segments

        def optimize(self, refill_samples_buffer=False, debug=False, **kw):
            assert refill_samples_buffer
            return self.paths

    H, dt, n = 32, 0.1, 6
    rrt = FakeRRT()
    opt = GPMP2(robot=robot, n_dof=2, n_support_points=H, num_particles_per_goal=n, opt_iters=3, dt=dt,
                start_state=start.to(dev), multi_goal_states=goal[None].to(dev), step_size=0.5,
                initial_particle_means=torch.zeros(n, H, 4, device=dev), collision_fields=[field],
                sigma_start=1e-3, sigma_gp=1.0, sigma_coll=1e-2, sigma_goal_prior=1e-3,
                solver_params=dict(delta=1e-2, trust_region=True, method='cholesky'), tensor_args=ta)
    hyb = HybridPlanner(rrt, opt, tensor_args=ta)
    means = hyb.paths_to_initial_means(rrt.paths)
    assert means.shape == (1, n, H, 4)
    for i, p in enumerate(rrt.paths):
        want = O.resample_path(torch.stack((start, goal)) if p is None else p, H, dt)
        _close(means[0, i], want.numpy(), 1e-5)
        assert torch.equal(means[0, i, 0, :2].cpu(), start) and torch.equal(means[0, i, -1, :2].cpu(), goal)
    iters = hyb.optimize(return_iterations=True)
    assert iters.shape == (4, n, H, 4) and torch.isfinite(iters).all()
    assert torch.allclose(iters[0], means[0])
    last = hyb.optimize()
    assert last.shape == (n, H, 4)
    with pytest.raises(ValueError):
        ops.traj_resample(torch.zeros(2, 3, 2, device=dev), torch.tensor([3, 3], device=dev), H, dt)   # int64 lengths


def test_factor_classes(gpu_device):
    """costs/factors: GPFactor / UnaryFactor / FieldFactor / MultiMPPrior against goldens of the reference's own
    factor objects and the oracle."""
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.costs.factors.gp_factor import GPFactor
    from motion_planning_baselines_amd.planners.costs.factors.unary_factor import UnaryFactor
    from motion_planning_baselines_amd.planners.costs.factors.field_factor import FieldFactor
    from motion_planning_baselines_amd.planners.costs.factors.mp_priors_multi import MultiMPPrior
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    g = load_golden('cost_terms_panda')
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    x = T(g['trajs']).to(dev)
    # GPFactor: constants and error
    gp = GPFactor(D, float(g['sigma_gp']), dt, H - 1, ta)
    ta64 = dict(device='cpu', dtype=torch.float64)
    assert torch.allclose(gp.phi.cpu().double(), O.gp_phi(D, dt, ta64), rtol=1e-6)
    assert torch.allclose(gp.Q_inv[0].cpu().double(), O.gp_Q_inv(D, dt, float(g['sigma_gp']), ta64), rtol=1e-6)
    err, H1, H2 = gp.get_error(x)
    assert err.shape == (x.shape[0], H - 1, 2 * D, 1) and H1.shape == (H - 1, 2 * D, 2 * D)
    want = O.gp_error(T(g['trajs']).double(), O.gp_phi(D, dt, ta64))
    _close(err.squeeze(-1), want.numpy(), 1e-5)
    cost = (err.transpose(2, 3) @ gp.Q_inv[0].reshape(1, 1, 2 * D, 2 * D) @ err).sum(1).reshape(-1)   # CostGPTrajectory.eval
    _close(cost, g['gptraj_f64'], _bar(g, 'gptraj'))
    # UnaryFactor
    un = UnaryFactor(2 * D, 0.1, T(g['start']).to(dev), ta)
    e, Hm = un.get_error(x[:, [0]])
    assert e.shape == (x.shape[0], 2 * D, 1) and Hm.shape == (x.shape[0], 2 * D, 2 * D)
    assert torch.allclose(e.reshape(-1, 2 * D), T(g['start']).to(dev) - x[:, 0])
    # FieldFactor: error and Jacobian over traj_range [1, None]
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    ff = FieldFactor(D, 0.2, [1, None])
    err_f, Hf = ff.get_error(x, field, robot=robot)
    rr, rf = make_ref_geometry(robot, field, ta64)
    xg = T(g['trajs']).double().requires_grad_(True)
    q = rr.get_position(xg)
    e_ref = rf.compute_cost(q[:, 1:], rr.fk_map_collision(q)[:, 1:]).reshape(x.shape[0], H - 1)
    H_ref = -torch.autograd.grad(e_ref.sum(), xg)[0][:, 1:, :D]
    _close(err_f, e_ref.detach().numpy(), RTOL)
    assert Hf.shape == (x.shape[0], H - 1, D)
    if float(H_ref.abs().max()) > 0:
        _close(Hf, H_ref.numpy(), 2e-3)
    assert ff.K == pytest.approx(25.0)
    # MultiMPPrior: dense precision, mean and samples against the reference's golden (fp64 MultiMPPrior run)
    gp8 = load_golden('gp_prior_d2_h8')
    D2, H2n = int(gp8['D']), int(gp8['H'])
    ta64d = dict(device=dev, dtype=torch.float64)
    sK = torch.eye(2 * D2, dtype=torch.float64) / float(gp8['sigma_start']) ** 2
    gK = torch.eye(2 * D2, dtype=torch.float64) / float(gp8['sigma_goal']) ** 2
    Qi = O.gp_Q_inv(D2, float(gp8['dt']), float(gp8['sigma_gp']), ta64)
    torch.manual_seed(0)
    prior = MultiMPPrior(H2n - 1, float(gp8['dt']), 2 * D2, D2, sK, Qi, T(gp8['start']), K_g_inv=gK,
                         goal_states=T(gp8['goal']).unsqueeze(0), tensor_args=ta64d)
    np.testing.assert_allclose(prior.Sigma_inv.cpu().numpy(), gp8['Sigma_inv'], rtol=1e-9, atol=1e-9 * np.abs(gp8['Sigma_inv']).max())
    np.testing.assert_allclose(prior.means.cpu().numpy(), gp8['mean'], rtol=1e-12, atol=1e-12)
    smp = prior.sample(6)
    assert smp.shape == (1, 6, H2n, 2 * D2)
    # same torch seed -> same standard normals as the golden run (drawn on the CPU in the reference's order)
    _close(smp.reshape(1, 6, H2n, 2 * D2), gp8['samples'], 1e-5)
    # non-isotropic precisions take the dense path (tests/test_gpu_api_holes.py checks it against a reference golden)
    aniso = MultiMPPrior(H2n - 1, float(gp8['dt']), 2 * D2, D2, sK @ torch.arange(1, 2 * D2 + 1).diag().double(), Qi,
                         T(gp8['start']), tensor_args=ta64d)
    assert aniso._general and aniso.sample(3).shape == (1, 3, H2n, 2 * D2)


@pytest.mark.parametrize('kind', ['pm2d', 'panda', 'panda_boxes_only'])
def test_robot_field_api_with_autograd(gpu_device, kind):
    """The duck-typed robot / field objects (fk_map_collision, compute_cost) drive the oracle's restatement of
    CostCollision.eval -- written against that API exactly like the reference's cost layer -- on GPU tensors; values
    and autograd gradients equal the fused kernels and the CPU oracle."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.robot_field import device_robot_field
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    if kind == 'pm2d':
        robot, field = G.RobotPointMass(2, radius=0.02), G.env_dense_2d()
    elif kind == 'panda':
        robot, field = G.RobotPanda(), G.env_spheres_3d()
    else:   # no spheres -> no broad-phase grid: exhaustive evaluator
        robot, field = G.RobotPanda(), G.CollisionField(boxes=np.array([[0.4, 0.0, 0.4, 0.2, 0.6, 0.3]], np.float32), margin=0.05)
    D, H, B = robot.q_dim, 24, 19
    gen = torch.Generator().manual_seed(3)
    qmin, qmax = torch.from_numpy(robot.q_min_np), torch.from_numpy(robot.q_max_np)
    x = torch.cat([qmin + (qmax - qmin) * torch.rand(B, H, D, generator=gen), torch.randn(B, H, D, generator=gen)], -1)
    drobot, dfield = device_robot_field(robot, field, dev)
    xd = x.to(dev).requires_grad_(True)
    cost = O.collision_cost(xd, drobot, dfield, 0.5)                      # oracle code, device objects
    cost.sum().backward()
    ta64 = dict(device='cpu', dtype=torch.float64)
    rr, rf = make_ref_geometry(robot, field, ta64)
    x64 = x.double().requires_grad_(True)
    want = O.collision_cost(x64, rr, rf, 0.5)
    want.sum().backward()
    assert float(want.detach().max()) > 0
    _close(cost.detach(), want.detach().numpy(), RTOL)
    diff = (xd.grad.cpu().double() - x64.grad).abs()
    tol = 3e-4 * x64.grad.abs().max() + 3e-4 * x64.grad.abs()
    assert float((diff > tol).float().mean()) < 3e-3
    assert float(xd.grad[..., D:].abs().max()) == 0.0                     # no dependence on the velocity channels
    # the same numbers from the fused kernels
    geom = ops.DeviceGeometry(robot, field, dev)
    fused, fgrad = ops.cost_collision_grad(x.to(dev), geom, 1.0 / 0.5 ** 2)
    _close(fused, cost.detach().cpu().double().numpy(), 1e-5)
    assert torch.allclose(fgrad, xd.grad, rtol=1e-4, atol=1e-4 * float(xd.grad.abs().max()))
    # FK positions themselves
    pts = drobot.fk_map_collision(x.to(dev)[..., :D])
    ref_pts = rr.fk_map_collision(x.double()[..., :D])
    assert pts.shape == ref_pts.shape
    _close(pts, ref_pts.numpy(), 1e-5)


def test_cost_get_q_pos_vel_and_fk_map(gpu_device):
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    cc = C.CostCollision(robot, 16, field=field, sigma_coll=0.1, tensor_args=ta)
    x = torch.randn(2, 3, 16, 14, device=dev)                     # 4-D input is flattened (cost_functions.py:42-48)
    trajs, q_pos, q_vel, Hp = cc.get_q_pos_vel_and_fk_map(x)
    assert trajs.shape == (6, 16, 14) and q_pos.shape == (6, 16, 7) and q_vel.shape == (6, 16, 7) and Hp.shape == (6, 16, 31, 3)
    rr, _ = make_ref_geometry(robot, field, dict(device='cpu', dtype=torch.float64))
    _close(Hp, rr.fk_map_collision(q_pos.cpu().double()).numpy(), 1e-5)
