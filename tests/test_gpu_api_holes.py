"""-m gpu: the remaining holes of the planner API (VERDICT r02 "What's missing" 2, 3, 5).

  * CHOMP differentiates whatever cost it is handed (chomp.py:135-139): a caller-supplied callable goes through the
    caller's own torch.autograd.grad, the prior / clamp / mask / step stay in the HIP kernel -- checked against the
    reference-generated CHOMP goldens (the caller's cost written on robot_field.DeviceRobot / DeviceField) and against
    the oracle's autograd loop (a pure-torch cost);
  * MPPI takes ANY cost object (point.py:191-196; quirk Q6: its per-rollout costs collapse into one scalar shift) --
    checked against the reference-generated golden by handing the golden's collision cost over in a form the kernel
    does not fuse, and against the oracle for a caller-defined cost class;
  * StochGPMP(initial_particle_means='const_vel') (stoch_gpmp.py:107-111, :197-215)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, product_geometry_from_golden
from test_gpu_planners import T, make_cost, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['chomp_pm2d_soft', 'chomp_panda'])
def test_chomp_caller_supplied_cost_vs_golden(gpu_device, name):
    """The golden's collision cost re-written as a CALLER's cost (not one of the package's Cost classes) on the
    differentiable robot / field objects: CHOMP must reproduce the reference run."""
    from motion_planning_baselines_amd.planners.chomp import CHOMP
    from motion_planning_baselines_amd.robot_field import device_robot_field
    g = load_golden(name)
    dev = gpu_device
    robot, field = product_geometry_from_golden(g)
    drobot, dfield = device_robot_field(robot, field, dev)
    k = float(g['weight']) / float(g['sigma_coll']) ** 2

    calls = []

    def caller_cost(x, **kw):
        calls.append(x.shape)
        q = drobot.get_position(x)
        pts = drobot.fk_map_collision(q)
        return k * dfield.compute_cost(q[:, 1:], pts[:, 1:]).sum(-1)          # cost_functions.py:171-189, quirk Q5

    B = int(g['B'])
    pl = CHOMP(n_dof=int(g['D']), n_support_points=int(g['H']), num_particles_per_goal=B, opt_iters=1,
               dt=float(g['dt']), start_state=T(g['means0'][0, 0, :int(g['D'])]).to(dev), cost=caller_cost,
               weight_prior_cost=float(g['w_prior']), initial_particle_means=T(g['means0']).to(dev),
               step_size=float(g['lr']), grad_clip=float(g['clip']), pos_only=bool(g['pos_only']),
               tensor_args=dict(device=dev, dtype=torch.float32))
    for it in range(g['means'].shape[0]):
        traj = pl.optimize()
        assert rel_err(traj, T(g['traj'][it])) < 1e-4, it
    assert len(calls) == g['means'].shape[0]


def test_chomp_pure_torch_cost_vs_oracle(gpu_device):
    """A cost written in plain torch ops (smooth attraction / repulsion terms): the product loop against the oracle's
    autograd restatement of chomp.py:134-149 on the CPU."""
    from motion_planning_baselines_amd.planners.chomp import CHOMP
    from oracle import planners_ref as O
    dev = gpu_device
    B, H, D = 12, 64, 3
    dt, w_prior, lr, clip = 0.04, 1e-3, 0.05, 0.3
    gen = torch.Generator().manual_seed(5)
    a, b = torch.rand(B, 1, D, generator=gen) * 2 - 1, torch.rand(B, 1, D, generator=gen) * 2 - 1
    s = torch.linspace(0, 1, H).reshape(1, H, 1)
    pos = a * (1 - s) + b * s + 0.01 * torch.randn(B, H, D, generator=gen)
    means0 = torch.cat([pos, torch.zeros(B, H, D)], -1)
    centre = torch.tensor([0.1, -0.2, 0.3])

    def cost_fn(x, **kw):
        c = centre.to(x.device)
        d2 = ((x[..., :D] - c) ** 2).sum(-1)
        return (3.0 * torch.exp(-4.0 * d2) + 0.05 * torch.sin(3.0 * x[..., D:]).sum(-1)).sum(-1) + 0.1 * (x[:, -1, :D] ** 2).sum(-1)

    pl = CHOMP(n_dof=D, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt, start_state=means0[0, 0, :D].to(dev),
               cost=cost_fn, weight_prior_cost=w_prior, initial_particle_means=means0.to(dev), step_size=lr, grad_clip=clip,
               pos_only=False, tensor_args=dict(device=dev, dtype=torch.float32))
    cta = dict(device='cpu', dtype=torch.float32)
    R = O.chomp_precision(H, dt, cta)
    m = means0.clone()
    for it in range(4):
        m = O.chomp_iteration(m, R, cost_fn, w_prior, lr, clip)['means']
        pl.optimize()
        assert rel_err(pl._particle_means, m) < 1e-5, it
    # a cost autograd cannot trace raises instead of stepping on a zero gradient
    pl.cost = lambda x, **kw: torch.zeros(x.shape[0], device=x.device)
    with pytest.raises(NotImplementedError):
        pl.optimize()


def _mppi(g, dev, noise='torch_cpu'):
    from motion_planning_baselines_amd.planners.mppi import MPPI, PointParticleDynamics
    ta = dict(device=dev, dtype=torch.float32)
    S, Tn = int(g['S']), int(g['T'])
    system = PointParticleDynamics(rollout_steps=Tn, control_dim=2, state_dim=2, dt=float(g['dt']), discount=1.,
                                   goal_state=T(g['goal']).to(dev), ctrl_min=[-100, -100], ctrl_max=[100, 100],
                                   c_weights={'pos': float(g['c_pos']), 'vel': float(g['c_vel']), 'ctrl': float(g['c_ctrl']),
                                              'pos_T': float(g['c_pos_T']), 'vel_T': 0.}, tensor_args=ta)
    return MPPI(system, num_ctrl_samples=S, rollout_steps=Tn, opt_iters=1, control_std=[float(v) for v in g['control_std']],
                temp=float(g['temp']), step_size=float(g['step_size']), cov_prior_type=str(g['cov_type']), tensor_args=ta,
                noise=noise)


def test_mppi_any_cost_object_vs_golden(gpu_device):
    """The golden's collision cost handed over inside a caller-defined class (not the single collision cost the kernel
    fuses) takes the generic path -- cost.eval on the rollouts, outside the kernel -- and must
    reproduce the reference run: costs, means, best sample."""
    from motion_planning_baselines_amd.planners.costs.cost_functions import fusable_collision
    name = 'mppi_pm2d_indep_cost'
    g = load_golden(name)
    assert bool(g['with_cost'])
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    inner, _, _ = make_cost(g, dev)
    Tn = int(g['T'])

    class CallersCost:                 # not a class of cost_functions.py: the kernel cannot fuse it
        def eval(self, trajs, **kw):
            return inner.eval(trajs, **kw)

    cost = CallersCost()
    assert fusable_collision(cost) is None
    torch.manual_seed(1)
    pl = _mppi(g, dev)
    obs = dict(state=T(g['start']).to(dev), goal_state=T(g['goal']).to(dev), cost=cost)
    S = int(g['S'])
    for it in range(g['eps'].shape[0]):
        U, X, c = pl.optimize(**obs)
        assert U.shape == (S, Tn, 2) and X.shape == (S, Tn, 2) and c.shape == (S, 1)
        np.testing.assert_allclose(c.cpu().numpy(), g['costs'][it], rtol=5e-5)
        assert rel_err(pl.get_mean_controls(), T(g['mean'][it])) < 1e-4, it
        flat = np.stack([g['costs'][k].reshape(-1) for k in range(it + 1)])
        it_b, s_b = np.unravel_index(np.argmin(flat), flat.shape)
        np.testing.assert_allclose(float(pl.best_cost), flat[it_b, s_b], rtol=5e-5)
        np.testing.assert_allclose(pl.best_traj.cpu().numpy(), g['states'][it_b][s_b], rtol=1e-4, atol=1e-5)


def test_mppi_caller_defined_cost_vs_oracle(gpu_device):
    """A caller's own cost class (only `.eval`): its scalar shift is added to every sample's cost -- against the oracle
    loop fed the same noise."""
    from oracle import planners_ref as O
    g = load_golden('mppi_pm2d_const')
    dev = gpu_device

    class Energy:
        def eval(self, full_traj, **kw):
            assert full_traj.shape[-1] == 4                                   # cat(X, U), point.py:192
            return (full_traj[..., :2] ** 2).sum(-1).sum(-1) * 0.3 + full_traj[..., 2:].abs().sum(-1).sum(-1) * 0.01

    torch.manual_seed(0)
    pl = _mppi(g, dev)
    obs = dict(state=T(g['start']).to(dev), goal_state=T(g['goal']).to(dev), cost=Energy())
    cpu = dict(device='cpu', dtype=torch.float32)
    Tn, S = int(g['T']), int(g['S'])
    mean = torch.zeros(Tn, 2)
    cw = {'pos': float(g['c_pos']), 'vel': float(g['c_vel']), 'ctrl': float(g['c_ctrl']), 'pos_T': float(g['c_pos_T'])}
    disc = torch.ones(Tn)
    best = np.inf
    for it in range(g['eps'].shape[0]):
        eps = T(g['eps'][it]).reshape(2, S, Tn)
        kw = dict(dt=float(g['dt']), ctrl_min=torch.tensor([-100., -100.]), ctrl_max=torch.tensor([100., 100.]), c_weights=cw,
                  discount_seq=disc, temp=float(g['temp']), step_size=float(g['step_size']), state_dim=2)
        dry = O.mppi_iteration(mean, eps, T(g['scale_tril']).float(), T(g['Cov_inv']).float(), T(g['start']), T(g['goal']), **kw)
        shift = float(Energy().eval(torch.cat((dry['states'], dry['controls']), -1)).sum(-1))
        ref = O.mppi_iteration(mean, eps, T(g['scale_tril']).float(), T(g['Cov_inv']).float(), T(g['start']), T(g['goal']), shift_cost=shift, **kw)
        U, X, c = pl.optimize(**obs)
        np.testing.assert_allclose(c.cpu().numpy(), ref['costs'].numpy(), rtol=5e-5)
        assert rel_err(pl.get_mean_controls(), ref['mean']) < 1e-4, it
        mean = ref['mean']
        best = min(best, float(ref['costs'].min()))
        np.testing.assert_allclose(float(pl.best_cost), best, rtol=5e-5)


def test_stoch_gpmp_const_vel_initial_means(gpu_device):
    from motion_planning_baselines_amd.planners.stoch_gpmp import StochGPMP
    g = load_golden('sgpmp_panda_h16_f64')
    dev = gpu_device
    robot, field = product_geometry_from_golden(g)
    S, H, D = int(g['S']), int(g['H']), int(g['D'])
    ppg, dt = 3, float(g['dt'])
    start = T(g['start']).float()
    goals = torch.stack([T(g['goal']).float(), T(g['goal']).float() * 0.5 + 0.1])
    pl = StochGPMP(robot=robot, n_dof=D, n_support_points=H, num_particles_per_goal=ppg, opt_iters=1, dt=dt,
                   start_state=start.to(dev), step_size=float(g['step_size']), multi_goal_states=goals.to(dev),
                   initial_particle_means='const_vel', sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0,
                   sigma_start_sample=float(g['sigma_start_sample']), sigma_goal_sample=float(g['sigma_goal_sample']),
                   sigma_gp_sample=float(g['sigma_gp_sample']), num_samples=S, temperature=float(g['temperature']),
                   collision_fields=[field], sigma_start=float(g['sigma_start']), sigma_gp=float(g['sigma_gp']),
                   sigma_coll=float(g['sigma_coll']), sigma_goal_prior=float(g['sigma_goal_prior']),
                   tensor_args=dict(device=dev, dtype=torch.float32), noise='philox', seed=3)
    # stoch_gpmp.py:197-215 restated: straight line per goal, velocity (goal - start) / (H dt), ppg copies
    want = torch.zeros(2, ppg, H, 2 * D)
    mean_vel = (goals[:, :D] - start[:D]) / (H * dt)
    for i in range(H):
        want[:, :, i, :D] = (start[:D] * (H - i - 1) / (H - 1) + goals[:, :D] * i / (H - 1)).unsqueeze(1)
    want[:, :, :, D:] = mean_vel.unsqueeze(1).unsqueeze(1)
    got = pl._particle_means.cpu()
    assert got.shape == (2 * ppg, H, 2 * D)
    assert torch.allclose(got, want.flatten(0, 1), rtol=1e-6, atol=1e-7)
    traj = pl.optimize(opt_iters=2)
    assert torch.isfinite(traj).all()
    pl.reset(initial_particle_means='const_vel')
    assert torch.allclose(pl._particle_means.cpu(), want.flatten(0, 1), rtol=1e-6, atol=1e-7)


def test_multi_mp_prior_general_precisions_vs_golden(gpu_device):
    """MultiMPPrior with NON-isotropic start / GP / goal precisions (mp_priors_multi.py:213-251 accepts any matrices): the
    dense path (host fp64 factor, GPU dense product) against the reference-generated golden -- Sigma_inv and samples on the
    reference's own normals; plus the device-noise stream's covariance."""
    from motion_planning_baselines_amd.planners.costs.factors.mp_priors_multi import MultiMPPrior
    g = load_golden('gp_prior_general_d2_h6')
    dev = gpu_device
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    ta = dict(device=dev, dtype=torch.float32)
    f = lambda k: torch.from_numpy(g[k])
    torch.manual_seed(2)                       # the golden's seed: MultiMPPrior.sample draws (n, modes, M) fp64 on the CPU generator
    pr = MultiMPPrior(H - 1, dt, 2 * D, D, f('K_s_inv'), f('K_gp_inv'), f('start'), K_g_inv=f('K_g_inv'), goal_states=f('goals'),
                      tensor_args=ta, noise='torch_cpu')
    assert pr._general and pr.num_modes == 2
    Sref = g['Sigma_inv']
    np.testing.assert_allclose(pr.Sigma_inv.cpu().double().numpy(), Sref, rtol=2e-6, atol=1e-6 * np.abs(Sref).max())
    np.testing.assert_allclose(pr.means.cpu().numpy(), g['mean'], rtol=1e-12, atol=1e-15)
    smp = pr.sample(5)
    want = g['samples']
    assert tuple(smp.shape) == want.shape
    np.testing.assert_allclose(smp.cpu().numpy(), want, rtol=2e-6, atol=1e-6 * np.abs(want).max())
    # device noise: sample covariance against K = Sigma_inv^-1
    pr2 = MultiMPPrior(H - 1, dt, 2 * D, D, f('K_s_inv'), f('K_gp_inv'), f('start'), K_g_inv=f('K_g_inv'), goal_states=f('goals'),
                       tensor_args=ta, noise='philox', seed=4)
    ns = 20000
    s2 = pr2.sample(ns).double().cpu()                                # (2, ns, H, 2D)
    dev_ = (s2[0] - torch.from_numpy(g['mean'][0]).reshape(H, 2 * D)).reshape(ns, -1)
    emp = dev_.t() @ dev_ / ns
    K = np.linalg.inv(Sref)
    assert float(np.abs(emp.numpy() - K).max() / np.abs(K).max()) < 0.05


# ---- MPPI's system / prior objects as callable API (VERDICT r03 missing #3) --------------------------------------------------

def _mppi_system(g, dev, **kw):
    from motion_planning_baselines_amd.planners.dynamics.point import PointParticleDynamics
    ta = dict(device=dev, dtype=torch.float32)
    return PointParticleDynamics(rollout_steps=int(g['T']), control_dim=2, state_dim=2, dt=float(g['dt']), discount=1.,
                                 goal_state=T(g['goal']).to(dev), ctrl_min=[-100, -100], ctrl_max=[100, 100],
                                 c_weights={'pos': float(g['c_pos']), 'vel': float(g['c_vel']), 'ctrl': float(g['c_ctrl']),
                                            'pos_T': float(g['c_pos_T']), 'vel_T': 0.}, tensor_args=ta, **kw)


@pytest.mark.parametrize('name', ['mppi_pm2d_const', 'mppi_pm2d_indep_cost'])
def test_point_dynamics_and_traj_cost_vs_reference_golden(gpu_device, name):
    """PointParticleDynamics.dynamics / .traj_cost called the way the reference's MPPI calls them (mppi.py:190-210, :111-128):
    the Euler rollout of the golden's control samples through dynamics() reproduces its states, traj_cost() (+ the
    importance term, which is zero while the mean is zero) its costs."""
    g = load_golden(name)
    dev = gpu_device
    system = _mppi_system(g, dev)
    S, Tn = int(g['S']), int(g['T'])
    U = T(g['controls'][0]).to(dev)                                     # (S, T, 2): iteration 0, mean = 0
    X = torch.empty(S, Tn, 2, device=dev)
    X[:, 0] = T(g['start']).to(dev)
    for i in range(Tn - 1):                                             # mppi.py:203-209, verbatim call pattern
        X[:, i + 1] = system.dynamics(X[:, i].unsqueeze(1), U[:, i].unsqueeze(1)).squeeze(1)
    np.testing.assert_allclose(X.cpu().numpy(), g['states'][0], rtol=1e-5, atol=1e-6)
    obs = dict(goal_state=T(g['goal']).to(dev))
    if bool(g['with_cost']):
        from test_gpu_planners import make_cost
        obs['cost'], _, _ = make_cost(g, dev)
    costs = system.traj_cost(X.transpose(0, 1).unsqueeze(2), U.transpose(0, 1).unsqueeze(2), **obs)
    assert costs.shape == (S, 1)
    np.testing.assert_allclose(costs.cpu().numpy(), g['costs'][0], rtol=5e-5)
    # clamping and the single-step interface (point.py:91-100)
    tight = _mppi_system(g, dev)
    tight.ctrl_min, tight.ctrl_max = torch.tensor([-0.1, -0.2], device=dev), torch.tensor([0.3, 0.05], device=dev)
    x = torch.zeros(3, 1, 2, device=dev)
    u = torch.tensor([[[1.0, -1.0]], [[-1.0, 1.0]], [[0.2, 0.01]]], device=dev)
    want = np.clip(u.cpu().numpy(), [-0.1, -0.2], [0.3, 0.05]) * float(g['dt'])
    np.testing.assert_allclose(tight.dynamics(x, u).cpu().numpy(), want, rtol=1e-6)
    s1, c1 = system.step(torch.tensor([0.5, -0.25]))
    np.testing.assert_allclose(s1.cpu().numpy(), np.array([0.5, -0.25], np.float32) * float(g['dt']), rtol=1e-6)
    assert c1.ndim == 0 and float(c1) > 0
    # noisy dynamics: the noise enters the control channel scaled by dyn_std (statistics only: torch.randn draws it)
    noisy = _mppi_system(g, dev, deterministic=False, dyn_std=np.array([0.5, 0.0, 0.0, 0.0]))
    xn = noisy.dynamics(torch.zeros(4096, 1, 2, device=dev), torch.zeros(4096, 1, 2, device=dev))
    assert abs(float(xn[..., 0].std()) / (0.5 * float(g['dt'])) - 1.0) < 0.1 and float(xn[..., 1].abs().max()) == 0.0


@pytest.mark.parametrize('name', ['mppi_pm2d_const', 'mppi_pm2d_indep_cost'])
def test_control_trajectory_gaussian_sample_vs_reference_golden(gpu_device, name):
    """ControlTrajectoryGaussian.sample (gaussian.py:276-298) on the golden's standard normals reproduces its control
    samples, iteration by iteration with the means the reference had; MPPI.ctrl_dist is that object."""
    from motion_planning_baselines_amd.planners.priors.gaussian import ControlTrajectoryGaussian, get_multivar_gaussian_prior
    g = load_golden(name)
    dev = gpu_device
    ta = dict(device=dev, dtype=torch.float32)
    S, Tn = int(g['S']), int(g['T'])
    dist = get_multivar_gaussian_prior([float(v) for v in g['control_std']], Tn, 2, Cov_type=str(g['cov_type']), tensor_args=ta)
    assert isinstance(dist, ControlTrajectoryGaussian) and rel_err(dist.Cov, T(g['Cov'])) == 0.0
    for it in range(g['eps'].shape[0]):
        if it > 0:
            dist.update_means(T(g['mean'][it - 1]).to(dev))
        U = dist.sample(S, eps=T(g['eps'][it]))
        assert U.shape == (S, Tn, 2)
        assert rel_err(U, T(g["controls"][it])) < 1e-5, it      # (the reference factors and multiplies in fp32: 2e-6 measured)
    # device noise: fresh, reproducible streams with the right covariance
    a, b = dist.sample(4096), dist.sample(4096)
    assert not torch.equal(a, b)
    emp = torch.cov((a[:, :, 0] - dist.mu[:, 0]).t().double())
    C = T(g['Cov'])[..., 0].double()
    assert float((emp.cpu() - C).abs().max() / C.abs().max()) < 0.15
    system = _mppi_system(g, dev)
    from motion_planning_baselines_amd.planners.mppi import MPPI
    pl = MPPI(system, num_ctrl_samples=S, rollout_steps=Tn, opt_iters=1, control_std=[float(v) for v in g['control_std']],
              cov_prior_type=str(g['cov_type']), tensor_args=ta)
    assert rel_err(pl.ctrl_dist.Cov, T(g['Cov'])) == 0.0 and pl.ctrl_dist.sample(5).shape == (5, Tn, 2)
    with pytest.raises(NotImplementedError):
        dist.log_prob(a)


def test_planner_prior_helpers_vs_reference_golden(gpu_device):
    """GPMP2.get_dist / set_prior_factors (gpmp2.py:201-271), StochGPMP.get_prior_dist (stoch_gpmp.py:212-233) and
    OptimizationPlanner.get_GP_prior (base.py:115-139): thin constructors of MultiMPPrior / the factor objects on the
    planner's own horizon -- checked through the general-prior golden (same arguments, same samples)."""
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.gpmp2 import GPMP2
    from motion_planning_baselines_amd.planners.stoch_gpmp import StochGPMP
    g = load_golden('gp_prior_general_d2_h6')
    dev = gpu_device
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    ta = dict(device=dev, dtype=torch.float32)
    f = lambda k: torch.from_numpy(g[k])
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d()
    goals = f('goals').float().to(dev)
    start = f('start').float()[:D].to(dev)
    kw = dict(robot=robot, n_dof=D, n_support_points=H, num_particles_per_goal=3, opt_iters=1, dt=dt, start_state=start,
              multi_goal_states=goals, sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0, sigma_start_sample=1e-3,
              sigma_goal_sample=1e-3, collision_fields=[field], tensor_args=ta)
    planners = [(GPMP2(**kw), 'get_dist'), (StochGPMP(sigma_gp_sample=1.0, num_samples=4, **kw), 'get_prior_dist')]
    planners.append((planners[0][0], 'get_GP_prior'))
    for pl, name in planners:
        torch.manual_seed(2)
        pr = getattr(pl, name)(f('K_s_inv'), f('K_gp_inv'), f('K_g_inv'), f('start'), goal_states=f('goals'))
        Sref = g['Sigma_inv']
        np.testing.assert_allclose(pr.Sigma_inv.cpu().double().numpy(), Sref, rtol=2e-6, atol=1e-6 * np.abs(Sref).max())
        smp = pr.sample(5)
        np.testing.assert_allclose(smp.cpu().numpy(), g['samples'], rtol=2e-6, atol=1e-6 * np.abs(g['samples']).max())
    for pl, _ in planners[:2]:
        pl.set_prior_factors()
        assert pl.start_prior_init.K.shape == (2 * D, 2 * D) and abs(float(pl.start_prior_init.K[0, 0]) - 1e6) < 1.0
        assert len(pl.multi_goal_prior_init) == goals.shape[0] == len(pl.multi_goal_prior_sample)
        assert pl.gp_prior_init.num_factors == H - 1 and pl.start_prior_sample.dim == 2 * D
    assert planners[1][0].gp_prior_sample.num_factors == H - 1


# ---- MultiMPPrior's remaining public methods (VERDICT r04 missing #1; mp_priors_multi.py:100-176, :213-259) ----------------

def test_multi_mp_prior_public_methods_vs_golden(gpu_device):
    """get_const_vel_mean / const_vel_trajectory / get_const_vel_covariance / log_prob / update_dist / set_Sigma_invs against values
    the reference class itself produced (tests/golden/make_goldens.py): the isotropic prior (log_prob through the STRUCTURED
    factor, no dense M x M matrix) and the general one (dense), incl. one precision per mode."""
    from motion_planning_baselines_amd.planners.costs.factors.gp_factor import GPFactor
    from motion_planning_baselines_amd.planners.costs.factors.mp_priors_multi import MultiMPPrior
    from motion_planning_baselines_amd.planners.costs.factors.unary_factor import UnaryFactor
    dev = gpu_device
    ta64 = dict(device=dev, dtype=torch.float64)
    f = lambda g, k: torch.from_numpy(g[k])
    # ---- isotropic factors: the structured path
    g = load_golden('gp_prior_d2_h8')
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    start, goal = f(g, 'start'), f(g, 'goal')
    sK = UnaryFactor(2 * D, float(g['sigma_start']), start.to(dev), ta64).K
    gK = UnaryFactor(2 * D, float(g['sigma_goal']), goal.to(dev), ta64).K
    Qi = GPFactor(D, float(g['sigma_gp']), dt, H - 1, ta64).Q_inv[0]
    pr = MultiMPPrior(H - 1, dt, 2 * D, D, sK, Qi, start, K_g_inv=gK, goal_states=goal.unsqueeze(0), tensor_args=ta64)
    assert not pr._general
    xs = f(g, 'samples').transpose(0, 1).reshape(6, 1, -1)
    off = xs + 0.05 * torch.linspace(-1, 1, xs.shape[-1], dtype=torch.float64)
    for x, key in ((xs, 'log_prob'), (off, 'log_prob_off')):
        lp = pr.log_prob(x.to(dev))
        assert tuple(lp.shape) == g[key].shape
        np.testing.assert_allclose(lp.cpu().numpy(), g[key], rtol=1e-9, atol=1e-6)
    assert pr._Sigma_inv is None                                     # the dense matrix was never built
    m = pr.get_const_vel_mean(start.to(dev), goal.unsqueeze(0).to(dev), dt, H - 1, D)
    np.testing.assert_allclose(m.cpu().numpy(), g['const_vel_mean'], rtol=1e-12, atol=1e-15)
    full = MultiMPPrior.const_vel_trajectory(start, goal, dt, H - 1, D, set_initial_final_vel_to_zero=False, tensor_args=dict(device='cpu', dtype=torch.float64))
    assert torch.allclose(full[:, D:], ((goal[:D] - start[:D]) / ((H - 1) * dt)).expand(H, D))
    Kp = pr.get_const_vel_covariance(dt, sK, Qi, gK)
    np.testing.assert_allclose(Kp.cpu().numpy(), g['const_vel_precision'], rtol=1e-10, atol=1e-10 * np.abs(g['const_vel_precision']).max())
    np.testing.assert_allclose(Kp.cpu().numpy(), pr.Sigma_inv.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(Kp.abs().max()))
    Kc = pr.get_const_vel_covariance(dt, sK, Qi, gK, precision_matrix=False)
    np.testing.assert_allclose(Kc.cpu().numpy(), g['const_vel_covariance'], rtol=1e-5, atol=1e-7 * np.abs(g['const_vel_covariance']).max())
    pr.update_dist(pr.means, pr.Sigma_invs)                          # the same distribution: stays structured
    assert not pr._general
    # ---- arbitrary precisions: the dense path, then one precision per mode
    g = load_golden('gp_prior_general_d2_h6')
    D, H, dt = int(g['D']), int(g['H']), float(g['dt'])
    torch.manual_seed(2)
    pr = MultiMPPrior(H - 1, dt, 2 * D, D, f(g, 'K_s_inv'), f(g, 'K_gp_inv'), f(g, 'start'), K_g_inv=f(g, 'K_g_inv'),
                      goal_states=f(g, 'goals'), tensor_args=ta64, noise='torch_cpu')
    xs = f(g, 'samples').transpose(0, 1).reshape(5, 2, -1)
    off = xs + 0.05 * torch.linspace(-1, 1, xs.shape[-1], dtype=torch.float64)
    np.testing.assert_allclose(pr.log_prob(xs.to(dev)).cpu().numpy(), g['log_prob'], rtol=1e-8, atol=1e-6)
    np.testing.assert_allclose(pr.log_prob(off.to(dev)).cpu().numpy(), g['log_prob_off'], rtol=1e-8, atol=1e-6)
    np.testing.assert_allclose(pr.get_const_vel_mean(f(g, 'start').to(dev), f(g, 'goals').to(dev), dt, H - 1, D).cpu().numpy(),
                               g['const_vel_mean'], rtol=1e-12, atol=1e-15)
    S2 = f(g, 'Sigma_invs2').to(dev)
    torch.manual_seed(2 + 7)                                         # the golden's seed for the second draw
    pr.set_Sigma_invs(S2)
    assert torch.equal(pr.Sigma_invs.cpu(), S2.cpu())
    smp2 = pr.sample(4)
    np.testing.assert_allclose(smp2.cpu().numpy(), g['samples2'], rtol=2e-6, atol=1e-6 * np.abs(g['samples2']).max())
    np.testing.assert_allclose(pr.log_prob(xs.to(dev)).cpu().numpy(), g['log_prob2'], rtol=1e-8, atol=1e-6)
    with pytest.raises(AssertionError):
        pr.set_Sigma_invs(S2[:1])


def test_gpmp2_dense_methods_and_gaussian_module_names(gpu_device):
    """VERDICT r04 missing #2 / #3: GPMP2._get_grad_terms / get_torch_solve as callables (the reference's dense methods, on
    device tensors) reproduce the golden's normal equations and step from the composite's own dense (A, b, K); GMM and
    get_indep_gaussian_prior exist with the reference's surface."""
    from test_gpu_planners import _gpmp2_from_golden
    from motion_planning_baselines_amd.planners.priors.gaussian import GMM, get_indep_gaussian_prior
    g = load_golden('gpmp2_pm2d_h8_f64')
    dev = gpu_device
    pl, robot, _ = _gpmp2_from_golden(g, dev)
    x0 = T(g['means0']).float().to(dev)
    A, b, K = (t.double() for t in pl.cost.get_linear_system(x0))
    JtJ, gr = pl._get_grad_terms(A, b, K, delta=float(g['delta']), trust_region=bool(g['trust_region']))
    B, H, D = int(g['B']), int(g['H']), int(g['D'])
    for method in ('cholesky', 'inverse', 'lstq'):
        d = pl.get_torch_solve(JtJ, gr, method=method).reshape(B, H, 2 * D)
        want = T(g['means'][0]).double() - T(g['means0']).double()
        assert float((float(g['step_size']) * d.cpu() - want).abs().max() / want.abs().max()) < 1e-4, method
    with pytest.raises(NotImplementedError):
        pl.get_torch_solve(JtJ, gr, method='cholesky-sparse')
    # the step the planner takes (block solve) is the step of these dense methods
    got = pl.optimize(opt_iters=1).cpu().double() - T(g['means0']).double()
    d = pl.get_torch_solve(JtJ, gr, method='cholesky').reshape(B, H, 2 * D).cpu()
    assert float((got - float(g['step_size']) * d).abs().max() / d.abs().max()) < 5e-5
    gm = GMM(torch.zeros(3, 5, 2, device=dev), torch.ones(3, 5, 2, device=dev), torch.ones(3, device=dev) / 3)
    s = gm.sample(7)
    assert tuple(s.shape) == (5, 7, 2) and tuple(gm.log_prob(s.transpose(0, 1)).shape) == (7,)
    pr = get_indep_gaussian_prior(0.3, 6, 2, mu_init=0.5, tensor_args=dict(device=dev, dtype=torch.float32))
    assert tuple(pr.sample().shape) == (6, 2) and float(pr.mean[0, 0]) == 0.5 and abs(float(pr.stddev[0, 0]) - 0.3) < 1e-7
