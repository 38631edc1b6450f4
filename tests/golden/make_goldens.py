"""Generate golden vectors by running the UNMODIFIED reference planner classes.

Runs ONLY in the build container (needs /root/reference).  The reference package is imported from
where it lies with a ``sys.modules`` stand-in for the absent ``torch_robotics`` dependency
(oracle/ref_stub.py) and the build-defined geometry back-end plugged in as the duck-typed ``robot`` /
``field`` (oracle/geometry_ref.py).  Only DATA is written (inputs, injected standard-normal draws,
per-iteration outputs) as small .npz fixtures next to this script; no reference source is copied.

    python tests/golden/make_goldens.py            # every fixture
    python tests/golden/make_goldens.py NAME ...   # only the named fixtures (e.g. stomp_panda_s32)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# output directory: next to this script, or MPB_GOLDEN_OUT (tests/test_goldens_regenerate.py writes to a temp dir and
# compares with the committed fixtures bit for bit)
HERE = os.environ.get('MPB_GOLDEN_OUT') or os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_stub  # noqa: E402
from oracle.geometry_ref import make_ref_geometry  # noqa: E402
from motion_planning_baselines_amd import geometry as G  # noqa: E402

ref_stub.import_reference()
import torch.distributions.multivariate_normal as _mvn  # noqa: E402
from mp_baselines.planners.stomp import STOMP  # noqa: E402
from mp_baselines.planners.chomp import CHOMP  # noqa: E402
from mp_baselines.planners.gpmp2 import GPMP2  # noqa: E402
from mp_baselines.planners.mppi import MPPI  # noqa: E402
from mp_baselines.planners.stoch_gpmp import StochGPMP  # noqa: E402
from mp_baselines.planners.dynamics.point import PointParticleDynamics  # noqa: E402
from mp_baselines.planners.costs.cost_functions import CostCollision, CostComposite  # noqa: E402
from mp_baselines.planners.costs.cost_functions import (CostGP, CostGPTrajectory,  # noqa: E402
                                                        CostGPTrajectoryPositionOnlyWrapper, CostSmoothnessCHOMP,
                                                        CostJointLimits, CostGoalPrior)
from mp_baselines.planners.costs.factors.mp_priors_multi import MultiMPPrior  # noqa: E402
from mp_baselines.planners.costs.factors.gp_factor import GPFactor  # noqa: E402
from mp_baselines.planners.costs.factors.unary_factor import UnaryFactor  # noqa: E402

TA32 = dict(device='cpu', dtype=torch.float32)
TA64 = dict(device='cpu', dtype=torch.float64)
ONLY = set(sys.argv[1:])     # fixture names to (re)generate; empty = all.  Every generator seeds itself, so the
                             # fixtures do not depend on which others were generated in the same run


def fixture(fn):
    def run(name, *a, **k):
        if ONLY and name not in ONLY:
            return
        return fn(name, *a, **k)
    return run


class EpsRecorder:
    """Records every standard-normal draw MultivariateNormal.rsample makes, in call order."""

    def __init__(self):
        self.draws = []
        self._orig = _mvn._standard_normal

    def __enter__(self):
        def rec(shape, dtype, device):
            e = self._orig(shape, dtype=dtype, device=device)
            self.draws.append(e.clone())
            return e
        _mvn._standard_normal = rec
        return self

    def __exit__(self, *a):
        _mvn._standard_normal = self._orig


def npf(t):
    return t.detach().cpu().numpy().copy()


def straight_line_means(start, goal, H, dt, P, pos_only, noise=0.0, gen=None):
    D = start.shape[-1]
    a = torch.linspace(0, 1, H).reshape(H, 1)
    pos = start.reshape(1, D) * (1 - a) + goal.reshape(1, D) * a
    pos = pos.unsqueeze(0).repeat(P, 1, 1)
    if noise > 0:
        pos[:, 1:-1] += noise * torch.randn(P, H - 2, D, generator=gen)
    if pos_only:
        return pos
    vel = ((goal - start) / ((H - 1) * dt)).reshape(1, 1, D).repeat(P, H, 1)
    return torch.cat([pos, vel], -1)


def geom_arrays(robot, field):
    rs, fs = robot.spec(), field.spec()
    return dict(robot_kind=np.int32(rs['kind']), n_dof=np.int32(rs['n_dof']),
                joint_tf=rs['joint_tf'], link_frame=rs['link_frame'], link_offset=rs['link_offset'],
                link_radius=rs['link_radius'], spheres=fs['spheres'], boxes=fs['boxes'],
                margin=np.float32(fs['margin']))


def free_configs(robot, field, n, seed, ta):
    rr, rf = make_ref_geometry(robot, field, ta)
    g = torch.Generator().manual_seed(seed)
    out = []
    while len(out) < n:
        q = rr.q_min + (rr.q_max - rr.q_min) * torch.rand(64, robot.q_dim, generator=g)
        c = rf.compute_cost(q.unsqueeze(1), rr.fk_map_collision(q.unsqueeze(1))).reshape(-1)
        out += [qi for qi, ci in zip(q, c) if ci == 0]
    return torch.stack(out[:n])


# ------------------------------------------------------------------------------------------------

@fixture
def gen_stomp(name, robot, field, start, goal, P, S, H, dt, sigma_coll, pos_only, iters, seed,
              lr=0.1, temperature=1.0, sigma_spectral=0.1, init_noise=0.0):
    ta = TA32
    rr, rf = make_ref_geometry(robot, field, ta)
    cost = CostComposite(rr, H, [CostCollision(rr, H, field=rf, sigma_coll=sigma_coll, tensor_args=ta)],
                         tensor_args=ta)
    g = torch.Generator().manual_seed(seed + 1000)
    means0 = straight_line_means(start, goal, H, dt, P, pos_only, noise=init_noise, gen=g)
    torch.manual_seed(seed)
    with EpsRecorder() as rec:
        pl = STOMP(n_dof=robot.q_dim, n_support_points=H, num_particles_per_goal=P, num_samples=S,
                   opt_iters=1, dt=dt, start_state=start, cost=cost, initial_particle_means=means0,
                   multi_goal_states=goal.unsqueeze(0), temperature=temperature, step_size=lr,
                   sigma_spectral=sigma_spectral, pos_only=pos_only, tensor_args=ta)
        n_reset_draws = len(rec.draws)   # STOMP.reset() samples once (stomp.py:120)
        out = dict(samples=[], costs=[], weights=[], means=[], traj=[])
        for _ in range(iters):
            traj = pl.optimize()
            out['samples'].append(npf(pl.state_particles))
            out['costs'].append(npf(pl.costs))
            out['weights'].append(npf(pl._weights).reshape(P, S))
            out['means'].append(npf(pl._particle_means))
            out['traj'].append(npf(traj))
    eps = np.stack([npf(e) for e in rec.draws[n_reset_draws:]])
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        planner='stomp', P=P, S=S, H=H, D=robot.q_dim, dt=dt, sigma_coll=sigma_coll, pos_only=pos_only,
        lr=lr, temperature=temperature, sigma_spectral=sigma_spectral, seed=seed,
        start=npf(start), goal=npf(goal), means0=npf(means0), eps_reset=npf(rec.draws[0]), eps=eps,
        R=npf(pl.Sigma_inv), Sigma=npf(pl.Sigma), L=npf(pl._noise_dist._unbroadcasted_scale_tril),
        **{k: np.stack(v) for k, v in out.items()}, **geom_arrays(robot, field))
    print(name, 'eps', eps.shape, 'final cost min/max', out['costs'][-1].min(), out['costs'][-1].max())


@fixture
def gen_chomp(name, robot, field, starts, goals, H, dt, sigma_coll, weight, iters, seed,
              w_prior=1e-4, lr=0.05, clip=0.05, pos_only=False, init_noise=0.01):
    ta = TA32
    rr, rf = make_ref_geometry(robot, field, ta)
    cost = CostComposite(rr, H, [CostCollision(rr, H, field=rf, sigma_coll=sigma_coll, tensor_args=ta)],
                         weights_cost_l=[weight], tensor_args=ta)
    g = torch.Generator().manual_seed(seed + 1000)
    B = starts.shape[0]
    means0 = torch.cat([straight_line_means(starts[i], goals[i], H, dt, 1, pos_only, noise=init_noise, gen=g)
                        for i in range(B)], 0)
    pl = CHOMP(n_dof=robot.q_dim, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt,
               start_state=starts[0], cost=cost, weight_prior_cost=w_prior, initial_particle_means=means0,
               step_size=lr, grad_clip=clip, multi_goal_states=goals[:1], pos_only=pos_only, tensor_args=ta)
    means, trajs = [], []
    for _ in range(iters):
        t = pl.optimize()
        means.append(npf(pl._particle_means))
        trajs.append(npf(t))
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        planner='chomp', B=B, H=H, D=robot.q_dim, dt=dt, sigma_coll=sigma_coll, weight=weight, w_prior=w_prior,
        lr=lr, clip=clip, pos_only=pos_only, means0=npf(means0), R=npf(pl.Sigma_inv),
        means=np.stack(means), traj=np.stack(trajs), **geom_arrays(robot, field))
    print(name, 'moved', np.abs(means[-1] - npf(means0)).max())


@fixture
def gen_gpmp2(name, robot, field, start, goal, B, H, dt, iters, seed, ta, sig=None, delta=1e-2,
              trust_region=True, step_size=1.0, init_noise=0.02, n_interp=None, extra_fields=(), keep_band=True):
    sig = sig or dict(sigma_start=1e-5, sigma_gp=1e-2, sigma_coll=1e-5, sigma_goal_prior=1e-5)
    rr, rf = make_ref_geometry(robot, field, ta)
    rfs = [rf] + [make_ref_geometry(robot, f, ta)[1] for f in extra_fields]      # one CostCollision per field
    start, goal = start.to(**ta), goal.to(**ta)
    g = torch.Generator().manual_seed(seed + 1000)
    means0 = straight_line_means(start.float(), goal.float(), H, dt, B, False, noise=init_noise, gen=g).to(**ta)
    means0[:, 0, robot.q_dim:] = 0
    means0[:, -1, robot.q_dim:] = 0
    pl = GPMP2(robot=rr, n_dof=robot.q_dim, n_support_points=H, num_particles_per_goal=B, opt_iters=1, dt=dt,
               start_state=start, step_size=step_size, multi_goal_states=goal.unsqueeze(0),
               initial_particle_means=means0.clone().unsqueeze(0),
               sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0,
               sigma_start_sample=1e-3, sigma_goal_sample=1e-3,
               solver_params=dict(delta=delta, trust_region=trust_region, method='cholesky'),
               collision_fields=rfs, tensor_args=ta, n_interpolated_points=n_interp, **sig)
    # quirk Q13: GPMP2.__init__ (gpmp2.py:94-131) swallows n_interpolated_points without forwarding it to the
    # base class, which therefore stores None (base.py:85); the interpolated Jacobian only runs when the
    # attribute is set on the instance afterwards
    assert pl.n_interpolated_points is None
    pl.n_interpolated_points = n_interp
    rec = dict(A=[], b=[], K=[], JtJ=[], g=[], means=[], costs=[])
    orig_ls = pl.cost.get_linear_system
    orig_gt = pl._get_grad_terms

    def ls(*a, **k):
        A, b, K = orig_ls(*a, **k)
        rec['A'].append(npf(A)); rec['b'].append(npf(b)); rec['K'].append(npf(torch.diagonal(K, dim1=-2, dim2=-1)))
        rec['_Kfull'] = K
        return A, b, K

    def gt(*a, **k):
        J, gg = orig_gt(*a, **k)
        rec['JtJ'].append(npf(J)); rec['g'].append(npf(gg))
        return J, gg
    pl.cost.get_linear_system = ls
    pl._get_grad_terms = gt
    for _ in range(iters):
        pl.optimize(opt_iters=1)
        rec['means'].append(npf(pl._particle_means))
        rec['costs'].append(npf(pl.costs))
    rec.pop('_Kfull')
    small = H * robot.q_dim * 2 <= 64
    keep = {k: np.stack(v) for k, v in rec.items() if small or k in ('means', 'costs', 'g')}
    if not small and keep_band:   # keep the banded part of JtJ only (diagonal blocks + first off-diagonal blocks)
        dim = 2 * robot.q_dim
        J = np.stack(rec['JtJ'])
        keep['JtJ_diag'] = np.stack([J[:, :, t * dim:(t + 1) * dim, t * dim:(t + 1) * dim] for t in range(H)], 2)
        keep['JtJ_off'] = np.stack([J[:, :, t * dim:(t + 1) * dim, (t + 1) * dim:(t + 2) * dim] for t in range(H - 1)], 2)
        mask = np.ones_like(J[0, 0], dtype=bool)
        for t in range(H):
            mask[t * dim:(t + 1) * dim, max(0, t - 1) * dim:(t + 2) * dim] = False
        keep['JtJ_outside_band_absmax'] = np.abs(J[..., mask]).max()
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        planner='gpmp2', B=B, H=H, D=robot.q_dim, dt=dt, delta=delta, trust_region=trust_region,
        n_interp=0 if n_interp is None else n_interp,
        step_size=step_size, dtype=str(ta['dtype']), start=npf(start), goal=npf(goal), means0=npf(means0),
        **sig, **keep, **geom_arrays(robot, field),
        **{f'extra{i}_{k}': v for i, f in enumerate(extra_fields) for k, v in
           dict(spheres=f.spec()['spheres'], boxes=f.spec()['boxes'], margin=np.float32(f.spec()['margin'])).items()},
        n_extra_fields=len(extra_fields))
    print(name, 'costs', rec['costs'][0][:3], '->', rec['costs'][-1][:3])


@fixture
def gen_stoch_gpmp(name, robot, field, start, goal, P, S, H, dt, iters, seed, ta, temperature=1.0, step_size=0.5,
                   sig_sample=(1e-3, 1e-3, 0.5), sig_cost=None, init_noise=0.02):
    """StochGPMP (stoch_gpmp.py): samples from the full GP prior around each particle, composite cost
    (GP + goal prior + collision) + importance term, softmax update without Sigma."""
    sig_cost = sig_cost or dict(sigma_start=1e-2, sigma_gp=1.0, sigma_coll=1e-1, sigma_goal_prior=1e-2)
    rr, rf = make_ref_geometry(robot, field, ta)
    start, goal = start.to(**ta), goal.to(**ta)
    g = torch.Generator().manual_seed(seed + 1000)
    means0 = straight_line_means(start.float(), goal.float(), H, dt, P, False, noise=init_noise, gen=g).to(**ta)
    torch.manual_seed(seed)
    with EpsRecorder() as rec:
        pl = StochGPMP(robot=rr, n_dof=robot.q_dim, n_support_points=H, num_particles_per_goal=P, opt_iters=1, dt=dt,
                       start_state=start, step_size=step_size, multi_goal_states=goal.unsqueeze(0),
                       initial_particle_means=means0.clone().unsqueeze(0),
                       sigma_start_init=1e-3, sigma_goal_init=1e-3, sigma_gp_init=1.0,
                       sigma_start_sample=sig_sample[0], sigma_goal_sample=sig_sample[1], sigma_gp_sample=sig_sample[2],
                       num_samples=S, temperature=temperature, collision_fields=[rf], tensor_args=ta, **sig_cost)
        n_reset = len(rec.draws)
        out = dict(samples=[], costs=[], weights=[], means=[])
        orig_costs = pl._get_costs

        def rec_costs(**kw):     # the class does not keep the costs: record what sample_and_eval sees
            c = orig_costs(**kw)
            out['costs'].append(npf(c))
            return c
        pl._get_costs = rec_costs
        for _ in range(iters):
            pl.optimize(opt_iters=1)
            out['samples'].append(npf(pl.state_samples))
            out['weights'].append(npf(pl._weights).reshape(P, S))
            out['means'].append(npf(pl._particle_means))
    eps = np.stack([npf(e) for e in rec.draws[n_reset:]])
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), planner='stoch_gpmp', P=P, S=S, H=H, D=robot.q_dim, dt=dt,
        temperature=temperature, step_size=step_size, dtype=str(ta['dtype']), start=npf(start), goal=npf(goal),
        means0=npf(means0), eps=eps, sigma_start_sample=sig_sample[0], sigma_goal_sample=sig_sample[1],
        sigma_gp_sample=sig_sample[2], Sigma_inv=npf(pl.Sigma_inv), **sig_cost,
        **{k: np.stack(v) for k, v in out.items()}, **geom_arrays(robot, field))
    print(name, 'eps', eps.shape, 'cost range', out['costs'][-1].min(), out['costs'][-1].max(),
          'moved', np.abs(out['means'][-1] - npf(means0)).max())


@fixture
def gen_mppi(name, S, T, dt, iters, seed, cov_type='const_ctrl', control_std=(0.15, 0.15), temp=1.0,
             step_size=1.0, with_cost=False):
    ta = TA32
    start = torch.tensor([-0.8, -0.8], **ta)
    goal = torch.tensor([0.8, 0.8], **ta)
    cw = {'pos': 1., 'vel': 1., 'ctrl': 1., 'pos_T': 1000., 'vel_T': 0.}
    system = PointParticleDynamics(rollout_steps=T, control_dim=2, state_dim=2, dt=dt, discount=1.,
                                   goal_state=goal, ctrl_min=[-100, -100], ctrl_max=[100, 100], verbose=False,
                                   c_weights=cw, tensor_args=ta)
    torch.manual_seed(seed)
    robot = G.RobotPointMass(2, radius=0.01)
    field = G.env_grid_circles_2d()
    obs = dict(state=start, goal_state=goal)
    if with_cost:
        rr, rf = make_ref_geometry(robot, field, ta)
        obs['cost'] = CostComposite(rr, T, [CostCollision(rr, T, field=rf, sigma_coll=1e-3, tensor_args=ta)],
                                    tensor_args=ta)
    with EpsRecorder() as rec:
        pl = MPPI(system, num_ctrl_samples=S, rollout_steps=T, opt_iters=1, control_std=list(control_std),
                  temp=temp, step_size=step_size, cov_prior_type=cov_type, tensor_args=ta)
        out = dict(controls=[], states=[], costs=[], weights=[], mean=[])
        for _ in range(iters):
            U, X, c = pl.optimize(**obs)
            out['controls'].append(npf(U)); out['states'].append(npf(X)); out['costs'].append(npf(c))
            out['weights'].append(npf(pl.weights)); out['mean'].append(npf(pl._mean))
    eps = np.stack([npf(e) for e in rec.draws]).reshape(iters, 2, S, T)
    tril = np.stack([npf(d._unbroadcasted_scale_tril) for d in pl.ctrl_dist.list_ctrl_dists])
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        planner='mppi', S=S, T=T, dt=dt, temp=temp, step_size=step_size, cov_type=cov_type,
        control_std=np.array(control_std, np.float32), start=npf(start), goal=npf(goal), with_cost=with_cost,
        c_pos=cw['pos'], c_vel=cw['vel'], c_ctrl=cw['ctrl'], c_pos_T=cw['pos_T'],
        Cov=npf(pl.ctrl_dist.Cov), Cov_inv=npf(pl.Cov_inv), scale_tril=tril, eps=eps,
        **{k: np.stack(v) for k, v in out.items()}, **geom_arrays(robot, field))
    print(name, 'cost', out['costs'][0].min(), '->', out['costs'][-1].min())


@fixture
def gen_gp_prior(name, D, H, dt, seed):
    """MultiMPPrior precision / mean / samples as OptimizationPlanner.get_random_trajs builds them
    (base.py:155-202)."""
    ta = TA64
    start = torch.cat([torch.linspace(-0.5, 0.3, D), torch.zeros(D)]).to(**ta)
    goal = torch.cat([torch.linspace(0.4, -0.2, D), torch.zeros(D)]).to(**ta)
    sK = UnaryFactor(2 * D, 1e-3, start, ta).K
    gK = UnaryFactor(2 * D, 1e-3, goal, ta).K
    Qi = GPFactor(D, 5.0, dt, H - 1, ta).Q_inv[0]
    torch.manual_seed(seed)
    with EpsRecorder() as rec:
        prior = MultiMPPrior(H - 1, dt, 2 * D, D, sK, Qi, start, K_g_inv=gK, goal_states=goal.unsqueeze(0),
                             tensor_args=ta)
        smp = prior.sample(6)
    # the class's other public methods (mp_priors_multi.py:130-176, :213-259): log density of the samples (and of points well
    # off the mean), the constant-velocity mean and the precision from the factor matrices
    xs = smp.transpose(0, 1).reshape(6, 1, -1)
    off = xs + 0.05 * torch.linspace(-1, 1, xs.shape[-1], dtype=torch.float64)
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), planner='gp_prior', D=D, H=H, dt=dt,
        start=npf(start), goal=npf(goal), sigma_start=1e-3, sigma_goal=1e-3, sigma_gp=5.0,
        Sigma_inv=npf(prior.Sigma_inv), mean=npf(prior.means), scale_tril=npf(prior.dist._unbroadcasted_scale_tril),
        eps=npf(rec.draws[0]), samples=npf(smp), log_prob=npf(prior.log_prob(xs)), log_prob_off=npf(prior.log_prob(off)),
        const_vel_mean=npf(prior.get_const_vel_mean(start, goal.unsqueeze(0), dt, H - 1, D)),
        const_vel_precision=npf(prior.get_const_vel_covariance(dt, sK, Qi, gK)),
        const_vel_covariance=npf(prior.get_const_vel_covariance(dt, sK, Qi, gK, precision_matrix=False)))
    print(name, 'samples', smp.shape)


@fixture
def gen_gp_prior_general(name, D, H, dt, seed):
    """MultiMPPrior with NON-isotropic start / GP / goal precisions (mp_priors_multi.py:213-251 takes arbitrary matrices):
    random SPD K_s_inv, K_gp_inv, K_g_inv; two goal modes."""
    ta = TA64
    g = torch.Generator().manual_seed(seed + 100)
    def spd(n, lo, hi):
        A = torch.randn(n, n, generator=g, dtype=torch.float64)
        Q, _ = torch.linalg.qr(A)
        ev = torch.exp(torch.linspace(np.log(lo), np.log(hi), n, dtype=torch.float64))
        return (Q * ev) @ Q.t()
    sd = 2 * D
    sK, gK = spd(sd, 1e2, 1e6), spd(sd, 1e3, 1e5)
    Qi = spd(sd, 1e-1, 1e3)
    start = torch.cat([torch.linspace(-0.5, 0.3, D), torch.zeros(D)]).to(**ta)
    goals = torch.stack([torch.cat([torch.linspace(0.4, -0.2, D), torch.zeros(D)]),
                         torch.cat([torch.linspace(-0.1, 0.6, D), torch.zeros(D)])]).to(**ta)
    torch.manual_seed(seed)
    with EpsRecorder() as rec:
        prior = MultiMPPrior(H - 1, dt, sd, D, sK, Qi, start, K_g_inv=gK, goal_states=goals, tensor_args=ta)
        smp = prior.sample(5)
    xs = smp.transpose(0, 1).reshape(5, 2, -1)
    off = xs + 0.05 * torch.linspace(-1, 1, xs.shape[-1], dtype=torch.float64)
    # set_Sigma_invs with a DIFFERENT precision per mode (mp_priors_multi.py:124-128), then a draw and the density
    Sinv2 = torch.stack([prior.Sigma_inv, 2.5 * prior.Sigma_inv])
    torch.manual_seed(seed + 7)
    with EpsRecorder() as rec2:
        prior.set_Sigma_invs(Sinv2)
        smp2 = prior.sample(4)
    lp2 = prior.log_prob(xs)
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), planner='gp_prior_general', D=D, H=H, dt=dt, start=npf(start), goals=npf(goals),
        K_s_inv=npf(sK), K_gp_inv=npf(Qi), K_g_inv=npf(gK), Sigma_inv=npf(prior.Sigma_inv), mean=npf(prior.means),
        scale_tril=npf(prior.dist._unbroadcasted_scale_tril), eps=npf(rec.draws[0]), samples=npf(smp),
        log_prob=npf(MultiMPPrior(H - 1, dt, sd, D, sK, Qi, start, K_g_inv=gK, goal_states=goals, tensor_args=ta).log_prob(xs)),
        log_prob_off=npf(MultiMPPrior(H - 1, dt, sd, D, sK, Qi, start, K_g_inv=gK, goal_states=goals, tensor_args=ta).log_prob(off)),
        const_vel_mean=npf(prior.get_const_vel_mean(start, goals, dt, H - 1, D)),
        Sigma_invs2=npf(Sinv2), eps2=npf(rec2.draws[0]), samples2=npf(smp2), log_prob2=npf(lp2))
    print(name, 'samples', smp.shape)


@fixture
def gen_cost_terms(name, robot, D, H, G_, npg, S, dt, seed):
    """The trajectory-only cost classes of cost_functions.py (CostGP :234-314, CostGPTrajectory :317-357,
    the position-only wrapper :360-368, CostSmoothnessCHOMP :371-390, CostJointLimits :393-429,
    CostGoalPrior :488-536) evaluated by the reference on one batch, in fp32 and fp64."""
    g = torch.Generator().manual_seed(seed)
    B = G_ * npg * S
    out = {}
    qmin, qmax = torch.as_tensor(robot.q_min_np), torch.as_tensor(robot.q_max_np)
    span = qmax - qmin
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    q0 = qmin + span * torch.rand(B, 1, D, generator=g)
    q1 = qmin + span * (torch.rand(B, 1, D, generator=g) * 1.1 - 0.05)      # a few ends beyond the limits
    pos = q0 * (1 - a) + q1 * a + 0.02 * span * torch.randn(B, H, D, generator=g)
    vel = (q1 - q0) / ((H - 1) * dt) + 0.05 * torch.randn(B, H, D, generator=g)
    trajs32 = torch.cat([pos, vel.expand(B, H, D)], -1).float().contiguous()
    start = torch.cat([q0[0, 0], torch.zeros(D)]).float()
    goals = torch.cat([q1[::npg * S, 0][:G_], torch.zeros(G_, D)], -1).float()
    sig = dict(sigma_start=1e-2, sigma_gp=0.5)
    for tag, ta in (('f32', TA32), ('f64', TA64)):
        rr, _ = make_ref_geometry(robot, G.CollisionField(spheres=np.array([[9., 9., 9., 0.1]], np.float32)), ta)
        rr.dt = dt
        x = trajs32.to(**ta)
        out['gp_' + tag] = npf(CostGP(rr, H, start.to(**ta), dt, sig, tensor_args=ta).eval(x))
        out['gptraj_' + tag] = npf(CostGPTrajectory(rr, H, dt, sigma_gp=0.5, tensor_args=ta).eval(x))
        out['gptraj_posonly_' + tag] = npf(
            CostGPTrajectoryPositionOnlyWrapper(rr, H, dt, sigma_gp=0.5, tensor_args=ta).eval(x[..., :D]))
        out['smooth_' + tag] = npf(CostSmoothnessCHOMP(rr, H, tensor_args=ta).eval(x))           # (B, d): see test
        out['jlim_' + tag] = npf(CostJointLimits(rr, H, eps=float(np.deg2rad(3)), tensor_args=ta).eval(x))
        out['goalprior_' + tag] = npf(CostGoalPrior(rr, H, multi_goal_states=goals.to(**ta), num_particles_per_goal=npg,
                                                    num_samples=S, sigma_goal_prior=0.1, tensor_args=ta).eval(x))
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'), planner='cost_terms', D=D, H=H, B=B, G=G_, npg=npg, S=S, dt=dt,
        trajs=npf(trajs32), start=npf(start), goals=npf(goals), sigma_start=1e-2, sigma_gp=0.5, sigma_goal_prior=0.1,
        jl_eps=float(np.deg2rad(3)), q_min=robot.q_min_np, q_max=robot.q_max_np, **out)
    print(name, {k: (v.shape, float(np.sum(v))) for k, v in out.items() if k.endswith('f64')})


def main():
    torch.set_num_threads(4)
    pm = G.RobotPointMass(2, radius=0.01)
    grid = G.env_grid_circles_2d()
    dense = G.env_dense_2d()
    panda = G.RobotPanda()
    sph3 = G.env_spheres_3d()
    s2, g2 = torch.tensor([-0.8, -0.8]), torch.tensor([0.8, 0.8])

    # STOMP: point mass, stiff (C1 parameters) and benign; Panda, both regimes, pos_only both ways
    gen_stomp('stomp_pm2d_stiff', pm, grid, s2, g2, P=4, S=8, H=64, dt=0.04, sigma_coll=1e-3,
              pos_only=False, iters=6, seed=0)
    gen_stomp('stomp_pm2d_benign', pm, grid, s2, g2, P=4, S=8, H=64, dt=0.04, sigma_coll=1.0,
              pos_only=False, iters=6, seed=1)
    gen_stomp('stomp_pm2d_c1', pm, grid, s2, g2, P=4, S=4, H=64, dt=0.04, sigma_coll=1e-3,
              pos_only=False, iters=20, seed=0)
    q = free_configs(panda, sph3, 2, 5, TA32)
    gen_stomp('stomp_panda_stiff', panda, sph3, q[0], q[1], P=4, S=8, H=64, dt=5 / 64, sigma_coll=1e-3,
              pos_only=False, iters=4, seed=2)
    gen_stomp('stomp_panda_benign', panda, sph3, q[0], q[1], P=4, S=8, H=64, dt=5 / 64, sigma_coll=1.0,
              pos_only=True, iters=4, seed=3, temperature=0.1, sigma_spectral=0.8)
    gen_stomp('stomp_panda_t1', panda, sph3, q[0], q[1], P=4, S=8, H=64, dt=5 / 64, sigma_coll=1.0,
              pos_only=False, iters=6, seed=6, temperature=1.0, sigma_spectral=0.5)
    gen_stomp('stomp_pm2d_h48', pm, dense, s2, g2, P=3, S=5, H=48, dt=0.05, sigma_coll=0.1,
              pos_only=True, iters=3, seed=4, init_noise=0.01)
    # C3's sample count (S = 32: the update kernel's register path is full) and S = 64 (its tail loop); H = 32 pos_only
    # with S = 64: H*d = 224 is not a multiple of 64 (partial last worker wave of the update kernel)
    gen_stomp('stomp_panda_s32', panda, sph3, q[0], q[1], P=2, S=32, H=64, dt=5 / 64, sigma_coll=1.0,
              pos_only=False, iters=2, seed=7, temperature=1.0, sigma_spectral=0.5)
    gen_stomp('stomp_panda_s64', panda, sph3, q[0], q[1], P=2, S=64, H=64, dt=5 / 64, sigma_coll=1.0,
              pos_only=False, iters=1, seed=8, temperature=1.0, sigma_spectral=0.5)
    gen_stomp('stomp_panda_h32_s64', panda, sph3, q[0], q[1], P=2, S=64, H=32, dt=5 / 32, sigma_coll=1.0,
              pos_only=True, iters=2, seed=9, temperature=1.0, sigma_spectral=0.1)
    # H = 128 (two 64-waypoint chunks per rollout), S = 32, d = 14: the shape of BASELINE config 4's horizon on STOMP
    gen_stomp('stomp_panda_h128_s32', panda, sph3, q[0], q[1], P=1, S=32, H=128, dt=5 / 128, sigma_coll=1.0,
              pos_only=False, iters=2, seed=10, temperature=1.0, sigma_spectral=0.5)

    # CHOMP: dense 2-D with boxes (C2 parameters), Panda
    gs = torch.Generator().manual_seed(7)
    starts = torch.rand(8, 2, generator=gs) * 1.9 - 0.95
    goals = torch.rand(8, 2, generator=gs) * 1.9 - 0.95
    gen_chomp('chomp_pm2d_dense', pm, dense, starts, goals, H=64, dt=0.04, sigma_coll=1.0, weight=10.0,
              iters=8, seed=3)
    gen_chomp('chomp_pm2d_soft', pm, dense, starts, goals, H=64, dt=0.04, sigma_coll=1.0, weight=10.0,
              iters=8, seed=3, w_prior=1e-9, clip=10.0, lr=1e-3)
    qs = free_configs(panda, sph3, 8, 11, TA32)
    gen_chomp('chomp_panda', panda, sph3, qs[:4], qs[4:], H=64, dt=5 / 64, sigma_coll=1.0, weight=10.0,
              iters=6, seed=5, w_prior=1e-9, clip=10.0, lr=1e-3)

    # GPMP2: tiny dense case with full A,b,K (fp32 + fp64), Panda banded (fp64)
    gen_gpmp2('gpmp2_pm2d_h8_f64', pm, dense, s2 * 0.5, g2 * 0.5, B=3, H=8, dt=0.04, iters=3, seed=0, ta=TA64)
    gen_gpmp2('gpmp2_pm2d_h8_f32', pm, dense, s2 * 0.5, g2 * 0.5, B=3, H=8, dt=0.04, iters=3, seed=0, ta=TA32)
    gen_gpmp2('gpmp2_pm2d_h8_notr_f64', pm, dense, s2 * 0.5, g2 * 0.5, B=3, H=8, dt=0.04, iters=3, seed=0, ta=TA64,
              trust_region=False)
    gen_gpmp2('gpmp2_panda_h16_f64', panda, sph3, q[0], q[1], B=2, H=16, dt=5 / 16, iters=3, seed=1, ta=TA64)
    qc = free_configs(panda, sph3, 12, 11, TA32)                                # pair (8, 9): the line crosses obstacles
    # C4's shape per particle (H = 128, D = 7: dense N = 1792) and H = 64, fp64 reference; means / costs / g only
    gen_gpmp2('gpmp2_panda_h64_f64', panda, sph3, q[0], q[1], B=2, H=64, dt=5 / 64, iters=2, seed=2, ta=TA64,
              keep_band=False)
    gen_gpmp2('gpmp2_panda_h128_f64', panda, sph3, qc[8], qc[9], B=2, H=128, dt=5 / 128, iters=2, seed=3, ta=TA64,
              keep_band=False)
    # two collision fields: one block of collision rows per field (gpmp2.py:70-78)
    gen_gpmp2('gpmp2_pm2d_h8_2fields_f64', pm, dense, s2 * 0.5, g2 * 0.5, B=3, H=8, dt=0.04, iters=3, seed=0, ta=TA64,
              extra_fields=(G.env_grid_circles_2d(margin=0.03),))
    # with n_interpolated_points: collision Jacobian of the interpolated trajectory (build-defined interpolation)
    gen_gpmp2('gpmp2_pm2d_h8_interp_f64', pm, dense, s2 * 0.5, g2 * 0.5, B=3, H=8, dt=0.04, iters=3, seed=0, ta=TA64,
              n_interp=3)
    gen_gpmp2('gpmp2_panda_h16_interp_f64', panda, sph3, qc[8], qc[9], B=2, H=16, dt=5 / 16, iters=3, seed=1, ta=TA64,
              n_interp=2)

    # StochGPMP (fp64 reference run: its fp32 dense scale_tril is not reproducible)
    gen_stoch_gpmp('sgpmp_pm2d_h16_f64', pm, dense, s2 * 0.5, g2 * 0.5, P=3, S=8, H=16, dt=0.08, iters=3, seed=0, ta=TA64)
    gen_stoch_gpmp('sgpmp_panda_h16_f64', panda, sph3, q[0], q[1], P=2, S=8, H=16, dt=5 / 16, iters=3, seed=1, ta=TA64,
                   sig_sample=(1e-3, 1e-3, 0.2))

    # MPPI
    gen_mppi('mppi_pm2d_const', S=32, T=64, dt=0.04, iters=5, seed=0)
    gen_mppi('mppi_pm2d_indep_cost', S=32, T=64, dt=0.04, iters=3, seed=1, cov_type='indep_ctrl', with_cost=True)

    # GP-prior initial sampling
    gen_gp_prior('gp_prior_d2_h8', D=2, H=8, dt=0.04, seed=0)
    gen_gp_prior_general('gp_prior_general_d2_h6', D=2, H=6, dt=0.05, seed=2)

    # trajectory-only cost classes
    gen_cost_terms('cost_terms_pm2d', pm, D=2, H=64, G_=2, npg=3, S=4, dt=0.04, seed=0)
    gen_cost_terms('cost_terms_panda', panda, D=7, H=48, G_=1, npg=2, S=5, dt=0.1, seed=1)


if __name__ == '__main__':
    main()
