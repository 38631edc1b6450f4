"""CPU: the oracle restatement (oracle/planners_ref.py) against golden vectors produced by the
unmodified reference classes (tests/golden/make_goldens.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, ref_geometry_from_golden
from oracle import planners_ref as O

TA = dict(device='cpu', dtype=torch.float32)
T = torch.from_numpy

STOMP_CASES = ['stomp_pm2d_stiff', 'stomp_pm2d_benign', 'stomp_pm2d_c1', 'stomp_panda_stiff',
               'stomp_panda_benign', 'stomp_panda_t1', 'stomp_pm2d_h48', 'stomp_panda_s32', 'stomp_panda_s64',
               'stomp_panda_h32_s64', 'stomp_panda_h128_s32']


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_constants(name):
    g = load_golden(name)
    R, Sigma, L = O.stomp_constants(int(g['H']), float(g['dt']), float(g['sigma_spectral']), TA)
    assert torch.equal(R, T(g['R']))
    assert torch.equal(Sigma, T(g['Sigma']))
    assert torch.equal(L, T(g['L']))


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_iterations(name):
    g = load_golden(name)
    robot, field = ref_geometry_from_golden(g)
    L, Sigma = T(g['L']), T(g['Sigma'])
    means = T(g['means0'])
    cost_fn = lambda x: O.collision_cost(x, robot, field, float(g['sigma_coll']))
    for it in range(g['eps'].shape[0]):
        out = O.stomp_iteration(means, T(g['eps'][it]), L, Sigma, cost_fn, float(g['lr']), float(g['temperature']))
        # teacher-forced per iteration: identical inputs -> identical torch ops -> bit-exact
        assert torch.equal(out['samples'], T(g['samples'][it])), it
        assert torch.equal(out['costs'], T(g['costs'][it])), it
        assert torch.equal(out['weights'], T(g['weights'][it])), it
        assert torch.equal(out['means'], T(g['means'][it])), it
        means = out['means']


@pytest.mark.parametrize('name', ['chomp_pm2d_dense', 'chomp_pm2d_soft', 'chomp_panda'])
def test_chomp_iterations(name):
    g = load_golden(name)
    robot, field = ref_geometry_from_golden(g)
    R = O.chomp_precision(int(g['H']), float(g['dt']), TA)
    assert torch.equal(R, T(g['R']))
    cost_fn = lambda x: O.collision_cost(x, robot, field, float(g['sigma_coll']), weight=float(g['weight']))
    means = T(g['means0'])
    for it in range(g['means'].shape[0]):
        out = O.chomp_iteration(means, R, cost_fn, float(g['w_prior']), float(g['lr']), float(g['clip']))
        np.testing.assert_allclose(out['means'].numpy(), g['means'][it], rtol=0, atol=1e-7)
        means = out['means']


@pytest.mark.parametrize('name', ['gpmp2_pm2d_h8_f64', 'gpmp2_pm2d_h8_f32', 'gpmp2_pm2d_h8_notr_f64',
                                  'gpmp2_panda_h16_f64', 'gpmp2_pm2d_h8_interp_f64', 'gpmp2_panda_h16_interp_f64',
                                  'gpmp2_pm2d_h8_2fields_f64', 'gpmp2_panda_h64_f64', 'gpmp2_panda_h128_f64'])
def test_gpmp2_iterations(name):
    g = load_golden(name)
    dt64 = 'float64' in str(g['dtype'])
    dtype = torch.float64 if dt64 else torch.float32
    ta = dict(device='cpu', dtype=dtype)
    robot, field = ref_geometry_from_golden(g, dtype)
    D, H = int(g['D']), int(g['H'])
    start = torch.cat([T(g['start']), torch.zeros(D, dtype=dtype)]).to(**ta)
    goal = torch.cat([T(g['goal']), torch.zeros(D, dtype=dtype)]).to(**ta)
    x = T(g['means0']).to(**ta)
    for it in range(g['means'].shape[0]):
        kw = dict(D=D, dt=float(g['dt']), sigma_start=float(g['sigma_start']), sigma_gp=float(g['sigma_gp']),
                  sigma_goal=float(g['sigma_goal_prior']), sigma_coll=float(g['sigma_coll']), tensor_args=ta,
                  n_interp=int(g['n_interp']) if 'n_interp' in g else None)
        if 'A' in g:
            A, b, K = O.gpmp2_linear_system(x, robot, field, start, goal, **kw)
            np.testing.assert_allclose(A.numpy(), g['A'][it], rtol=1e-6 if dt64 else 1e-4, atol=1e-9 if dt64 else 1e-5)
            np.testing.assert_allclose(b.numpy(), g['b'][it], rtol=1e-6 if dt64 else 1e-4, atol=1e-9 if dt64 else 1e-5)
            np.testing.assert_allclose(torch.diagonal(K, dim1=-2, dim2=-1).numpy(), g['K'][it], rtol=1e-6)
        out = O.gpmp2_iteration(x, robot, field, start, goal, delta=float(g['delta']),
                                trust_region=bool(g['trust_region']), step_size=float(g['step_size']), **kw)
        if dt64:
            np.testing.assert_allclose(out['g'].numpy(), g['g'][it], rtol=1e-9, atol=1e-6)
            np.testing.assert_allclose(out['means'].numpy(), g['means'][it], rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(out['costs'].numpy(), g['costs'][it], rtol=1e-9)
        if 'JtJ_outside_band_absmax' in g:
            assert float(g['JtJ_outside_band_absmax']) == 0.0   # block-tridiagonal structure of the reference system
        # teacher-force with the reference's own iterate (fp32 dense Cholesky at kappa ~ 1e10+ is not reproducible)
        x = T(g['means'][it]).to(**ta)


@pytest.mark.parametrize('name', ['mppi_pm2d_const', 'mppi_pm2d_indep_cost'])
def test_mppi_iterations(name):
    g = load_golden(name)
    S, Tn = int(g['S']), int(g['T'])
    cov = O.mppi_covariance([float(v) for v in g['control_std']], Tn, 2, str(g['cov_type']), TA)
    assert torch.equal(cov, T(g['Cov']))
    robot, field = ref_geometry_from_golden(g)
    cw = dict(pos=float(g['c_pos']), vel=float(g['c_vel']), ctrl=float(g['c_ctrl']), pos_T=float(g['c_pos_T']))
    mean = torch.zeros(Tn, 2)
    disc = torch.ones(Tn)
    for it in range(g['eps'].shape[0]):
        shift = 0.0
        out = O.mppi_iteration(mean, T(g['eps'][it]), T(g['scale_tril']), T(g['Cov_inv']), T(g['start']), T(g['goal']),
                               float(g['dt']), torch.tensor([-100., -100.]), torch.tensor([100., 100.]), cw, disc,
                               float(g['temp']), float(g['step_size']), 2, shift_cost=shift)
        if bool(g['with_cost']):   # Q6: the collision cost collapses to one scalar added to every sample
            full = torch.cat((out['states'], out['controls']), -1)
            shift = O.collision_cost(full, robot, field, 1e-3).sum(-1)
            out = O.mppi_iteration(mean, T(g['eps'][it]), T(g['scale_tril']), T(g['Cov_inv']), T(g['start']),
                                   T(g['goal']), float(g['dt']), torch.tensor([-100., -100.]),
                                   torch.tensor([100., 100.]), cw, disc, float(g['temp']), float(g['step_size']), 2,
                                   shift_cost=shift)
        np.testing.assert_allclose(out['controls'].numpy(), g['controls'][it], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['states'].numpy(), g['states'][it], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['costs'].numpy(), g['costs'][it], rtol=2e-6)
        np.testing.assert_allclose(out['weights'].numpy().reshape(-1), g['weights'][it].reshape(-1), rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(out['mean'].numpy(), g['mean'][it], rtol=1e-4, atol=1e-5)
        mean = T(g['mean'][it])


def test_gp_prior():
    g = load_golden('gp_prior_d2_h8')
    D, H = int(g['D']), int(g['H'])
    ta = dict(device='cpu', dtype=torch.float64)
    Kinv = O.gp_prior_precision(H, float(g['dt']), D, float(g['sigma_start']), float(g['sigma_gp']), float(g['sigma_goal']))
    np.testing.assert_allclose(Kinv.numpy(), g['Sigma_inv'], rtol=1e-12)
    mean = O.gp_prior_mean(T(g['start']), T(g['goal']), H, float(g['dt']), D, ta)
    np.testing.assert_allclose(mean.reshape(-1).numpy(), g['mean'].reshape(-1), rtol=1e-12, atol=1e-15)
    L = O.precision_to_scale_tril(Kinv)
    smp = mean.reshape(1, -1) + (L @ T(g['eps']).unsqueeze(-1)).squeeze(-1)
    np.testing.assert_allclose(smp.reshape(g['samples'].shape[1], g['samples'].shape[0], H, 2 * D).transpose(0, 1).numpy(),
                               g['samples'], rtol=1e-9, atol=1e-12)


def test_gp_prior_general_precisions():
    """MultiMPPrior with non-isotropic K_s_inv / K_gp_inv / K_g_inv, two goal modes: precision, scale_tril, samples."""
    g = load_golden('gp_prior_general_d2_h6')
    D, H = int(g['D']), int(g['H'])
    Kinv = O.gp_prior_precision_general(H, float(g['dt']), D, g['K_s_inv'], g['K_gp_inv'], g['K_g_inv'])
    np.testing.assert_allclose(Kinv.numpy(), g['Sigma_inv'], rtol=1e-12, atol=1e-9 * np.abs(g['Sigma_inv']).max())
    L = O.precision_to_scale_tril(Kinv)
    np.testing.assert_allclose(L.numpy(), g['scale_tril'][0], rtol=1e-7, atol=1e-12)
    mean = T(g['mean'])                                             # (modes, M)
    eps = T(g['eps'])                                               # (n, modes, M)
    smp = mean.unsqueeze(0) + torch.einsum('ij,nmj->nmi', L, eps)
    want = T(g['samples'])                                          # (modes, n, H, 2D)
    np.testing.assert_allclose(smp.transpose(0, 1).reshape(want.shape).numpy(), want.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', ['sgpmp_pm2d_h16_f64', 'sgpmp_panda_h16_f64'])
def test_stoch_gpmp_iterations(name):
    g = load_golden(name)
    dtype = torch.float64
    ta = dict(device='cpu', dtype=dtype)
    robot, field = ref_geometry_from_golden(g, dtype)
    D, H = int(g['D']), int(g['H'])
    start = torch.cat([T(g['start']), torch.zeros(D, dtype=dtype)])
    goal = torch.cat([T(g['goal']), torch.zeros(D, dtype=dtype)])
    Kinv = O.gp_prior_precision(H, float(g['dt']), D, float(g['sigma_start_sample']), float(g['sigma_gp_sample']),
                                float(g['sigma_goal_sample']))
    np.testing.assert_allclose(Kinv.numpy(), g['Sigma_inv'], rtol=1e-12)
    L = O.precision_to_scale_tril(Kinv)

    def cost_fn(x):
        return (O.cost_gp_eval(x, start, D, float(g['dt']), float(g['sigma_start']), float(g['sigma_gp']), ta)
                + O.goal_prior_eval(x, goal, float(g['sigma_goal_prior']))
                + O.collision_cost(x, robot, field, float(g['sigma_coll'])))
    means = T(g['means0'])
    for it in range(g['eps'].shape[0]):
        out = O.stoch_gpmp_iteration(means, T(g['eps'][it]), L, Kinv, cost_fn, float(g['temperature']), float(g['step_size']))
        np.testing.assert_allclose(out['samples'].numpy(), g['samples'][it], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(out['costs'].numpy(), g['costs'][it], rtol=1e-9)
        np.testing.assert_allclose(out['weights'].numpy(), g['weights'][it], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(out['means'].numpy(), g['means'][it], rtol=1e-7, atol=1e-9)
        means = T(g['means'][it])


@pytest.mark.parametrize('name', ['cost_terms_pm2d', 'cost_terms_panda'])
@pytest.mark.parametrize('tag', ['f32', 'f64'])
def test_cost_terms(name, tag):
    """Trajectory-only cost classes (CostGP, CostGPTrajectory, position-only wrapper, CostSmoothnessCHOMP,
    CostJointLimits, CostGoalPrior) as the reference evaluates them."""
    g = load_golden(name)
    ta = dict(device='cpu', dtype=torch.float32 if tag == 'f32' else torch.float64)
    x = T(g['trajs']).to(**ta)
    D, dt = int(g['D']), float(g['dt'])
    rtol = 2e-5 if tag == 'f32' else 1e-12
    ck = lambda got, key: np.testing.assert_allclose(got.numpy(), g[key + '_' + tag], rtol=rtol, atol=0)
    ck(O.cost_gp_eval(x, T(g['start']).to(**ta), D, dt, float(g['sigma_start']), float(g['sigma_gp']), ta), 'gp')
    ck(O.cost_gp_trajectory_eval(x, D, dt, float(g['sigma_gp']), ta), 'gptraj')
    ck(O.cost_gp_trajectory_pos_only_eval(x[..., :D], D, dt, float(g['sigma_gp']), ta), 'gptraj_posonly')
    tot, per_col = O.cost_smoothness_chomp_eval(x, dt, ta)
    ck(per_col, 'smooth')
    jl = O.cost_joint_limits_eval(x, D, T(g['q_min']).to(**ta), T(g['q_max']).to(**ta), float(g['jl_eps']))
    assert g['jlim_' + tag].shape == ()                       # the reference's batch-global scalar
    ck(jl, 'jlim')
    ck(O.cost_goal_prior_multi_eval(x, T(g['goals']).to(**ta), int(g['npg']) * int(g['S']),
                                    float(g['sigma_goal_prior'])), 'goalprior')
