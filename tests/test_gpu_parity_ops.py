"""GPU parity tests proper: the HIP path (through the C-ABI) against the oracle and the golden vectors
generated from the unmodified reference.  Tolerance stated by north_star: 1e-4 relative fp32 on final
trajectory waypoints."""
import numpy as np
import pytest
import torch

from conftest import load_golden, product_geometry_from_golden, ref_geometry_from_golden

pytestmark = pytest.mark.gpu

T = torch.from_numpy
REL = 1e-4


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def dev_geom(g, device):
    from motion_planning_baselines_amd.ops import DeviceGeometry
    robot, field = product_geometry_from_golden(g)
    return DeviceGeometry(robot, field, device)


def random_trajs(g, B, H, d, seed, spread):
    gen = torch.Generator().manual_seed(seed)
    D = int(g['n_dof'])
    if int(g['robot_kind']) == 0:
        x = (torch.rand(B, H, d, generator=gen) * 2 - 1) * spread
    else:
        from motion_planning_baselines_amd.geometry import _PANDA_Q_MIN, _PANDA_Q_MAX
        lo, hi = torch.tensor(_PANDA_Q_MIN), torch.tensor(_PANDA_Q_MAX)
        a = lo + (hi - lo) * torch.rand(B, 1, D, generator=gen)
        b = lo + (hi - lo) * torch.rand(B, 1, D, generator=gen)
        t = torch.linspace(0, 1, H).reshape(1, H, 1)
        q = a * (1 - t) + b * t + 0.05 * torch.randn(B, H, D, generator=gen)
        x = torch.cat([q, torch.randn(B, H, d - D, generator=gen)], -1) if d > D else q
    return x.contiguous()


@pytest.mark.parametrize('name,B,H,dd,sigma', [
    ('chomp_pm2d_dense', 37, 64, 4, 1.0), ('chomp_pm2d_dense', 5, 100, 2, 0.1),
    ('stomp_panda_benign', 33, 64, 7, 1.0), ('stomp_panda_stiff', 9, 64, 14, 1e-3),
    ('stomp_pm2d_c1', 16, 64, 4, 1e-3), ('stomp_panda_benign', 3, 150, 14, 1.0)])
def test_collision_cost_vs_oracle(gpu_device, name, B, H, dd, sigma):
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    g = load_golden(name)
    robot, field = ref_geometry_from_golden(g)
    x = random_trajs(g, B, H, dd, 0, 1.0)
    ref = O.collision_cost(x, robot, field, sigma, weight=3.0)
    ref_pw = field.compute_cost(None, robot.fk_map_collision(robot.get_position(x)))
    out, pw = ops.cost_collision_eval(x.to(gpu_device), dev_geom(g, gpu_device), 1.0 / sigma ** 2, weight=3.0,
                                      per_waypoint=True)
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0, 'test inputs must actually collide'
    np.testing.assert_allclose(pw.cpu().numpy()[:, 1:], ref_pw.numpy()[:, 1:], rtol=2e-5, atol=2e-6)
    assert (pw[:, 0] == 0).all()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=1e-6 * 3.0 / sigma ** 2)


@pytest.mark.parametrize('name,B,H,dd', [
    ('chomp_pm2d_dense', 64, 64, 4), ('chomp_panda', 16, 64, 14), ('chomp_panda', 7, 33, 7)])
def test_collision_grad_vs_autograd(gpu_device, name, B, H, dd):
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    g = load_golden(name)
    robot, field = ref_geometry_from_golden(g)
    x = random_trajs(g, B, H, dd, 1, 1.0).requires_grad_(True)
    ref = O.collision_cost(x, robot, field, 1.0, weight=10.0)
    ref.sum().backward()
    out, grad = ops.cost_collision_grad(x.detach().to(gpu_device), dev_geom(g, gpu_device), 1.0, weight=10.0)
    torch.cuda.synchronize()
    assert float(x.grad.abs().max()) > 0
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-5)
    # hinge / min kinks: an element may sit on the other side of a kink after fp32 re-ordering
    diff = (grad.cpu() - x.grad).abs()
    tol = 1e-4 * x.grad.abs().max() + 1e-4 * x.grad.abs()
    frac_bad = float((diff > tol).float().mean())
    assert frac_bad < 2e-3, frac_bad


STOMP_CASES = ['stomp_pm2d_stiff', 'stomp_pm2d_benign', 'stomp_pm2d_c1', 'stomp_panda_stiff',
               'stomp_panda_benign', 'stomp_panda_t1', 'stomp_pm2d_h48',
               'stomp_panda_s32',       # C3's S = 32: every register slot of the update kernel in use
               'stomp_panda_s64',       # S = 64: the update kernel's tail loop
               'stomp_panda_h32_s64',   # H*d = 224: partial last worker wave in the update kernel, chunked sampler
               'stomp_panda_h128_s32']  # H = 128: two horizon chunks per rollout (persistent: mpb_stomp_fused_hx.hip)


def reference_fp32_envelope(g):
    """|reference fp32 golden - oracle run in fp64 on the same injected noise| / |.|, final means."""
    from oracle import planners_ref as O
    robot, field = ref_geometry_from_golden(g, torch.float64)
    L, Sigma = T(g['L']).double(), T(g['Sigma']).double()
    m = T(g['means0']).double()
    for it in range(g['eps'].shape[0]):
        m = O.stomp_iteration(m, T(g['eps'][it]).double(), L, Sigma,
                              lambda x: O.collision_cost(x, robot, field, float(g['sigma_coll'])),
                              float(g['lr']), float(g['temperature']))['means']
    return rel_err(T(g['means'][-1]), m)


def reference_fp32_envelope_one_iteration(g, it):
    """The same envelope for ONE teacher-forced pass of the loop body from the reference's own iterate."""
    from oracle import planners_ref as O
    robot, field = ref_geometry_from_golden(g, torch.float64)
    prev = T(g['means0'] if it == 0 else g['means'][it - 1]).double()
    m = O.stomp_iteration(prev, T(g['eps'][it]).double(), T(g['L']).double(), T(g['Sigma']).double(),
                          lambda x: O.collision_cost(x, robot, field, float(g['sigma_coll'])),
                          float(g['lr']), float(g['temperature']))['means']
    return rel_err(T(g['means'][it]), m)


def _stomp_bufs(g, dev):
    P, S, H = int(g['P']), int(g['S']), int(g['H'])
    d = g['means0'].shape[-1]
    mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    return P, S, H, d, mk(P, S, H, d), mk(P, S), mk(P, S)


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_teacher_forced_vs_golden(gpu_device, name):
    """Each iteration starts from the reference's own means: isolates one pass of the loop body."""
    from motion_planning_baselines_amd import ops
    g = load_golden(name)
    dev = gpu_device
    P, S, H, d, samples, costs, weights = _stomp_bufs(g, dev)
    geom = dev_geom(g, dev)
    L, Sigma = T(g['L']).to(dev), T(g['Sigma']).to(dev)
    ksig = 1.0 / float(g['sigma_coll']) ** 2
    prev = T(g['means0'])
    for it in range(g['eps'].shape[0]):
        means = prev.clone().to(dev)
        eps = T(g['eps'][it:it + 1]).contiguous().to(dev)
        ops.stomp_step(means, eps, samples, costs, weights, L, Sigma, geom, S, int(g['D']), ksig, 1.0,
                       float(g['lr']), float(g['temperature']))
        torch.cuda.synchronize()
        assert rel_err(samples, T(g['samples'][it])) < 2e-5, it
        ref_c = T(g['costs'][it])
        np.testing.assert_allclose(costs.cpu().numpy(), ref_c.numpy(), rtol=5e-5, atol=1e-6 * ksig)
        # weights: softmax of costs that carry a 1/sigma^2 factor -- compare where it is well conditioned
        wref = T(g['weights'][it])
        cond = float(ksig) * 1e-6
        if cond < 1e-2:
            np.testing.assert_allclose(weights.cpu().numpy(), wref.numpy(), rtol=1e-3, atol=1e-5)
        # means: north_star's 1e-4; where the softmax is ill-conditioned in the reference itself (costs carry a
        # 1/sigma^2 = 1e6 factor: an fp32 rounding of the cost sum moves the logits by O(0.1)) the reference's fp32
        # result is only defined up to its own fp32-vs-fp64 envelope, computed here on the same noise
        bar = REL if cond < 1e-2 else max(REL, 2.0 * reference_fp32_envelope_one_iteration(g, it))
        assert rel_err(means, T(g['means'][it])) < bar, (it, bar)
        prev = T(g['means'][it])


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_free_running_vs_golden(gpu_device, name):
    """All iterations in ONE C-ABI call from means0 with the reference's injected noise: the
    north_star criterion (final waypoints within 1e-4 relative)."""
    from motion_planning_baselines_amd import ops
    g = load_golden(name)
    dev = gpu_device
    P, S, H, d, samples, costs, weights = _stomp_bufs(g, dev)
    n = g['eps'].shape[0]
    means = T(g['means0']).clone().to(dev)
    ops.stomp_step(means, T(g['eps']).contiguous().to(dev), samples, costs, weights, T(g['L']).to(dev),
                   T(g['Sigma']).to(dev), dev_geom(g, dev), S, int(g['D']), 1.0 / float(g['sigma_coll']) ** 2, 1.0,
                   float(g['lr']), float(g['temperature']), n_iters=n)
    torch.cuda.synchronize()
    err = rel_err(means, T(g['means'][-1]))
    # The reference's own fp32 result is only defined up to its rounding envelope: the same algorithm in
    # fp64 on the same noise.  For softmax logits c/T >> 1 (stomp_panda_benign: c/T ~ 126) that envelope
    # exceeds 1e-4; the bar is then twice the envelope, otherwise the north_star's 1e-4.
    env = reference_fp32_envelope(g)
    print(name, 'final-waypoint rel err', err, 'reference fp32-vs-fp64 envelope', env)
    assert err < max(REL, 2.0 * env)
    if name != 'stomp_panda_benign':
        assert err < REL


@pytest.mark.parametrize('P,S,H,d', [
    (3, 64, 32, 7),     # Panda pos_only H = 32: H*d = 224 = 3.5 waves, two samples per group beyond the registers
    (2, 48, 16, 14),    # StochGPMP Panda H = 16 (dim 14)
    (2, 40, 48, 14),    # H*d = 672 = 10.5 waves
    (5, 24, 8, 2),      # H*d = 16: a quarter wave of workers
    (2, 64, 64, 14),    # full waves, S = 64 (tail loop), C3's tile
    (4, 32, 64, 14),    # C3's shape
    (3, 33, 64, 7), (2, 9, 20, 3), (2, 70, 64, 4), (2, 5, 100, 4)])   # generic kernel: odd sizes / S > 64 / H > 64
@pytest.mark.parametrize('with_sigma', [True, False])
def test_stomp_update_vs_oracle_shapes(gpu_device, P, S, H, d, with_sigma):
    """Kernel B alone (softmax weights + covariance-weighted update, stomp.py:199-220) against the oracle on random
    samples / costs, over shapes that hit every dispatch path: full and partial worker waves, S <= 32, 32 < S <= 64
    (tail loop), S > 64 and H > 64 (generic kernel), and the update without the covariance product (StochGPMP)."""
    from motion_planning_baselines_amd import ops
    from oracle import planners_ref as O
    dev = gpu_device
    gen = torch.Generator().manual_seed(100 * S + H + d)
    means = torch.randn(P, H, d, generator=gen)
    samples = means.unsqueeze(1) + 0.3 * torch.randn(P, S, H, d, generator=gen)
    costs = 5.0 * torch.rand(P, S, generator=gen)
    A = torch.randn(H, H, generator=gen) / H ** 0.5
    Sigma = (A @ A.T + 0.1 * torch.eye(H)).contiguous()
    lr, temp = 0.3, 0.7
    w_ref = O.stomp_weights(costs.double(), temp)
    if with_sigma:
        m_ref = O.stomp_update(means.double(), samples.double(), w_ref, Sigma.double(), lr)
    else:   # stoch_gpmp.py:272-275: no covariance product
        m_ref = means.double() + lr * (w_ref.reshape(P, S, 1, 1) * (samples.double() - means.double().unsqueeze(1))).sum(1)
    m = means.clone().to(dev)
    w = torch.empty(P, S, device=dev)
    ops.stomp_update(m, samples.to(dev), costs.to(dev), w, Sigma.to(dev) if with_sigma else None, lr, temp)
    torch.cuda.synchronize()
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.float().numpy(), rtol=2e-5, atol=1e-8)
    assert abs(float(w.sum(1).mean()) - 1.0) < 1e-5
    assert rel_err(m, m_ref) < 2e-6


def test_stomp_split_halves_equal_fused(gpu_device):
    """sample -> (cost) -> update through the split entry points == fused step (drop-in for a user cost)."""
    from motion_planning_baselines_amd import ops
    g = load_golden('stomp_panda_benign')
    dev = gpu_device
    P, S, H, d, samples, costs, weights = _stomp_bufs(g, dev)
    geom = dev_geom(g, dev)
    L, Sigma = T(g['L']).to(dev), T(g['Sigma']).to(dev)
    eps = T(g['eps'][0:1]).contiguous().to(dev)
    m1 = T(g['means0']).clone().to(dev)
    ops.stomp_step(m1, eps, samples, costs, weights, L, Sigma, geom, S, int(g['D']), 1.0, 1.0, 0.1, 0.1)
    m2 = T(g['means0']).clone().to(dev)
    s2, w2 = torch.empty_like(samples), torch.empty_like(weights)
    ops.stomp_sample(m2, eps[0], s2, L, S)
    c2 = ops.cost_collision_eval(s2.flatten(0, 1), geom, 1.0).reshape(P, S)
    ops.stomp_update(m2, s2, c2, w2, Sigma, 0.1, 0.1)
    torch.cuda.synchronize()
    assert torch.equal(s2, samples) and torch.equal(c2, costs) and torch.equal(w2, weights)
    # both paths run the same update kernel on the same samples / costs
    assert torch.equal(m1, m2)


def test_stomp_device_rng(gpu_device):
    """Device Philox path: deterministic, independent of how particles are sharded, and distributed as
    N(0, Sigma) along the horizon."""
    from motion_planning_baselines_amd import ops
    g = load_golden('stomp_pm2d_benign')
    dev = gpu_device
    H, d, S, P = 64, 4, 64, 32
    L, Sigma = T(g['L']).to(dev), T(g['Sigma']).to(dev)
    means = torch.zeros(P, H, d, device=dev)
    a = torch.empty(P, S, H, d, device=dev)
    b = torch.empty_like(a)
    ops.stomp_sample(means, None, a, L, S, seed=1234, it=7)
    ops.stomp_sample(means, None, b, L, S, seed=1234, it=7)
    assert torch.equal(a, b)
    half = torch.empty(P // 2, S, H, d, device=dev)
    ops.stomp_sample(means[P // 2:].contiguous(), None, half, L, S, seed=1234, it=7, particle_offset=P // 2)
    assert torch.equal(half, a[P // 2:])
    ops.stomp_sample(means, None, b, L, S, seed=1234, it=8)
    assert not torch.equal(a, b)
    assert (a[:, :, 0] == 0).all() and (a[:, :, -1] == 0).all()
    x = a[:, :, 1:-1].permute(0, 1, 3, 2).reshape(-1, H - 2).double().cpu()   # (P*S*d, H-2) draws of N(0, Sigma)
    n = x.shape[0]
    assert abs(float(x.mean())) < 0.05 * float(x.std())
    emp = (x.T @ x) / n
    Sg = T(g['Sigma']).double()[1:-1, 1:-1]
    assert float((emp - Sg).abs().max() / Sg.abs().max()) < 0.06


def chomp_reference_fp32_envelope(g):
    """|reference fp32 golden - oracle in fp64| / |.| on the final means (free running)."""
    from oracle import planners_ref as O
    robot, field = ref_geometry_from_golden(g, torch.float64)
    R = O.chomp_precision(int(g['H']), float(g['dt']), dict(device='cpu', dtype=torch.float32)).double()
    m = T(g['means0']).double()
    for _ in range(g['means'].shape[0]):
        m = O.chomp_iteration(m, R, lambda x: O.collision_cost(x, robot, field, float(g['sigma_coll']),
                                                               weight=float(g['weight'])),
                              float(g['w_prior']), float(g['lr']), float(g['clip']))['means']
    return rel_err(T(g['means'][-1]), m)


@pytest.mark.parametrize('name', ['chomp_pm2d_dense', 'chomp_pm2d_soft', 'chomp_panda'])
def test_chomp_vs_golden(gpu_device, name):
    from motion_planning_baselines_amd import ops
    g = load_golden(name)
    dev = gpu_device
    geom = dev_geom(g, dev)
    R = T(g['R']).to(dev)
    n = g['means'].shape[0]
    kw = dict(D=int(g['D']), k_sigma=1.0 / float(g['sigma_coll']) ** 2, weight=float(g['weight']),
              w_prior=float(g['w_prior']), lr=float(g['lr']), grad_clip=float(g['clip']))
    # teacher forced: one pass of the loop body from the reference's own iterate -- strict
    prev = T(g['means0'])
    for it in range(n):
        means = prev.clone().to(dev)
        ops.chomp_step(means, R, geom, n_iters=1, **kw)
        torch.cuda.synchronize()
        assert rel_err(means, T(g['means'][it])) < 1e-5, it
        prev = T(g['means'][it])
    # free running, one iteration per call and the whole loop inside one launch: identical results
    means = T(g['means0']).clone().to(dev)
    for it in range(n):
        ops.chomp_step(means, R, geom, n_iters=1, **kw)
    m2 = T(g['means0']).clone().to(dev)
    costs = torch.empty(m2.shape[0], device=dev)
    ops.chomp_step(m2, R, geom, n_iters=n, costs_out=costs, **kw)
    torch.cuda.synchronize()
    assert torch.equal(m2, means)
    assert torch.isfinite(costs).all()
    # free running vs the reference: 1e-4, or the reference's own fp32 rounding envelope where the
    # clipped, B-scaled smoothness gradient (quirk Q3) amplifies rounding noise beyond that
    err = rel_err(m2, T(g['means'][-1]))
    env = chomp_reference_fp32_envelope(g)
    print(name, 'free-running final rel err', err, 'reference fp32-vs-fp64 envelope', env)
    assert err < max(REL, 2.0 * env)
