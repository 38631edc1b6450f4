"""-m gpu: the HIP path against the ORACLE at BASELINE's full sizes.

The goldens generated from the reference hold P <= 4 particles (<= 8 workgroups): the ticket / pool / partner machinery of
the persistent STOMP launch only becomes non-trivial at P >= 8.  The oracle restatement of the reference loop
(oracle/planners_ref.py: stomp.py:150-160, chomp.py:127-151) runs the full C3 batch in about a second, so here it does:

  * C3 (panda_spheres STOMP, P = 128 x S = 32, H = 64, d = 14, sigma_coll = 1e-3): two iterations on injected noise
    through mpb_stomp_run_checked on the persistent exchange layout -- teacher forced and free running;
  * the same at P = 256 (one workgroup per particle, two batches of 16 samples);
  * C2 (pointmass_dense_2d CHOMP, B = 1024, H = 64): 20 iterations in one launch.

Bars: samples 2e-5, costs 5e-5 (as in the golden tests); means 1e-4 in BOTH norms (global max and per-waypoint L2,
conftest.rel_err_waypoint) for every particle whose update is well conditioned in the reference itself.  At sigma_coll =
1e-3 the costs are ~1e6 and softmax(-c / T) is one-hot unless two samples tie to within fp32 rounding of the costs: a
particle counts as ILL-conditioned when the oracle's own fp64 costs, perturbed by the fp32 cost resolution, move one of its
weights by more than 1e-5 (first-order bound, every cost moved by a few ulps of its own value) -- for those the reference fp32 result is only defined up to its fp32-vs-fp64 envelope and the
bar is max(1e-4, 2 x envelope), as in the golden tests."""
import numpy as np
import pytest
import torch

from conftest import rel_err_waypoint

pytestmark = pytest.mark.gpu
REL = 1e-4


def _gmax(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _c3(dev, P, S=32, H=64):
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=False)
    prm = wl['params']
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, prm['dt'], prm['sigma_spectral'], cpu)
    return wl, torch.inverse(R).contiguous(), precision_to_scale_tril(R).contiguous(), ops.DeviceGeometry(wl['robot'], wl['field'], dev)


def _oracle_iter(wl, means, eps, L, Sigma, dtype):
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    robot, field = make_ref_geometry(wl['robot'], wl['field'], dict(device='cpu', dtype=dtype))
    prm = wl['params']
    return O.stomp_iteration(means.to(dtype), eps.to(dtype), L.to(dtype), Sigma.to(dtype),
                             lambda x: O.collision_cost(x, robot, field, wl['sigma_coll']), prm['step_size'], prm['temperature'])


def _ill_conditioned(costs64, temperature):
    """(P,) bool: particles whose softmax weights move by > 1e-5 when every cost moves by its own fp32 resolution (a few
    ulps of ITS value: samples that are collision free cost exactly 0 on both sides and tie exactly, which is harmless):
    first-order bound |dw_s| <= w_s (1 - w_s) dc_s + w_s sum_{j != s} w_j dc_j."""
    c = costs64.double()
    w = torch.softmax(-c / temperature, dim=1)
    dc = 8.0 * 2.0 ** -24 * c.abs() / temperature
    wd = w * dc
    dw = w * (1.0 - w) * dc + w * (wd.sum(dim=1, keepdim=True) - wd)
    return dw.max(dim=1).values > 1e-5


def _check_iteration(tag, got, ref32, ref64, n_pos, temperature):
    means, samples, costs, weights = got
    assert _gmax(samples, ref32['samples']) < 2e-5, tag
    ksig_atol = 1e-6 * float(ref32['costs'].abs().max())
    np.testing.assert_allclose(costs.cpu().numpy(), ref32['costs'].numpy(), rtol=5e-5, atol=ksig_atol)
    ill = _ill_conditioned(ref64['costs'], temperature)
    good = ~ill
    assert int(good.sum()) >= 0.9 * good.numel(), (tag, int(ill.sum()))     # the bulk of the batch is checked at the strict bar
    m, r32, r64 = means.cpu(), ref32['means'], ref64['means']
    e_g, e_w = _gmax(m[good], r32[good]), rel_err_waypoint(m[good], r32[good], n_pos)
    np.testing.assert_allclose(weights.cpu().numpy()[good.numpy()], ref32['weights'].numpy()[good.numpy()], rtol=1e-3, atol=1e-5)
    env = _gmax(r32, r64)
    e_all = _gmax(m, r32)
    print('%s: %d / %d particles well conditioned: means global-max %.2e, per-waypoint %.2e; all particles %.2e (reference '
          'fp32-vs-fp64 envelope %.2e)' % (tag, int(good.sum()), good.numel(), e_g, e_w, e_all, env))
    assert e_g < REL and e_w < REL, (tag, e_g, e_w)
    assert e_all < max(REL, 2.0 * env), (tag, e_all, env)
    return e_g, e_w


@pytest.mark.parametrize('P,path,H', [(128, 'exchange', 64), (256, 'two-batch', 64), (128, 'exchange', 128)])
def test_stomp_c3_persistent_vs_oracle_full_size(gpu_device, P, path, H):
    """(H = 128: the bench line's `h128` entry, the generalised persistent kernel)"""
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    S, n_it = 32, 2
    wl, Sigma, L, geom = _c3(dev, P, S, H)
    prm = wl['params']
    d = wl['means0'].shape[-1]
    ksig = 1.0 / wl['sigma_coll'] ** 2
    ws = ops.stomp_workspace(P, S, H, d, dev)
    want = ops.STOMP_PATH_PERSISTENT_EXCHANGE if path == 'exchange' else ops.STOMP_PATH_PERSISTENT
    assert ops.stomp_run_path(geom, ws, P, S, H, d) == want
    eps = torch.randn(n_it, S, d, P, H, generator=torch.Generator().manual_seed(42))
    eps_d = eps.to(dev)
    samples, costs, weights = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
    status = ops.StompRunStatus()

    def run(m0, e, n):
        means = m0.clone().to(dev)
        ops.stomp_run(means, e.contiguous(), samples, costs, weights, L.to(dev), Sigma.to(dev), geom, S, 7, ksig, 1.0,
                      prm['step_size'], prm['temperature'], ws, n_iters=n, status=status)
        torch.cuda.synchronize()
        assert not ops.stomp_run_timed_out(ws) and status.lost() is None
        return means, samples.clone(), costs.clone(), weights.clone()

    # teacher forced: every iteration from the oracle's own fp32 iterate
    prev = wl['means0'].cpu()
    refs = []
    for it in range(n_it):
        r32, r64 = _oracle_iter(wl, prev, eps[it], L, Sigma, torch.float32), _oracle_iter(wl, prev, eps[it], L, Sigma, torch.float64)
        _check_iteration('C3 P=%d H=%d teacher-forced it %d' % (P, H, it), run(prev, eps_d[it:it + 1], 1), r32, r64, 7, prm['temperature'])
        refs.append(r32)
        prev = r32['means']
    # free running: both iterations inside ONE persistent launch, against the oracle's free-running fp32 / fp64 runs
    m64 = wl['means0'].cpu().double()
    for it in range(n_it):
        r64 = _oracle_iter(wl, m64, eps[it], L, Sigma, torch.float64)
        m64 = r64['means']
    got = run(wl['means0'].cpu(), eps_d, n_it)
    env = _gmax(refs[-1]['means'], m64)
    err, errw = _gmax(got[0], refs[-1]['means']), rel_err_waypoint(got[0], refs[-1]['means'], 7)
    print('C3 P=%d H=%d free-running %d iterations: global-max %.2e per-waypoint %.2e (reference fp32-vs-fp64 envelope %.2e)' % (P, H, n_it, err, errw, env))
    assert err < max(REL, 2.0 * env)


K20_CASES = [
    # P,  path,        H,   particles the oracle runs (the first n: particles are independent), floor on the strict fraction
    # (20 oracle iterations in fp32 and fp64 cost ~1 s per particle on the box's cores: the test takes a prefix of the batch;
    # bench.py's `parity.philox` object runs the WHOLE headline batch at K = 20 on the driver's box -- round 6, first run of
    # this test with 128 / 64 / 48 oracle particles: strict fractions 1.0 / 0.984 (one particle off in the reference's own
    # fp64 run too: a near-tie, envelope 0.58) / 1.0, worst strict error 3.8e-7 global, 1.6e-6 per waypoint)
    (128, 'exchange', 64, 40),         # C3, the headline
    (256, 'two-batch', 64, 16),        # the `c5` entry's layout
    (128, 'exchange', 128, 12),        # the `h128` entry: generalised kernel
]


@pytest.mark.parametrize('P,path,H,n_or', K20_CASES)
def test_stomp_k20_free_running_vs_oracle(gpu_device, P, path, H, n_or):
    """Parity AT THE TIMED HORIZON: bench.py times K = 20 iterations inside one persistent launch on
    device-drawn noise; here exactly that launch -- eps = NULL, 20 iterations, C3 / two-batch / H = 128 -- runs free from the
    initial means, its drawn normals are fetched (mpb_debug_stomp_normals_h), and
      (a) the same normals INJECTED (the reference's draw order, stomp.py:97-108) give bit-identical means / samples / costs /
          weights after the 20 iterations: both noise modes are one computation;
      (b) the oracle (stomp.py:150-160) runs the 20 iterations free on those normals in fp32 and in fp64: the FINAL waypoints
          are judged per particle (bench.parity_by_particle): the share of particles within 1e-4 in both norms must reach the
          reference's own share (its fp32 run against its fp64 run) less 10 points and at least FLOOR, and every other particle
          stays within 2 x the reference's fp32-vs-fp64 envelope -- over 20 iterations of a one-hot softmax (sigma_coll = 1e-3)
          a near-tie that two arithmetics resolve differently sends a particle down another sample's path, in the reference
          itself."""
    import bench
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    S, K, seed, it0, off = 32, 20, 11, 0, 0
    FLOOR = 0.5
    wl, Sigma, L, geom = _c3(dev, P, S, H)
    prm = wl['params']
    d = wl['means0'].shape[-1]
    ksig = 1.0 / wl['sigma_coll'] ** 2
    ws = ops.stomp_workspace(P, S, H, d, dev)
    want = ops.STOMP_PATH_PERSISTENT_EXCHANGE if path == 'exchange' else ops.STOMP_PATH_PERSISTENT
    assert ops.stomp_run_path(geom, ws, P, S, H, d) == want
    out = [torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)]
    status = ops.StompRunStatus()

    def run(eps):
        means = wl['means0'].clone()
        ops.stomp_run(means, eps, *out, L.to(dev), Sigma.to(dev), geom, S, 7, ksig, 1.0, prm['step_size'], prm['temperature'], ws,
                      n_iters=K, seed=seed, iter0=it0, particle_offset=off, status=status)
        torch.cuda.synchronize()
        assert not ops.stomp_run_timed_out(ws) and status.lost() is None
        return [means.clone()] + [t.clone() for t in out]
    nrm = ops.debug_stomp_normals(P, S, d, K, dev, seed=seed, iter0=it0, particle_offset=off, H=H)
    eps = nrm[..., :H].permute(0, 2, 3, 1, 4).contiguous()            # the reference's draw order (K, S, d, P, H)
    del nrm
    drawn = run(None)
    injected = run(eps)
    for name, a, b in zip(('means', 'samples', 'costs', 'weights'), drawn, injected):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    eps_cpu = eps[:, :, :, :n_or].cpu()
    del eps, injected
    finals = {}
    for dtype in (torch.float32, torch.float64):
        m = wl['means0'][:n_or].cpu().to(dtype)
        for it in range(K):
            m = _oracle_iter(wl, m, eps_cpu[it], L, Sigma, dtype)['means']
        finals[dtype] = m
    st = bench.parity_by_particle(drawn[0][:n_or], finals[torch.float32], finals[torch.float64], n_pos=7, bar=REL)
    print('K=20 free-running, P=%d H=%d %s (oracle on %d particles): %s' % (P, H, path, n_or, st))
    assert st['frac_strict'] >= max(FLOOR, st['reference_frac_strict'] - 0.10), st
    assert st['worst_rest'] <= 2.0 * st['envelope'] + REL, st


def test_chomp_c2_vs_oracle_full_size(gpu_device):
    """C2 at B = 1024: teacher-forced single iterations at the strict bar, then 20 iterations in ONE launch against the
    oracle's autograd restatement (chomp.py:134-149); the clipped, B-scaled smoothness gradient (quirk Q3) makes the free
    run ill-conditioned in the reference itself, hence the envelope."""
    from motion_planning_baselines_amd import ops, workloads
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    B, H, n_it = 1024, 64, 20
    wl = workloads.pointmass_dense_chomp(B, dev)
    prm = wl['params']
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    kw = dict(D=2, k_sigma=1.0 / wl['sigma_coll'] ** 2, weight=wl['weight'], w_prior=prm['weight_prior_cost'], lr=prm['step_size'],
              grad_clip=prm['grad_clip'])

    def oracle(m, dtype, n):
        ta = dict(device='cpu', dtype=dtype)
        robot, field = make_ref_geometry(wl['robot'], wl['field'], ta)
        R = O.chomp_precision(H, prm['dt'], dict(device='cpu', dtype=torch.float32)).to(dtype)
        m = m.to(dtype)
        out = []
        for _ in range(n):
            m = O.chomp_iteration(m, R, lambda x: O.collision_cost(x, robot, field, wl['sigma_coll'], wl['weight']),
                                  prm['weight_prior_cost'], prm['step_size'], prm['grad_clip'])['means']
            out.append(m)
        return out
    R = O.chomp_precision(H, prm['dt'], dict(device='cpu', dtype=torch.float32)).to(dev)
    ref32 = oracle(wl['means0'].cpu(), torch.float32, n_it)
    ref64 = oracle(wl['means0'].cpu(), torch.float64, n_it)
    # teacher forced from the oracle's own iterate.  At B = 1024 quirk Q3's factor B puts the fp32 rounding of the smoothness
    # gradient (a stencil that cancels to ~1e-7 of its terms) at the size of the clip even within ONE iteration: the bar is
    # max(1e-5, 2 x the reference's own fp32-vs-fp64 envelope of that iteration), and the HIP result must be at least that
    # close to the fp64 run as well
    for it in (0, 1, n_it - 1):
        prev = wl['means0'].cpu() if it == 0 else ref32[it - 1]
        r32, r64 = oracle(prev, torch.float32, 1)[0], oracle(prev, torch.float64, 1)[0]
        m = prev.clone().to(dev)
        ops.chomp_step(m, R, geom, n_iters=1, **kw)
        torch.cuda.synchronize()
        env = _gmax(r32, r64)
        e, e64, ew = _gmax(m, r32), _gmax(m, r64), rel_err_waypoint(m, r64, 2)
        print('C2 B=%d teacher-forced it %d: vs oracle fp32 %.2e, vs oracle fp64 %.2e (per-waypoint %.2e); reference fp32-vs-fp64 '
              'envelope %.2e' % (B, it, e, e64, ew, env))
        assert e < max(1e-5, 2.0 * env) and e64 < max(1e-5, 2.0 * env), (it, e, e64, env)
    m = wl['means0'].clone()
    ops.chomp_step(m, R, geom, n_iters=n_it, **kw)
    torch.cuda.synchronize()
    env = _gmax(ref32[-1], ref64[-1])
    err, errw = _gmax(m, ref32[-1]), rel_err_waypoint(m, ref32[-1], 2)
    print('C2 B=%d free-running %d iterations: global-max %.2e per-waypoint %.2e (reference fp32-vs-fp64 envelope %.2e)' % (B, n_it, err, errw, env))
    assert err < max(REL, 2.0 * env)


def test_mppi_bench_shape_vs_oracle(gpu_device):
    """The `mppi` entry of the bench line (point mass among the grid circles, S = 32, T = 64, c = 2, k_sigma = 1e6: one
    workgroup per problem, collision through the broad-phase grid as offset words) at 512 of its 1 024 problems, two
    iterations on injected normals against the oracle's sequential rollout (mppi.py:131-209, point.py:102-226; quirk Q6: the
    summed collision cost of ALL samples shifts every sample's cost).  Controls, states and costs of every problem: 1e-5 /
    1e-5 / 2e-5 (measured 4e-7 / 7e-7 / 2e-6).  The shift is ~4e6 at temperature 1: softmax(-cost) is one-hot on the
    cheapest sample, and where the two cheapest costs sit within fp32 rounding of such costs of each other the reference's own
    fp32 result is a coin toss -- weights and mean are compared (1e-4) on the problems whose winner is unambiguous in both
    iterations (268 of the 512)."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.planners.priors.gaussian import const_ctrl_Cov
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    NP, S, T, c, n_it = 512, 32, 64, 2, 2
    ta = dict(device='cpu', dtype=torch.float32)
    Cov = const_ctrl_Cov([0.3, 0.3], T, c, ta)
    tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous()
    cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous()
    gen = torch.Generator().manual_seed(0)
    state0 = torch.rand(NP, c, generator=gen) * 0.2 - 0.9
    goal = torch.rand(NP, c, generator=gen) * 0.2 + 0.7
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d()
    geom = ops.DeviceGeometry(robot, field, dev)
    assert (geom.flags & 0x1500) == 0x1500                       # one grid-backed field, point robot: the grid instantiation
    rr, rf = make_ref_geometry(robot, field, ta)
    eps = torch.randn(n_it, NP, c, S, T, generator=gen)
    f = lambda t: t.contiguous().to(dev)
    mean = torch.zeros(NP, T, c, device=dev)
    controls, states = torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, T, c, device=dev)
    costs, weights = torch.empty(NP, S, device=dev), torch.empty(NP, S, device=dev)
    cmin, cmax, disc = torch.tensor([-1., -1.]), torch.tensor([1., 1.]), torch.ones(T)
    cw = dict(pos=1.0, vel=1.0, ctrl=1.0, pos_T=100.0)
    ks, wt, temp, step, dt = 1e6, 1.0, 1.0, 0.7, 0.04
    ops.mppi_step(mean, f(eps), f(tril), f(cinv), f(state0), f(goal), f(cmin), f(cmax), f(disc), f(torch.tensor([1., 1., 1., 100.])),
                  geom, controls, states, costs, weights, dt, k_sigma=ks, weight=wt, temp=temp, step_size=step, n_iters=n_it)
    torch.cuda.synchronize()
    n_clear = 0
    for p in range(NP):
        m = torch.zeros(T, c)
        clear = True
        for it in range(n_it):
            pre = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c)
            q = pre['states'][:, 1:, :2]
            shift = wt * ks * float(rf.compute_cost(q, rr.fk_map_collision(q)).sum())
            out = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c, shift_cost=shift)
            m = out['mean']
            cs = np.sort(out['costs'].reshape(-1).double().numpy())
            # (the kernel's costs agree with the oracle's to ~2e-6 relative: a winner is unambiguous when it leads by twice that
            # plus twelve temperatures -- exp(-12) = 6e-6, below the bar on the weights)
            clear = clear and (cs[1] - cs[0]) > 2.0 * 2e-6 * abs(cs[0]) + 12.0 * temp
        assert _gmax(controls[p], out['controls']) < 1e-5 and _gmax(states[p], out['states']) < 1e-5, p
        assert _gmax(costs[p], out['costs'].reshape(-1)) < 2e-5, p
        if clear:
            n_clear += 1
            assert float((weights[p].cpu() - out['weights'].reshape(-1)).abs().max()) < 1e-4, p
            assert _gmax(mean[p], m) < REL, p
    print('problems with an unambiguous winner in both iterations: %d of %d' % (n_clear, NP))
    assert n_clear > NP // 3, n_clear
