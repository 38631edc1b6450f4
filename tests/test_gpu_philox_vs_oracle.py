"""-m gpu: the code path bench.py TIMES -- device-drawn noise (eps = NULL) -- under the oracle.

With eps = NULL the STOMP kernels draw their normals on the device (Philox4x32-7 + Box-Muller) and push them through the
TWO-component bf16 split and the 30-instruction matrix product (csrc/mpb_stomp_noise.h, stomp_split8<false> /
stomp_noise_product_kb<..., LOW = false>); every golden / oracle test of rounds 1-4 injects eps and therefore runs the
THREE-component, 36-instruction instantiation.  The two are different template instantiations: "what is timed" was only
ever compared with itself (persistent == two-kernel) or at L = identity.  Here the drawn normals are fetched through the
test aid mpb_debug_stomp_normals_h for the same (seed, iter0, particle_offset) and

  (i)   samples == means + L @ normals (fp64 product on the host, rows 0 / H - 1 zero: stomp.py:97-108) to <= 2e-6;
  (ii)  those normals, permuted to the reference's draw order (S, d, P, H), are fed to oracle.stomp_iteration
        (stomp.py:150-160): samples / costs / weights / means meet the bars of tests/test_gpu_oracle_full_size.py,
        teacher-forced per iteration and free-running over both iterations of ONE launch;
  (iii) the same normals INJECTED through the eps argument (the LOW = true instantiation; their third bf16 component is
        zero by construction) give bit-identical samples, costs, weights and means.

(Round 5, later: the H = 64 kernel has separate instantiations for drawn and injected noise -- `INJ` -- so (iii) also holds the two
instantiations to the same bits.)
Shapes: C3 (P = 128, S = 32, H = 64, d = 14, sigma_coll = 1e-3: persistent exchange layout, stomp_fused_kernel<14,1,1,false> --
the headline's instantiation), P = 256 (two-batch layout, <14,1,2>: the `c5` entry's), H = 128 (stomp_fused_hx_kernel, the
product issued TRANSPOSED, chunked draw), d = 7 pos_only (stomp_noise_bf16_pair: two rollouts per product), H = 48 (the
generalised kernel on a partial chunk: the drawn columns k >= H meet zero columns of L) and the two-kernel path
(mpb_stomp_step: stomp_sample_cost_h64_kernel / _hx_kernel + the update kernel)."""
import pytest
import torch

from conftest import rel_err_waypoint
from test_gpu_oracle_full_size import _c3, _check_iteration, _gmax, _oracle_iter, REL

pytestmark = pytest.mark.gpu

SEED, IT0, OFF = 20251004, 3, 4096


def _setup(dev, P, S, H, pos_only):
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=pos_only)
    prm = wl['params']
    R = stomp_precision_matrix(H, prm['dt'], prm['sigma_spectral'], dict(device='cpu', dtype=torch.float32))
    Sigma, L = torch.inverse(R).contiguous(), precision_to_scale_tril(R).contiguous()
    return wl, Sigma, L, ops.DeviceGeometry(wl['robot'], wl['field'], dev)


CASES = [
    # P,   S,  H,   pos_only, path
    (128, 32, 64, False, 'exchange'),       # C3: the headline's instantiation
    (256, 32, 64, False, 'two-batch'),      # the `c5` entry's layout
    (128, 32, 128, False, 'exchange'),      # the `h128` entry: generalised kernel, transposed product, chunked draw
    (128, 32, 64, True, 'exchange'),        # d = 7: two rollouts per product (stomp_noise_bf16_pair)
    (24, 32, 48, False, 'exchange'),        # generalised kernel, one partial chunk
    (128, 32, 64, False, 'two-kernel'),     # mpb_stomp_step: kernel A + kernel B
    (16, 32, 128, False, 'two-kernel'),     # ... and its chunked form
]


@pytest.mark.parametrize('P,S,H,pos_only,path', CASES)
def test_device_noise_path_vs_oracle(gpu_device, P, S, H, pos_only, path):
    from motion_planning_baselines_amd import ops
    dev = gpu_device
    n_it = 2
    wl, Sigma, L, geom = _setup(dev, P, S, H, pos_only)
    prm = wl['params']
    d = wl['means0'].shape[-1]
    assert d == (7 if pos_only else 14)
    ksig = 1.0 / wl['sigma_coll'] ** 2
    Ld, Sd = L.to(dev), Sigma.to(dev)
    ws = ops.stomp_workspace(P, S, H, d, dev)
    if path != 'two-kernel':
        want = ops.STOMP_PATH_PERSISTENT_EXCHANGE if path == 'exchange' else ops.STOMP_PATH_PERSISTENT
        assert ops.stomp_run_path(geom, ws, P, S, H, d) == want
    samples, costs, weights = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
    status = ops.StompRunStatus()

    def run(m0, n, it0, eps=None):
        """n iterations from m0 with the device's own draw (eps None) or injected normals (eps (n,S,d,P,H))"""
        means = m0.clone().to(dev)
        kw = dict(n_iters=n, seed=SEED, iter0=it0, particle_offset=OFF)
        if path == 'two-kernel':
            ops.stomp_step(means, eps, samples, costs, weights, Ld, Sd, geom, S, 7, ksig, 1.0, prm['step_size'], prm['temperature'], **kw)
            torch.cuda.synchronize()
        else:
            ops.stomp_run(means, eps, samples, costs, weights, Ld, Sd, geom, S, 7, ksig, 1.0, prm['step_size'], prm['temperature'],
                          ws, status=status, **kw)
            torch.cuda.synchronize()
            assert not ops.stomp_run_timed_out(ws) and status.lost() is None
        return means, samples.clone(), costs.clone(), weights.clone()

    # the normals of iterations IT0, IT0 + 1 exactly as the kernels draw them: (n_it, P, S, d, 64 ceil(H / 64))
    nrm = ops.debug_stomp_normals(P, S, d, n_it, dev, seed=SEED, iter0=IT0, particle_offset=OFF, H=H)
    torch.cuda.synchronize()
    assert torch.isfinite(nrm).all()
    eps = nrm[..., :H].permute(0, 2, 3, 1, 4).contiguous()          # the reference's draw order (n_it, S, d, P, H): stomp.py:98-101
    eps_cpu = eps.cpu()
    # the drawn normals have two bf16 components: the third one of the injected path's split is exactly zero
    e64 = eps_cpu.double()
    hi = (eps_cpu.view(torch.int32) & -65536).view(torch.float32)
    lo = eps_cpu - hi
    assert torch.equal((lo.view(torch.int32) & -65536).view(torch.float32), lo)

    tag = 'P=%d S=%d H=%d d=%d %s' % (P, S, H, d, path)
    prev = wl['means0'].cpu()
    refs = []
    for it in range(n_it):
        got = run(prev, 1, IT0 + it)
        # (i) samples == means + L @ normals, rows 0 / H-1 of the noise zero (fp64 on the host)
        noise = torch.einsum('hk,sdpk->pshd', L.double(), e64[it])
        noise[:, :, 0, :] = 0
        noise[:, :, -1, :] = 0
        want = prev.double().unsqueeze(1) + noise
        e_s = float((got[1].cpu().double() - want).abs().max() / want.abs().max())
        # (iii) the same normals injected: bit-identical
        inj = run(prev, 1, IT0 + it, eps=eps[it:it + 1].contiguous())
        for name, a, b in zip(('means', 'samples', 'costs', 'weights'), got, inj):
            assert torch.equal(a, b), (tag, it, name, float((a - b).abs().max()))
        # (ii) the oracle on those normals
        r32, r64 = _oracle_iter(wl, prev, eps_cpu[it], L, Sigma, torch.float32), _oracle_iter(wl, prev, eps_cpu[it], L, Sigma, torch.float64)
        print('%s it %d: samples vs means + L @ normals (fp64) %.2e' % (tag, it, e_s))
        assert e_s < 2e-6, (tag, it, e_s)
        _check_iteration('device noise, %s teacher-forced it %d' % (tag, it), got, r32, r64, 7, prm['temperature'])
        refs.append(r32)
        prev = r32['means']
    # free running: both iterations inside ONE launch on the device's own draw
    m64 = wl['means0'].cpu().double()
    for it in range(n_it):
        m64 = _oracle_iter(wl, m64, eps_cpu[it], L, Sigma, torch.float64)['means']
    got = run(wl['means0'].cpu(), n_it, IT0)
    inj = run(wl['means0'].cpu(), n_it, IT0, eps=eps)
    for name, a, b in zip(('means', 'samples', 'costs', 'weights'), got, inj):
        assert torch.equal(a, b), (tag, 'free-running', name)
    env = _gmax(refs[-1]['means'], m64)
    err, errw = _gmax(got[0], refs[-1]['means']), rel_err_waypoint(got[0], refs[-1]['means'], 7)
    print('device noise, %s free-running %d iterations: global-max %.2e per-waypoint %.2e (reference fp32-vs-fp64 envelope %.2e)'
          % (tag, n_it, err, errw, env))
    assert err < max(REL, 2.0 * env)


def test_mppi_device_noise_is_the_injected_path(gpu_device):
    """MPPI's throughput-mode draw (Philox4x32-7 + box_muller_m23, csrc/mpb_mppi.hip) fetched through mpb_debug_mppi_normals
    and INJECTED gives the same bits as the device draw (both feed the same fp32 matrix product), at the bench entry's shape;
    and the oracle's sequential rollout (mppi.py:131-209, point.py:102-226) on those normals agrees on controls / states / costs
    for every problem checked -- so what `mppi` times is under the oracle too."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.planners.priors.gaussian import const_ctrl_Cov
    from oracle import planners_ref as O
    from oracle.geometry_ref import make_ref_geometry
    dev = gpu_device
    NP, S, T, c, n_it, seed, it0 = 64, 32, 64, 2, 2, 11, 5
    ta = dict(device='cpu', dtype=torch.float32)
    Cov = const_ctrl_Cov([0.3, 0.3], T, c, ta)
    tril = torch.stack([torch.linalg.cholesky(Cov[..., i]) for i in range(c)]).contiguous()
    cinv = torch.stack([torch.inverse(Cov[..., i]) for i in range(c)]).contiguous()
    gen = torch.Generator().manual_seed(0)
    state0 = torch.rand(NP, c, generator=gen) * 0.2 - 0.9
    goal = torch.rand(NP, c, generator=gen) * 0.2 + 0.7
    robot, field = G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d()
    geom = ops.DeviceGeometry(robot, field, dev)
    rr, rf = make_ref_geometry(robot, field, ta)
    f = lambda t: t.contiguous().to(dev)
    cmin, cmax, disc = torch.tensor([-1., -1.]), torch.tensor([1., 1.]), torch.ones(T)
    cw = dict(pos=1.0, vel=1.0, ctrl=1.0, pos_T=100.0)
    ks, wt, temp, step, dt = 1e6, 1.0, 1.0, 0.7, 0.04

    def run(eps):
        mean = torch.zeros(NP, T, c, device=dev)
        out = [torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, T, c, device=dev), torch.empty(NP, S, device=dev),
               torch.empty(NP, S, device=dev)]
        ops.mppi_step(mean, eps, f(tril), f(cinv), f(state0), f(goal), f(cmin), f(cmax), f(disc), f(torch.tensor([1., 1., 1., 100.])),
                      geom, *out, dt, k_sigma=ks, weight=wt, temp=temp, step_size=step, n_iters=n_it, seed=seed, iter0=it0)
        torch.cuda.synchronize()
        return [mean] + out
    nrm = ops.debug_mppi_normals(NP, S, T, c, n_it, dev, seed=seed, iter0=it0)
    torch.cuda.synchronize()
    assert torch.isfinite(nrm).all() and float(nrm.abs().max()) <= 5.66 and abs(float(nrm.mean())) < 0.01 and abs(float(nrm.std()) - 1.0) < 0.01
    drawn, injected = run(None), run(nrm)
    for name, a, b in zip(('mean', 'controls', 'states', 'costs', 'weights'), drawn, injected):
        assert torch.equal(a, b), name
    eps = nrm.cpu()
    for p in range(0, NP, 8):
        m = torch.zeros(T, c)
        for it in range(n_it):
            pre = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c)
            q = pre['states'][:, 1:, :2]
            shift = wt * ks * float(rf.compute_cost(q, rr.fk_map_collision(q)).sum())
            out = O.mppi_iteration(m, eps[it, p], tril, cinv, state0[p], goal[p], dt, cmin, cmax, cw, disc, temp, step, c, shift_cost=shift)
            m = out['mean']
        assert _gmax(drawn[1][p], out['controls']) < 1e-5 and _gmax(drawn[2][p], out['states']) < 1e-5, p
        assert _gmax(drawn[3][p], out['costs'].reshape(-1)) < 2e-5, p


def test_stomp_entry_points_refuse_misaligned_pointers(gpu_device):
    """eps / L / Sigma / means / samples are moved as 16-byte vectors; a view at a 4-byte offset is refused with
    MPB_E_INVALID (include/mpb.h) instead of becoming misaligned dwordx4 accesses."""
    from motion_planning_baselines_amd import ops
    from motion_planning_baselines_amd._lib import MPBError
    dev = gpu_device
    P, S, H = 4, 8, 64
    wl, Sigma, L, geom = _setup(dev, P, S, H, False)
    d = wl['means0'].shape[-1]
    samples, costs, weights = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
    flat = torch.zeros(1 * S * d * P * H + 1, device=dev)
    eps_off = flat[1:].view(1, S, d, P, H)
    assert eps_off.is_contiguous() and eps_off.data_ptr() % 16 == 4
    args = (samples, costs, weights, L.to(dev), Sigma.to(dev), geom, S, 7, 1.0, 1.0, 0.1, 1.0)
    with pytest.raises(MPBError, match='16-byte aligned'):
        ops.stomp_step(wl['means0'].clone(), eps_off, *args, n_iters=1)
    ws = ops.stomp_workspace(P, S, H, d, dev)
    with pytest.raises(MPBError, match='16-byte aligned'):
        ops.stomp_run(wl['means0'].clone(), eps_off, *args, ws, n_iters=1)
    with pytest.raises(MPBError, match='16-byte aligned'):
        ops.stomp_sample(wl['means0'].clone(), eps_off[0], samples, L.to(dev), S)
    # the aligned call goes through
    ops.stomp_step(wl['means0'].clone(), torch.zeros(1, S, d, P, H, device=dev), *args, n_iters=1)
    torch.cuda.synchronize()
