"""-m gpu: the persistent one-launch STOMP loop (csrc/mpb_stomp_fused.hip, mpb_stomp_run) against the reference-generated
goldens, against the oracle, and against the two-kernel path (mpb_stomp_step) it replaces where the shape allows."""
import numpy as np
import pytest
import torch

from conftest import load_golden, ref_geometry_from_golden
from test_gpu_parity_ops import (REL, STOMP_CASES, T, _stomp_bufs, dev_geom, reference_fp32_envelope,
                                 reference_fp32_envelope_one_iteration, rel_err)

pytestmark = pytest.mark.gpu


def _ws(P, S, H, d, dev):
    from motion_planning_baselines_amd import ops
    return ops.stomp_workspace(P, S, H, d, dev)


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_run_teacher_forced_vs_golden(gpu_device, name):
    """One pass of the loop body from the reference's own iterate, through mpb_stomp_run (H = 64 cases run the
    persistent kernel, the others its two-kernel fallback): samples, costs, weights, means against the golden."""
    from motion_planning_baselines_amd import ops
    g = load_golden(name)
    dev = gpu_device
    P, S, H, d, samples, costs, weights = _stomp_bufs(g, dev)
    geom = dev_geom(g, dev)
    ws = _ws(P, S, H, d, dev)
    L, Sigma = T(g['L']).to(dev), T(g['Sigma']).to(dev)
    ksig = 1.0 / float(g['sigma_coll']) ** 2
    prev = T(g['means0'])
    for it in range(g['eps'].shape[0]):
        means = prev.clone().to(dev)
        eps = T(g['eps'][it:it + 1]).contiguous().to(dev)
        ops.stomp_run(means, eps, samples, costs, weights, L, Sigma, geom, S, int(g['D']), ksig, 1.0,
                      float(g['lr']), float(g['temperature']), ws)
        torch.cuda.synchronize()
        assert not ops.stomp_run_timed_out(ws)
        assert rel_err(samples, T(g['samples'][it])) < 2e-5, it
        np.testing.assert_allclose(costs.cpu().numpy(), T(g['costs'][it]).numpy(), rtol=5e-5, atol=1e-6 * ksig)
        cond = float(ksig) * 1e-6
        if cond < 1e-2:
            np.testing.assert_allclose(weights.cpu().numpy(), T(g['weights'][it]).numpy(), rtol=1e-3, atol=1e-5)
        bar = REL if cond < 1e-2 else max(REL, 2.0 * reference_fp32_envelope_one_iteration(g, it))
        assert rel_err(means, T(g['means'][it])) < bar, (it, bar)
        prev = T(g['means'][it])


@pytest.mark.parametrize('name', STOMP_CASES)
def test_stomp_run_free_running_vs_golden(gpu_device, name):
    """All iterations in ONE launch from means0 on the reference's injected noise: north_star's criterion."""
    from motion_planning_baselines_amd import ops
    g = load_golden(name)
    dev = gpu_device
    P, S, H, d, samples, costs, weights = _stomp_bufs(g, dev)
    n = g['eps'].shape[0]
    ws = _ws(P, S, H, d, dev)
    means = T(g['means0']).clone().to(dev)
    ops.stomp_run(means, T(g['eps']).contiguous().to(dev), samples, costs, weights, T(g['L']).to(dev),
                  T(g['Sigma']).to(dev), dev_geom(g, dev), S, int(g['D']), 1.0 / float(g['sigma_coll']) ** 2, 1.0,
                  float(g['lr']), float(g['temperature']), ws, n_iters=n)
    torch.cuda.synchronize()
    assert not ops.stomp_run_timed_out(ws)
    err = rel_err(means, T(g['means'][-1]))
    env = reference_fp32_envelope(g)
    print(name, 'final-waypoint rel err', err, 'reference fp32-vs-fp64 envelope', env)
    assert err < max(REL, 2.0 * env)
    if name != 'stomp_panda_benign':
        assert err < REL


@pytest.mark.parametrize('P,S,pos_only,n_iters', [
    (128, 32, False, 4),     # C3: two workgroups per particle, the chip filled once
    (8, 16, False, 3),       # one workgroup per particle: no exchange
    (8, 64, True, 3),        # four workgroups per particle, d = 7
    (3, 30, False, 3),       # the reference example's S = 30: a partial last chunk; P not a multiple of 8
    (5, 5, True, 2),         # a quarter of one chunk
    (300, 32, False, 2),     # more units than CUs: workgroups run in rounds, partners are still co-scheduled
    (256, 32, False, 2),     # as many particles as CUs: one workgroup per particle, two batches of 16 samples (no exchange)
    (260, 24, True, 2)])     # the same layout with a partial second batch, d = 7 (MPB_STOMP_BATCHES unset: chosen by the launcher)
def test_stomp_run_equals_two_kernel_path(gpu_device, P, S, pos_only, n_iters):
    """Same device noise (Philox keyed by particle / sample / iteration), same geometry.
    (a) One iteration: the persistent kernel and the two-kernel loop write the same samples and costs bit for bit and
        agree on weights and means to rounding (the persistent kernel combines per-chunk softmax partials:
        exp(x - m) / z in a different association).
    (b) n iterations in ONE persistent launch == n launches of one iteration each, bit for bit: the in-launch
        hand-off between the workgroups of a particle (parity buffers, flags, next-iteration noise drawn early)
        changes nothing.  (Comparing n free-running iterations across the two PATHS is not a test of either: the loop
        amplifies a 1e-6 difference of the weights ~1e3-fold per iteration at these cost scales.)"""
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    dev = gpu_device
    H = 64
    wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=pos_only)
    d = wl['means0'].shape[-1]
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, wl['params']['dt'], 0.02, cpu)
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    mk = lambda: (torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev))
    ws = _ws(P, S, H, d, dev)
    args = (L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1e5)
    # (a)
    mf, (sf, cf, wf) = wl['means0'].clone(), mk()
    ops.stomp_run(mf, None, sf, cf, wf, *args, ws, n_iters=1, seed=11)
    mt, (st, ct, wt) = wl['means0'].clone(), mk()
    ops.stomp_step(mt, None, st, ct, wt, *args, n_iters=1, seed=11)
    torch.cuda.synchronize()
    assert not ops.stomp_run_timed_out(ws)
    assert float(ct.max()) > 0
    assert torch.equal(sf, st) and torch.equal(cf, ct)
    np.testing.assert_allclose(wf.cpu().numpy(), wt.cpu().numpy(), rtol=2e-5, atol=1e-7)
    assert rel_err(mf, mt) < 2e-6
    # (b)
    m1, (s1, c1, w1) = wl['means0'].clone(), mk()
    ops.stomp_run(m1, None, s1, c1, w1, *args, ws, n_iters=n_iters, seed=11, iter0=5)
    torch.cuda.synchronize()
    assert not ops.stomp_run_timed_out(ws)
    m2, (s2, c2, w2) = wl['means0'].clone(), mk()
    for it in range(n_iters):
        ops.stomp_run(m2, None, s2, c2, w2, *args, ws, n_iters=1, seed=11, iter0=5 + it)
    torch.cuda.synchronize()
    assert torch.equal(m1, m2) and torch.equal(s1, s2) and torch.equal(c1, c2) and torch.equal(w1, w2)
    assert torch.isfinite(m1).all()


def test_stomp_run_pointmass_generic_model(gpu_device):
    """The generic (table-driven) instantiation: C1's point-mass problem, d = 4, S = 4 (a quarter chunk)."""
    from motion_planning_baselines_amd import ops, workloads
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    dev = gpu_device
    wl = workloads.pointmass_grid_circles_stomp(dev)
    P, H, d = wl['means0'].shape
    S = wl['params']['num_samples']
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(H, wl['params']['dt'], 0.1, cpu)
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
    def run(n):
        res = []
        for fused in (True, False):
            means = wl['means0'].clone()
            samples = torch.empty(P, S, H, d, device=dev)
            costs, weights = torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
            if fused:
                ws = _ws(P, S, H, d, dev)
                ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 2, 1e6, 1.0, 0.1, 1e5, ws, n_iters=n, seed=2)
            else:
                ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 2, 1e6, 1.0, 0.1, 1e5, n_iters=n, seed=2)
            torch.cuda.synchronize()
            res.append((means, samples, costs))
        return res
    # one iteration: the same samples and costs bit for bit (same noise functions, same evaluator), means to the rounding of
    # the two softmax forms; five free-running iterations on device noise: north_star's bar (the two forms' 1e-6 per
    # iteration is amplified by the loop: 2.8e-5 measured with this seed)
    one = run(1)
    assert torch.equal(one[0][1], one[1][1]) and torch.equal(one[0][2], one[1][2])
    assert rel_err(one[0][0], one[1][0]) < 5e-6
    res = run(5)
    assert rel_err(res[0][0], res[1][0]) < 1e-4 and rel_err(res[0][1], res[1][1]) < 1e-4
    np.testing.assert_allclose(res[0][2].cpu().numpy(), res[1][2].cpu().numpy(), rtol=1e-3, atol=1e-2)


_LAYOUT_CHILD = r"""
import sys, numpy as np, torch
from motion_planning_baselines_amd import ops, workloads
from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
out, P, S, pos_only, n_iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), int(sys.argv[5])
dev = torch.device('cuda:0')
H = 64
wl = workloads.panda_spheres_stomp(P, dev, H=H, S=S, pos_only=pos_only)
d = wl['means0'].shape[-1]
R = stomp_precision_matrix(H, wl['params']['dt'], 0.02, dict(device='cpu', dtype=torch.float32))
Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
geom = ops.DeviceGeometry(wl['robot'], wl['field'], dev)
m = wl['means0'].clone()
s, c, w = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
ws = ops.stomp_workspace(P, S, H, d, dev)
ops.stomp_run(m, None, s, c, w, L, Sigma, geom, S, 7, 1e6, 1.0, 0.1, 1e5, ws, n_iters=n_iters, seed=3, iter0=2)
torch.cuda.synchronize()
assert not ops.stomp_run_timed_out(ws)
np.savez(out, means=m.cpu().numpy(), samples=s.cpu().numpy(), costs=c.cpu().numpy(), weights=w.cpu().numpy())
"""


@pytest.mark.parametrize('P,S,pos_only', [(6, 32, False), (5, 24, True), (9, 17, False)])
def test_two_batch_layout_equals_exchange_layout(gpu_device, tmp_path, P, S, pos_only):
    """The two work layouts of the persistent kernel -- two workgroups per particle exchanging their partials through the
    workspace, or one workgroup running the particle's samples as two batches of 16 -- give the same bits: means,
    samples, costs and weights after three iterations (MPB_STOMP_BATCHES forces the layout; one child process each)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = {}
    for nb in (1, 2):
        out = str(tmp_path / f'nb{nb}.npz')
        env = dict(os.environ, MPB_STOMP_BATCHES=str(nb), PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
        r = subprocess.run([sys.executable, '-c', _LAYOUT_CHILD, out, str(P), str(S), str(int(pos_only)), '3'],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[nb] = np.load(out)
    for k in ('samples', 'costs', 'weights', 'means'):
        assert res[1][k].tobytes() == res[2][k].tobytes(), k
    assert np.isfinite(res[1]['means']).all() and float(res[1]['costs'].max()) > 0


@pytest.mark.parametrize('shift', [(0.0, 0.0, 0.0), (-7.3, 2.1, -0.4), (55.0, -31.0, 12.0)])
def test_lattice_grid_offset_equals_floor_form_on_shifted_scenes(gpu_device, shift):
    """Geometry version 6: the persistent kernel takes a point's cell as round(x / h) - K from one fma per axis (grid_cell_rel,
    csrc/mpb_geom.h), the two-kernel path as floor((x - lo) / h) from the same header.  A 3-D point robot among spheres shifted
    far from the origin (negative and large lattice indices K): rollouts that cross cell faces everywhere -- the costs of the two
    paths are the same BITS (candidate sets are conservative on both), and equal the exhaustive evaluator's."""
    from motion_planning_baselines_amd import geometry as G, ops
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    dev = gpu_device
    rng = np.random.default_rng(3)
    off = np.array(shift)
    sph = np.concatenate([rng.uniform(-1, 1, (14, 3)) + off, rng.uniform(0.05, 0.2, (14, 1))], 1).astype(np.float32)
    robot, field = G.RobotPointMass(3, radius=0.02), G.CollisionField(spheres=sph, margin=0.03)
    geom = ops.DeviceGeometry(robot, field, dev)
    assert geom.flags & 0x100
    P, S, H, d = 6, 16, 64, 6
    a = torch.linspace(0, 1, H).reshape(1, H, 1)
    s0 = torch.tensor(rng.uniform(-1, 1, (P, 1, 3)) + off, dtype=torch.float32)
    g0 = torch.tensor(rng.uniform(-1, 1, (P, 1, 3)) + off, dtype=torch.float32)
    means0 = torch.cat([s0 * (1 - a) + g0 * a, torch.zeros(P, H, 3)], -1).contiguous().to(dev)
    R = stomp_precision_matrix(H, 0.04, 0.1, dict(device='cpu', dtype=torch.float32))
    Sigma, L = torch.inverse(R).to(dev).contiguous(), precision_to_scale_tril(R).to(dev).contiguous()
    out = []
    for fused in (True, False):
        means = means0.clone()
        samples, costs, weights = torch.empty(P, S, H, d, device=dev), torch.empty(P, S, device=dev), torch.empty(P, S, device=dev)
        if fused:
            ws = _ws(P, S, H, d, dev)
            assert ops.stomp_run_path(geom, ws, P, S, H, d) != ops.STOMP_PATH_TWO_KERNEL
            ops.stomp_run(means, None, samples, costs, weights, L, Sigma, geom, S, 3, 1e2, 1.0, 0.1, 1.0, ws, n_iters=1, seed=5)
        else:
            ops.stomp_step(means, None, samples, costs, weights, L, Sigma, geom, S, 3, 1e2, 1.0, 0.1, 1.0, n_iters=1, seed=5)
        torch.cuda.synchronize()
        out.append((samples, costs))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert float(out[0][1].max()) > 0.0                                     # the rollouts do collide
    # ... and the stand-alone evaluator (exhaustive for point robots) on the same samples
    ref = ops.cost_collision_eval(out[0][0].reshape(P * S, H, d).contiguous(), geom, 1e2).reshape(P, S)
    np.testing.assert_allclose(out[0][1].cpu().numpy(), ref.cpu().numpy(), rtol=2e-6, atol=1e-6)
