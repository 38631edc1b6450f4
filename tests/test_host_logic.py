"""CPU: host-side logic of the product (no compute calls): the C-ABI library loads and exports every
symbol include/mpb.h declares, geometry packing round-trips through the library's validator, and the
planner constants match the reference's (golden) R / Sigma / scale_tril bit for bit."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, product_geometry_from_golden


def test_library_exports_every_declared_symbol():
    from motion_planning_baselines_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'mpb.h')).read()
    declared = set(re.findall(r'\b(mpb_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name)
    assert (h.mpb_version() & 0xFFFF) == _lib.ABI_VERSION == int(re.search(r'#define MPB_ABI_VERSION (\d+)', hdr).group(1))
    # the product library exports the product ABI ONLY: the test aids live in a library of their own (include/mpb_debug.h)
    assert not [n for n in declared if n.startswith('mpb_debug')]
    import subprocess
    syms = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (mpb_[a-z0-9_]+)$', syms, flags=re.M))
    assert exported == declared, exported ^ declared
    dhdr = open(os.path.join(ROOT, 'include', 'mpb_debug.h')).read()
    ddecl = set(re.findall(r'\b(mpb_[a-z0-9_]+)\s*\(', dhdr))
    assert ddecl == set(_lib.DEBUG_SIGNATURES), ddecl ^ set(_lib.DEBUG_SIGNATURES)
    dh = _lib.debug_lib()
    for name in ddecl:
        assert hasattr(dh, name)


def test_binding_refuses_a_library_of_another_abi_version(tmp_path, monkeypatch):
    """signatures change positionally between ABI versions -- a library that reports another MPB_ABI_VERSION (an
    old build_variants/*.so through MPB_LIB_PATH) must be refused at load time, not called with shifted arguments."""
    import subprocess
    from motion_planning_baselines_amd import _lib
    src = tmp_path / 'old.c'
    names = [n for n in _lib.SIGNATURES if n not in ('mpb_version', 'mpb_last_error')]
    src.write_text('int mpb_version(void) { return %d; }\nconst char *mpb_last_error(void) { return ""; }\n' % (_lib.ABI_VERSION - 1)
                   + ''.join('int %s(void) { return 0; }\n' % n for n in names))
    so = tmp_path / 'libold.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)])
    monkeypatch.setattr(_lib, 'LIB_PATH', str(so))
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.MPBError, match='ABI version'):
        _lib.lib()


def test_geometry_pack_validates_and_rejects_corruption():
    from motion_planning_baselines_amd import _lib, geometry as G
    for robot, field in [(G.RobotPanda(), G.env_spheres_3d()), (G.RobotPointMass(2), G.env_dense_2d()),
                         (G.RobotPointMass(3), G.env_spheres_3d()), (G.RobotPointMass(2), G.env_grid_circles_2d())]:
        buf = G.pack_geometry(robot, field)
        _lib.geom_check(buf)
        bad = buf.copy()
        bad.view(np.int32)[0] ^= 1
        with pytest.raises(_lib.MPBError):
            _lib.geom_check(bad)
        bad = buf.copy()
        bad.view(np.int32)[13] += 4
        with pytest.raises(_lib.MPBError):
            _lib.geom_check(bad)
        with pytest.raises(_lib.MPBError):
            _lib.geom_check(buf[:20])


def test_cull_table_is_conservative():
    """Every (link position, obstacle) pair within the hinge threshold must pass the cheap test."""
    from motion_planning_baselines_amd import geometry as G
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    buf = G.pack_geometry(robot, field)
    i = buf.view(np.int32)
    n_sph = i[6]
    cull = buf[i[14]:i[14] + 8 * ((n_sph + 3) // 4 * 4)].reshape(-1, 8)
    rng = np.random.RandomState(0)
    T = field.margin + robot.link_radius.max() + field.spheres[:, 3]
    for o in range(n_sph):
        c = field.spheres[o, :3].astype(np.float64)
        d = rng.randn(20000, 3)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        x = (c + d * (T[o] * rng.uniform(0.0, 1.0, (20000, 1)))).astype(np.float32)   # inside the threshold ball
        xx = (x * x).sum(1, dtype=np.float32)
        t = xx + x[:, 0] * cull[o, 0] + x[:, 1] * cull[o, 1] + x[:, 2] * cull[o, 2]
        assert (t < cull[o, 3]).all()
    assert (cull[n_sph:, 3] < -1e29).all()


@pytest.mark.parametrize('name', ['stomp_pm2d_c1', 'stomp_panda_t1', 'stomp_pm2d_h48'])
def test_stomp_constants_match_reference(name):
    from motion_planning_baselines_amd.planners.stomp import precision_to_scale_tril, stomp_precision_matrix
    g = load_golden(name)
    cpu = dict(device='cpu', dtype=torch.float32)
    R = stomp_precision_matrix(int(g['H']), float(g['dt']), float(g['sigma_spectral']), cpu)
    assert torch.equal(R, torch.from_numpy(g['R']))
    assert torch.equal(torch.inverse(R), torch.from_numpy(g['Sigma']))
    assert torch.equal(precision_to_scale_tril(R), torch.from_numpy(g['L']))


def test_chomp_constants_match_reference():
    from motion_planning_baselines_amd.planners.chomp import chomp_precision_matrix
    g = load_golden('chomp_pm2d_dense')
    R = chomp_precision_matrix(dt=float(g['dt']), n_support_points=int(g['H']), tensor_args=dict(device='cpu', dtype=torch.float32))
    assert torch.equal(R, torch.from_numpy(g['R']))


def test_planner_refuses_cpu_device():
    from motion_planning_baselines_amd._lib import MPBError
    from motion_planning_baselines_amd.planners.stomp import STOMP
    with pytest.raises(MPBError):
        STOMP(n_dof=2, n_support_points=64, num_particles_per_goal=2, num_samples=4, opt_iters=1, dt=0.04,
              start_state=torch.zeros(2), initial_particle_means=torch.zeros(2, 64, 2),
              tensor_args=dict(device='cpu', dtype=torch.float32))


def test_product_geometry_round_trip():
    g = load_golden('chomp_panda')
    robot, field = product_geometry_from_golden(g)
    from motion_planning_baselines_amd import geometry as G
    buf = G.pack_geometry(robot, field, prune_static=False)
    i = buf.view(np.int32)
    assert i[5] == len(g['link_radius']) and i[6] == len(g['spheres'])


def test_static_link_pruning_is_conservative():
    """pack_geometry leaves out only collision spheres whose hinge is zero for EVERY joint configuration: spheres on
    frame 1 whose circle about the first joint axis stays clear of all obstacles.  Checked by brute force with the oracle's
    FK over a dense sweep of q1 (the other joints cannot move frame 1) and random full configurations."""
    from motion_planning_baselines_amd import geometry as G
    from oracle.geometry_ref import make_ref_geometry
    robot = G.RobotPanda()
    for field in (G.env_spheres_3d(seed=0), G.env_spheres_3d(seed=3),
                  G.CollisionField(spheres=np.array([[0.12, 0.05, 0.18, 0.05], [0.5, 0.0, 0.5, 0.1]]), margin=0.05),
                  G.CollisionField(boxes=np.array([[0.2, 0.0, 0.1, 0.05, 0.05, 0.05]]), margin=0.05)):
        rs, fs = robot.spec(), field.spec()
        keep = G.links_that_can_touch(rs, fs)
        assert keep[np.asarray(rs['link_frame']) != 1].all()                  # later frames are never dropped
        rr, rf = make_ref_geometry(robot, field, dict(device='cpu', dtype=torch.float64))
        q = torch.zeros(4096, 7, dtype=torch.float64)
        q[:, 0] = torch.linspace(-3.2, 3.2, 4096)
        q[:, 1:] = 6.0 * torch.rand(4096, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) - 3.0
        pts = rr.fk_map_collision(q)                                          # (N, L, 3)
        sd = rf.signed_distance(pts)                                          # (N, L)
        hinge = (float(fs['margin']) + torch.as_tensor(rs['link_radius'], dtype=torch.float64) - sd).clamp_min(0)
        worst = hinge.max(0).values.numpy()
        assert (worst[~keep] == 0).all(), worst[~keep]
        buf = G.pack_geometry(robot, field)
        assert buf.view(np.int32)[5] == int(keep.sum())
    # the synthetic C3 scene keeps obstacles off the base axis: the three base-link spheres go
    assert int(G.links_that_can_touch(robot.spec(), G.env_spheres_3d(seed=0).spec()).sum()) == 28


def test_cost_term_specs_merge_into_one_launch():
    """CostComposite groups its trajectory-only members into as few mpb_cost_terms_eval launches as possible."""
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd.planners.costs import cost_functions as C
    robot = G.RobotPointMass(2, radius=0.01, dt=0.1)
    ta = dict(device='cpu', dtype=torch.float32)
    H = 16
    gp = C.CostGPTrajectory(robot, H, 0.1, sigma_gp=0.5, tensor_args=ta)
    gp_other_dt = C.CostGPTrajectory(robot, H, 0.2, sigma_gp=0.5, tensor_args=ta)
    sm = C.CostSmoothnessCHOMP(robot, H, tensor_args=ta)
    jl = C.CostJointLimits(robot, H, tensor_args=ta)
    coll = C.CostCollision(robot, H, field=G.env_dense_2d(), sigma_coll=0.1, tensor_args=ta)
    comp = C.CostComposite(robot, H, [coll, gp, sm, jl], weights_cost_l=[1.0, 2.0, 3.0, 4.0], tensor_args=ta)
    colls, groups, other = comp.device_plan('cpu')
    assert [c is coll for c, _ in colls] == [True] and not other and len(groups) == 1
    g = groups[0]
    assert g['terms'] == {'gp', 'smooth', 'jlim'}
    assert g['k_gp'] == pytest.approx(2.0 / 0.25) and g['k_smooth'] == 3.0 and g['k_jlim'] == 4.0 and g['dt'] == 0.1
    # same term twice, or two different dt: separate launches
    assert len(C.CostComposite(robot, H, [gp, gp], tensor_args=ta).device_plan('cpu')[1]) == 2
    assert len(C.CostComposite(robot, H, [sm, gp_other_dt], tensor_args=ta).device_plan('cpu')[1]) == 2
    # what the planners can fuse
    assert C.device_plan(comp, 'cpu') is not None
    # several collision members are chained into one evaluator: weight * k_0 * sum_f s_f cost_f
    coll2 = C.CostCollision(robot, H, field=G.env_grid_circles_2d(), sigma_coll=0.2, tensor_args=ta)
    plan = C.device_plan(C.CostComposite(robot, H, [coll, coll2], weights_cost_l=[2.0, 3.0], tensor_args=ta), 'cpu')
    merged, w0, groups0 = plan
    assert isinstance(merged, C.MergedCollision) and w0 == 2.0 and groups0 == []
    assert merged.k_sigma == pytest.approx(100.0) and merged.scales == pytest.approx([1.0, (3.0 * 25.0) / (2.0 * 100.0)])
    assert C.device_plan(C.CostComposite(robot, H, [coll] * 5, tensor_args=ta), 'cpu') is None      # more than four fields
    assert C.device_plan(lambda x: x, 'cpu') is None                                                   # user callable
    # no CPU fallback: evaluating on CPU tensors raises instead of computing
    with pytest.raises(ValueError, match='no CPU path'):
        gp(torch.zeros(2, H, 4))
    with pytest.raises(ValueError, match='no CPU path'):
        comp(torch.zeros(2, H, 4))


def test_oracle_trajectory_utilities():
    """Build-defined interpolation / resampling restatements: exact on straight lines, end points kept."""
    from oracle import planners_ref as O
    x = torch.stack([torch.linspace(0, 1, 5), torch.linspace(2, 0, 5)], -1)[None]           # (1,5,2) straight line
    xi = O.interpolate_trajs(x, 3)
    assert xi.shape == (1, 17, 2)
    assert torch.allclose(xi[0, :, 0], torch.linspace(0, 1, 17), atol=1e-6)
    assert torch.equal(xi[:, ::4], x)
    path = np.array([[0., 0.], [1., 0.], [1., 3.]])                                            # L-shaped, length 4
    out = O.resample_path(path, 9, 0.5).numpy()
    seg = np.linalg.norm(np.diff(out[:, :2], axis=0), axis=1)
    assert np.allclose(seg, 0.5)                                                               # uniform in arc length
    assert np.allclose(out[0, :2], path[0]) and np.allclose(out[-1, :2], path[-1])
    assert np.allclose(out[1:-1, 2:], (path[-1] - path[0]) / (8 * 0.5)) and np.all(out[[0, -1], 2:] == 0)


def test_chained_geometry_buffers():
    """Several collision fields in one buffer: header word 27 links them, word 28 carries the per-field scale;
    the C-ABI validator walks the chain."""
    from motion_planning_baselines_amd import geometry as G
    from motion_planning_baselines_amd import _lib
    from motion_planning_baselines_amd._lib import MPBError
    robot = G.RobotPointMass(2, radius=0.01)
    f1, f2 = G.env_dense_2d(), G.env_grid_circles_2d()
    one = G.pack_geometry(robot, f1)
    two = G.pack_geometry(robot, [f1, f2], scales=[1.0, 0.25])
    assert G.count_fields(one) == 1 and G.count_fields(two) == 2
    assert np.array_equal(two[:one.size].view(np.int32)[:27], one.view(np.int32)[:27])
    assert two.view(np.int32)[27] == one.size and two[one.size + 28] == np.float32(0.25) and two[28] == 1.0
    _lib.geom_check(one)
    _lib.geom_check(two)
    bad = two.copy()
    bad.view(np.int32)[27] = one.size - 4          # next header inside the first field
    with pytest.raises(MPBError):
        _lib.geom_check(bad)
    bad = two.copy()
    bad[one.size + 28] = -1.0                      # negative scale
    with pytest.raises(MPBError):
        _lib.geom_check(bad)
    with pytest.raises(MPBError):                  # second field for a different robot
        _lib.geom_check(np.concatenate([_link(one), G.pack_geometry(G.RobotPointMass(3, radius=0.01), G.env_spheres_3d())]))
    with pytest.raises(AssertionError):
        G.pack_geometry(robot, [f1] * 5)


def _link(buf):
    b = buf.copy()
    b.view(np.int32)[27] = b.size
    return b


def test_gp_prior_scale_tril_matches_reference():
    """Per-dof dense scale_tril from the bidiagonal factor == the sub-matrix of the reference's dense
    MultivariateNormal scale_tril (golden from MultiMPPrior, fp64)."""
    from conftest import load_golden
    from motion_planning_baselines_amd.planners.base import gp_prior_factor, gp_prior_scale_tril
    g = load_golden('gp_prior_d2_h8')
    D, H = int(g['D']), int(g['H'])
    Ud, Uo = gp_prior_factor(H, float(g['dt']), float(g['sigma_start']), float(g['sigma_gp']), float(g['sigma_goal']))
    Tm = gp_prior_scale_tril(Ud, Uo)
    assert Tm.shape == (2 * H, 2 * H) and np.allclose(np.triu(Tm, 1), 0.0)
    full = g['scale_tril'][0]                                 # (2D*H, 2D*H), index t*2D + c
    for d in range(D):
        idx = np.array([[t * 2 * D + d, t * 2 * D + D + d] for t in range(H)]).reshape(-1)
        sub = full[np.ix_(idx, idx)]
        np.testing.assert_allclose(Tm, sub, rtol=1e-9, atol=1e-12 * np.abs(sub).max())


def test_ctypes_signatures_match_the_header():
    """Every prototype of include/mpb.h, parameter by parameter, against the ctypes argtypes of _lib.SIGNATURES
    (a mismatch would not fail to load -- it would pass garbage)."""
    import ctypes
    from motion_planning_baselines_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'mpb.h')).read()
    hdr = re.sub(r'/\*.*?\*/', ' ', hdr, flags=re.S)
    protos = re.findall(r'\b(?:int|size_t|const char \*|const char\*)\s*\**\s*(mpb_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', hdr, flags=re.S)
    assert len(protos) == len(_lib.SIGNATURES)
    def ctype_of(param):
        param = ' '.join(param.split())
        if param in ('void', ''):
            return None
        if '*' in param:
            return ctypes.c_void_p
        base = param.rsplit(' ', 1)[0].replace('const ', '').strip()
        return {'int': ctypes.c_int, 'float': ctypes.c_float, 'uint64_t': ctypes.c_uint64, 'uint32_t': ctypes.c_uint32,
                'size_t': ctypes.c_size_t}[base]
    for name, params in protos:
        want = [t for t in (ctype_of(p) for p in params.split(',')) if t is not None]
        assert want == list(_lib.SIGNATURES[name]), (name, [t.__name__ for t in want], [t.__name__ for t in _lib.SIGNATURES[name]])


def test_robot_model_header_is_generated_from_geometry():
    """csrc/mpb_model_panda.h (the compile-time Panda tables the model kernels fold) is exactly what model_gen emits from
    geometry.py -- the single source of the numbers -- and the DH tables carry exact zeros / ones (what folds)."""
    import os
    from motion_planning_baselines_amd import geometry as G, model_gen
    path = os.path.join(os.path.dirname(G.__file__), 'csrc', 'mpb_model_panda.h')
    assert open(path).read() == model_gen.header_text('panda')
    tf = np.asarray(G.RobotPanda().spec()['joint_tf'])[:7, :, :3]
    assert set(np.unique(np.abs(tf))) <= {0.0, 1.0}


def test_geometry_model_tag_and_flags():
    """pack_geometry tags a buffer with the robot model only when the tables match bit for bit; mpb_geom_flags reports it
    (and the all-grids bit); mpb_geom_check refuses a forged tag."""
    from motion_planning_baselines_amd import _lib, geometry as G
    robot, field = G.RobotPanda(), G.env_spheres_3d()
    buf = G.pack_geometry(robot, field)
    gi = buf.view(np.int32)
    assert gi[29] == 1 and int(buf.view(np.uint32)[30]) == ((1 << 31) - 1) & ~0b111      # the three base spheres are pruned here
    _lib.geom_check(buf)
    # (bit 12: one field; bits 16-28: the cells of its broad-phase grid)
    n_cells = int(gi[26])
    assert _lib.geom_flags(buf) == (1 | 0x100 | 0x1000 | n_cells << 16)
    assert _lib.geom_flags(G.pack_geometry(robot, field, use_model=False)) == (0x100 | 0x1000 | n_cells << 16)
    full = G.pack_geometry(robot, field, prune_static=False)
    assert int(full.view(np.uint32)[30]) == (1 << 31) - 1 and _lib.geom_flags(full) == (1 | 0x100 | 0x1000 | n_cells << 16)
    two = G.pack_geometry(robot, [field, G.env_spheres_3d(seed=5)])
    assert not (_lib.geom_flags(two) & 0x1000) and (_lib.geom_flags(two) & 0x1FF) == (1 | 0x100)
    # a robot that is not the Panda bit for bit is not tagged
    other = G.RobotPanda()
    other.link_offset[5, 1] += 1e-4
    assert G.pack_geometry(other, field).view(np.int32)[29] == 0
    # a forged tag on that buffer is refused
    forged = G.pack_geometry(other, field, prune_static=False)
    forged.view(np.int32)[29] = 1
    forged.view(np.uint32)[30] = (1 << 31) - 1
    with pytest.raises(_lib.MPBError):
        _lib.geom_check(forged)
    # point robot: no model, grid usable; box-only field: no grid at all
    pm = G.pack_geometry(G.RobotPointMass(2, radius=0.01), G.env_grid_circles_2d())
    assert (_lib.geom_flags(pm) & 0xFFFF) == (0x100 | 0x400 | 0x1000)          # 49 circles: too many for the in-register CHOMP kernel (bit 9)
    assert _lib.geom_flags(pm) >> 16 == int(pm.view(np.int32)[26])
    assert (_lib.geom_flags(G.pack_geometry(G.RobotPointMass(2, radius=0.01), G.env_dense_2d())) & 0xFFFF) == (0x100 | 0x200 | 0x400 | 0x1000)
    boxes = G.CollisionField(boxes=np.array([[0.2, 0.2, 0.1, 0.1]], np.float32), margin=0.01)
    assert _lib.geom_flags(G.pack_geometry(G.RobotPointMass(2, radius=0.01), boxes)) == (0x200 | 0x400 | 0x1000)


def test_stomp_workspace_size():
    from motion_planning_baselines_amd import _lib
    f = _lib.lib().mpb_stomp_workspace_bytes
    # header + 8-byte {value, tag} granules, two parities: the H = 64 kernel's exchange slots hold 912 granules, the generalised
    # kernel's 2064 -- which also serves H = 64 when the geometry carries list grids (round 6): the workspace fits either
    assert f(128, 32, 64, 14) == 4 * 320 + 8 * (2 * 128 * 2 * 2064)
    assert f(3, 5, 64, 7) == 4 * 320 and f(0, 32, 64, 14) == 0              # one workgroup per particle: the header only
    assert f(4096, 32, 64, 14) == 4 * 320                                   # at least as many particles as CUs: two batches, no exchange
    assert f(8, 32, 48, 14) == 4 * 320 + 8 * (2 * 8 * 2 * 2064)            # H != 64: the generalised kernel's exchange slots (2 + H*d <= 2064 granules)
    assert f(8, 32, 200, 14) == 4 * 320                                     # a shape no persistent kernel serves (H > 128)
    # which path a call takes is a pure host-side function of the shape, the geometry flags and the workspace size
    path = _lib.lib().mpb_stomp_run_path
    assert path(0x100 | 1, f(128, 32, 64, 14), 128, 32, 64, 14) == 1       # exchange layout
    assert path(0x100 | 1, 4 * 320 + 8 * (2 * 128 * 2 * 912), 128, 32, 64, 14) == 1    # ... which needs the smaller slots only
    assert path(0x2000 | 1, f(128, 32, 64, 14), 128, 32, 64, 14) == 1 and path(0x2000 | 1, 1 << 30, 128, 32, 100, 14) == 0   # list grids: H <= 64
    assert path(0x100 | 1, 64, 128, 32, 64, 14) == 0                       # workspace too small: two-kernel loop
    assert path(0x100, 1280, 4096, 32, 64, 14) == 2 and path(0x100, 1280, 8, 16, 64, 7) == 2      # the header alone (MPB_STOMP_WS_HEADER_BYTES)
    assert path(0, 1 << 30, 128, 32, 64, 14) == 0 and path(0x100, 1 << 30, 128, 32, 200, 14) == 0
    assert path(0x100, 1 << 30, 128, 32, 128, 14) == 1 and path(0x100, 1280, 300, 32, 128, 14) == 2   # H = 128: generalised kernel
    assert path(0x100, 1 << 30, 2, 128, 32, 7) == 1 and path(0x100, 1 << 30, 128, 32, 64, 5) == 1  # S = 128; d = 5


@pytest.mark.parametrize('scene', ['spheres_3d', 'dense_2d', 'grid_circles_2d', 'large_3d'])
def test_broad_phase_grid_is_conservative_and_fits(scene):
    """geometry.build_grid: the grid fits the LDS image of the kernels (MPB_GRID_MAX_CELLS words), is refined below the
    coarse target when there is room, and is conservative -- every obstacle within reach (r_o + a_max) of a point is listed
    in the word of the point's cell (or the cell is marked overflow), with the cell index computed as the kernels do in fp32."""
    from motion_planning_baselines_amd import geometry as G
    rng = np.random.default_rng(5)
    if scene == 'spheres_3d':
        sph, a_max = G.env_spheres_3d().spheres, 0.13
    elif scene == 'dense_2d':
        sph, a_max = G.env_dense_2d().spheres, 0.02
    elif scene == 'grid_circles_2d':
        sph, a_max = G.env_grid_circles_2d().spheres, 0.015
    else:   # a scene too large for the coarse target cell: the builder has to coarsen
        c = rng.uniform(-6.0, 6.0, size=(40, 3))
        sph, a_max = np.concatenate([c, rng.uniform(0.1, 0.4, size=(40, 1))], 1).astype(np.float32), 0.13
    sph = np.asarray(sph, dtype=np.float32)
    g = G.build_grid(sph, a_max, planar=bool(np.ptp(sph[:, 2]) == 0.0))     # (pack_geometry passes planar for 2-D point robots)
    dims, lo, inv, words = g['dims'].astype(np.int64), g['lo'], g['inv'], g['words']
    n = len(sph)
    assert dims.prod() == len(words) <= G.GRID_MAX_CELLS and (dims >= 1).all() and (dims <= G.GRID_MAX_DIM).all()
    cell = 1.0 / inv.astype(np.float64)
    # geometry version 6: cells on a lattice through the origin, lo = (K - 1/2) h per axis
    K = g['K'].astype(np.int64)
    assert np.allclose(lo.astype(np.float64) * inv.astype(np.float64) + 0.5, K, atol=1e-3)
    assert g['k_lin'] == K[0] + dims[0] * (K[1] + dims[1] * K[2])
    if scene == 'large_3d':
        assert cell.max() > G.GRID_CELL                      # coarsened
    else:
        assert cell[dims > 1].max() <= G.GRID_CELL * 1.0001  # never coarser than the target when it fits
    planar = np.ptp(sph[:, 2]) == 0.0
    assert (dims[2] == 1) == bool(planar)
    # random points around the obstacles, cell index in fp32 exactly as grid_cell<true> (mpb_geom.h)
    k = rng.integers(0, n, size=20000)
    p = (sph[k, :3] + rng.normal(size=(20000, 3)).astype(np.float32) * (sph[k, 3:4] + a_max) * 1.2).astype(np.float32)
    if planar:
        p[:, 2] = sph[0, 2]
    f = np.floor(p * inv + (-lo * inv).astype(np.float32)).astype(np.float32)     # fma vs mul+add: inside the 1e-5 m slack
    f = np.clip(f, 0.0, (dims - 1).astype(np.float32))
    idx = ((f[:, 2] * np.float32(dims[1]) + f[:, 1]) * np.float32(dims[0]) + f[:, 0]).astype(np.int64)
    inside = ((p >= lo) & (p <= lo + dims / inv)).all(1)
    # ... and as grid_cell_rel does (persistent kernels, MPPI): fma(x, 1/h, 1.5 * 2^23) -- exact product, ONE rounding to the
    # integer grid of [2^23, 2^24), ties to even --, the linear index combined in the same float form, clamped
    MAGIC = 12582912.0
    t = (p.astype(np.float64) * inv.astype(np.float64) + MAGIC).astype(np.float32).astype(np.float64) - MAGIC      # round(x / h) per axis
    lin = t[:, 0] + dims[0] * (t[:, 1] + dims[1] * t[:, 2]) - g['k_lin']
    idx2 = np.clip(lin, 0, len(words) - 1).astype(np.int64)
    ins = ((t - K >= 0) & (t - K <= dims - 1)).all(1)                 # points whose cell lies inside the box on every axis
    assert (np.abs(idx2[ins & inside] - idx[ins & inside]) <= dims[0] * dims[1] + dims[0] + 1).all()
    for which, ii in (('floor', idx), ('lattice', idx2)):
        ww = words[ii]
        lst = np.stack([(ww >> (8 * s_)) & 0xFF for s_ in range(4)], 1)
        d_ = np.linalg.norm(p[:, None, :3].astype(np.float64) - sph[None, :, :3].astype(np.float64), axis=2)
        near_ = d_ < (sph[None, :, 3].astype(np.float64) + a_max)
        for i in np.nonzero(near_.any(1))[0]:
            if ww[i] == G.GRID_OVERFLOW:
                continue
            need = set(np.nonzero(near_[i])[0].tolist())
            assert need <= set(lst[i].tolist()), (scene, which, i, need, lst[i])
    w = words[idx]
    listed = np.stack([(w >> (8 * s)) & 0xFF for s in range(4)], 1)
    d = np.linalg.norm(p[:, None, :3].astype(np.float64) - sph[None, :, :3].astype(np.float64), axis=2)
    near = d < (sph[None, :, 3].astype(np.float64) + a_max)                       # obstacles that can matter at p
    assert not near[~inside].any()                           # outside the grid box nothing is within reach by construction
    for i in np.nonzero(near.any(1))[0]:
        if w[i] == G.GRID_OVERFLOW:
            continue
        need = set(np.nonzero(near[i])[0].tolist())
        assert need <= set(listed[i].tolist()), (scene, i, need, listed[i])
    # slots fill in order and empty slots hold n (the far dummy of the obstacle table)
    ok = w != G.GRID_OVERFLOW
    used = listed[ok] != n
    assert (listed[ok] <= n).all() and (used[:, 1:] <= used[:, :-1]).all()


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` started bare hands its N ranks to torch.distributed.run as fresh child processes BEFORE any
    GPU call, on 127.0.0.1, with the caller's own arguments, and returns the launcher's exit code."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen.update(cmd=cmd, env=env)
        return subprocess.CompletedProcess(cmd, 3)
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a, **k: (_ for _ in ()).throw(AssertionError('GPU touched before the launch')))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('MPB_DIST_BACKEND', raising=False)
    monkeypatch.setattr(bench, 'visible_gpu_count', lambda: 8)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '20', '--warmup', '5'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '20', '--warmup', '5'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_bench_self_launch_preflight(monkeypatch, capsys, tmp_path):
    """`python bench.py --gpus N` on a node that shows fewer than N GPUs says so and exits non-zero BEFORE
    it starts a rank (and without touching the GPU: the count comes from the KFD topology); a gloo rehearsal (several ranks on
    one GPU) and an unreadable topology are let through."""
    import subprocess
    import bench
    started = []
    monkeypatch.setattr(subprocess, 'run', lambda cmd, env=None, **kw: started.append(cmd) or subprocess.CompletedProcess(cmd, 0))
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a, **k: (_ for _ in ()).throw(AssertionError('GPU touched before the launch')))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('MPB_DIST_BACKEND', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    monkeypatch.setattr(bench, 'visible_gpu_count', lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and not started
    err = capsys.readouterr().err
    assert '--gpus 8' in err and '1 GPU' in err
    monkeypatch.setenv('MPB_DIST_BACKEND', 'gloo')                       # rehearsal: ranks share GPUs
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(started) == 1
    monkeypatch.delenv('MPB_DIST_BACKEND')
    monkeypatch.setattr(bench, 'visible_gpu_count', lambda: None)        # topology unreadable: the ranks will say what is wrong
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(started) == 2
    # the count itself: nodes with SIMDs only, cut down by a *_VISIBLE_DEVICES list
    import builtins
    import os as _os
    nodes = tmp_path / 'nodes'
    for i, simd in enumerate((0, 0, 256, 256, 256)):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / 'properties').write_text('cpu_cores_count %d\nsimd_count %d\n' % (0 if simd else 64, simd))
    monkeypatch.undo()
    real_listdir, real_open = _os.listdir, builtins.open
    root = '/sys/class/kfd/kfd/topology/nodes'
    monkeypatch.setattr(_os, 'listdir', lambda p: real_listdir(str(nodes)) if p == root else real_listdir(p))
    monkeypatch.setattr(builtins, 'open', lambda f, *a, **k: real_open(str(f).replace(root, str(nodes)), *a, **k))
    for v in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2')
    assert bench.visible_gpu_count() == 2


def test_headline_kernels_have_no_scratch():
    """The persistent STOMP kernels run at their register budget (16 waves per workgroup: 128 VGPRs): a change anywhere in
    their phases can tip loop invariants into scratch -- it happened twice in round 4, once unnoticed for several commits
    (-4 % on the headline).  build() records what the compiler made of every kernel (csrc/kernel_resources.json,
    -Rpass-analysis=kernel-resource-usage); the instantiations the bench line runs must have 0 B of scratch."""
    import json
    from motion_planning_baselines_amd import build
    build.build(verbose=False)
    res = json.load(open(build.RESOURCES))

    def find(prefix):
        hits = {k: v for k, v in res.items() if k.startswith(prefix)}
        assert len(hits) == 1, (prefix, list(hits))
        return next(iter(hits.values()))
    must = ['_Z18stomp_fused_kernelILi14ELi1ELi1ELb0ELb0EE', '_Z18stomp_fused_kernelILi14ELi1ELi2ELb0ELb0EE',   # C3 (exchange), C5 (two batches): device noise, ONE field
            '_Z18stomp_fused_kernelILi14ELi1ELi1ELb1ELb0EE', '_Z18stomp_fused_kernelILi14ELi1ELi2ELb1ELb0EE',   # ... and their injected-noise twins (parity tests)
            '_Z18stomp_fused_kernelILi14ELi1ELi1ELb0ELb1EE', '_Z18stomp_fused_kernelILi14ELi1ELi2ELb0ELb1EE',   # ... and the chained-field forms of all four
            '_Z18stomp_fused_kernelILi14ELi1ELi1ELb1ELb1EE',
            '_Z18stomp_fused_kernelILi7ELi1ELi1ELb0ELb0EE', '_Z18stomp_fused_kernelILi7ELi1ELi2ELb0ELb0EE',     # the same, pos_only
            '_Z18stomp_fused_kernelILi7ELi1ELi1ELb1ELb0EE', '_Z18stomp_fused_kernelILi7ELi1ELi2ELb1ELb0EE',
            '_Z18stomp_fused_kernelILi7ELi1ELi1ELb0ELb1EE',
            '_Z18stomp_fused_kernelILi7ELi1ELi1ELb1ELb1EE',
            '_Z21stomp_fused_hx_kernelILi14ELi1ELi2ELb0ELb0EE', '_Z21stomp_fused_hx_kernelILi14ELi1ELi2ELb1ELb0EE',   # H = 128: device / injected noise
            '_Z21stomp_fused_hx_kernelILi14ELi1ELi1ELb1ELb0EE',                                                       # H < 64, injected noise
            '_Z11mppi_kernelILi2ELb1ELb1ELb0EE', '_Z11mppi_kernelILi2ELb1ELb1ELb1EE']                     # the mppi entry (device / injected noise)
    for name in must:
        r = find(name)
        assert r['scratch'] == 0 and r['vgpr_spill'] == 0, (name, r)
        assert r['vgprs'] + r.get('agprs', 0) <= 128, (name, r)
    # ---- every OTHER kernel of the library against a per-kernel budget: 0 B unless it is listed here with
    #      the value it has today -- a regression anywhere fails on the CPU box, an improvement asks for the table to be tightened
    budget = {   # demangled-prefix -> bytes of scratch per lane
        # MPPI with the exhaustive obstacle walk (a scene without a usable grid) on the matrix path
        'mppi_kernel<0,false,true,true>': 20, 'mppi_kernel<0,false,false,true>': 8,
        'mppi_kernel<2,false,true,false>': 44, 'mppi_kernel<2,false,true,true>': 48,
        'gpmp2_solve_kernel<7,true,true>': 32,     # several collision fields AND the Sherman-Morrison form (256 registers + 7)
        # persistent STOMP, two batches per workgroup (P > 128): the table-driven walk, and the chained-field forms of the models
        # (mostly the injected-noise twins: only the parity tests of several fields at P > 128 come here)
        'stomp_fused_kernel<14,0,2,true,true>': 32, 'stomp_fused_kernel<7,0,2,true,true>': 32, 'stomp_fused_kernel<7,0,2,false,true>': 16,
        'stomp_fused_kernel<14,1,2,true,true>': 48, 'stomp_fused_kernel<7,1,2,true,true>': 36, 'stomp_fused_kernel<7,1,2,false,true>': 16,
        'stomp_fused_hx_kernel<0,0,1,true,false>': 8, 'stomp_fused_hx_kernel<0,0,2,true,false>': 120,   # run-time d, table-driven walk
        'stomp_fused_hx_kernel<0,0,2,false,false>': 68,
        'stomp_fused_hx_kernel<7,1,1,true,false>': 8,   # H < 64
        # list-grid instantiations (round 6: scenes beyond 63 obstacle spheres, H <= 64)
        'stomp_fused_hx_kernel<0,0,1,false,true>': 28, 'stomp_fused_hx_kernel<0,0,1,true,true>': 80,
        'stomp_fused_hx_kernel<14,1,1,false,true>': 8, 'stomp_fused_hx_kernel<14,1,1,true,true>': 8,
        'stomp_fused_hx_kernel<7,1,1,false,true>': 8, 'stomp_fused_hx_kernel<7,1,1,true,true>': 8,
    }

    def short(mangled):
        m = re.match(r'_Z\d+([a-z0-9_]+?)I(.*?)EEv', mangled)
        if not m:
            return mangled
        args = re.findall(r'L([ib])(\d+)E', m.group(2) + 'E')
        return '%s<%s>' % (m.group(1), ','.join(('true' if v == '1' else 'false') if t == 'b' else v for t, v in args))
    seen = set()
    for k, v in res.items():
        name = short(k)
        allowed = budget.get(name, 0)
        assert v.get('scratch', 0) <= allowed, (name, v.get('scratch'), 'budget', allowed)
        if name in budget:
            seen.add(name)
            assert v.get('scratch', 0) >= allowed // 2, ('tighten the budget of ' + name, v.get('scratch'), allowed)
    assert seen == set(budget), set(budget) - seen


def test_model_tag_requires_hinges_below_one():
    """The compile-time robot-model kernels take relu(hinge) as the [0, 1] clamp of the instruction encoding (csrc/mpb_geom.h, UNIT):
    pack_geometry tags a buffer with the model only while margin + largest collision sphere + deepest possible penetration < 1 m, and
    mpb_geom_check refuses a tagged buffer that violates it (a scene like that runs on the table-driven walk)."""
    from motion_planning_baselines_amd import geometry as G, _lib
    robot = G.RobotPanda()
    ok = G.pack_geometry(robot, G.env_spheres_3d())
    assert ok.view(np.int32)[29] != 0
    _lib.geom_check(ok)
    big = G.CollisionField(spheres=np.array([[0.5, 0.5, 0.5, 0.9], [-0.6, 0.2, 0.4, 0.1]], np.float32), margin=0.05)
    buf = G.pack_geometry(robot, big)
    assert buf.view(np.int32)[29] == 0                       # 0.05 + 0.08 + 0.9 >= 1: no model tag
    _lib.geom_check(buf)
    forged = buf.copy()
    forged.view(np.int32)[29] = ok.view(np.int32)[29]
    forged.view(np.uint32)[30] = ok.view(np.uint32)[30]
    with pytest.raises(_lib.MPBError):
        _lib.geom_check(forged)


def test_grid_lattice_far_and_negative_scenes():
    """Geometry version 6: the grid's origin sits on the lattice lo = (K - 1/2) h for scenes anywhere near the origin (negative K
    included) and mpb_geom_check verifies header word 31 against it; a scene too far away for fp32 to resolve its cells gets no grid."""
    from motion_planning_baselines_amd import geometry as G, _lib
    robot = G.RobotPointMass(3, radius=0.02)
    rng = np.random.default_rng(1)
    for shift in ([0, 0, 0], [-7.3, 2.1, -0.4], [55.0, -31.0, 12.0]):
        sph = np.concatenate([rng.uniform(-1, 1, (12, 3)) + np.array(shift), rng.uniform(0.05, 0.2, (12, 1))], 1).astype(np.float32)
        buf = G.pack_geometry(robot, G.CollisionField(spheres=sph, margin=0.03))
        gi = buf.view(np.int32)
        assert gi[26] > 0
        _lib.geom_check(buf)
        lo, inv = buf[20:23].astype(np.float64), buf[23:26].astype(np.float64)
        K = np.rint(lo * inv + 0.5)
        assert np.abs(lo * inv + 0.5 - K).max() < 1e-3 and gi[31] == int(K[0] + gi[17] * (K[1] + gi[18] * K[2]))
        bad = buf.copy()
        bad.view(np.int32)[31] += 1
        with pytest.raises(_lib.MPBError):
            _lib.geom_check(bad)
    far = np.array([[1.0e6, -1.0e6, 3.0e5, 0.1]], np.float32)
    buf = G.pack_geometry(robot, G.CollisionField(spheres=far, margin=0.03))
    assert buf.view(np.int32)[26] == 0 and not (_lib.geom_flags(buf) & 0x100)
    _lib.geom_check(buf)


def test_dpp_weighted_sum_keeps_its_wait_states(tmp_path):
    """The persistent STOMP kernels accumulate the softmax-weighted sum with v_fmac_f32_dpp ... row_newbcast:k, written as inline
    assembly (mpb_common.h, fmac_row_bcast_seq).  The compiler's hazard recogniser does not look into inline assembly and the
    hardware does not interlock a DPP read against a preceding vector write of its source (2 wait states) or of exec (5):
    the block opens with its own `s_nop 4`.  Checked on the code objects the library was linked from: every
    row_newbcast:0 fmac is directly preceded by that s_nop, the k-th fmac of a block by the (k-1)-th, and no fmac of a block
    writes the register the block reads through DPP."""
    import shutil
    import subprocess
    from motion_planning_baselines_amd import build
    build.build(verbose=False)
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not in this image')
    n_blocks = 0
    for src in ('mpb_stomp_fused', 'mpb_stomp_fused_hx'):
        obj = str(tmp_path / (src + '.o'))
        shutil.copy(os.path.join(build.CSRC, src + '.o'), obj)
        subprocess.run([objdump, '--offloading', obj], check=True, capture_output=True, cwd=str(tmp_path))
        co = [f for f in os.listdir(tmp_path) if f.startswith(src + '.o.') and 'gfx950' in f]
        assert len(co) == 1, co
        text = subprocess.run([objdump, '-d', str(tmp_path / co[0])], check=True, capture_output=True, text=True).stdout
        ins = [l.split('//')[0].strip() for l in text.splitlines() if l.startswith('\t')]
        for i, l in enumerate(ins):
            m = re.match(r'v_fmac_f32_dpp (v\d+), (v\d+), (v\d+) row_newbcast:(\d+) ', l)
            if not m:
                continue
            acc, a, _, k = m.group(1), m.group(2), m.group(3), int(m.group(4))
            assert acc != a, l
            if k == 0:
                n_blocks += 1
                assert ins[i - 1] == 's_nop 4', (src, ins[i - 3:i + 1])
            else:
                p = re.match(r'v_fmac_f32_dpp (v\d+), (v\d+), (v\d+) row_newbcast:(\d+) ', ins[i - 1])
                assert p and int(p.group(4)) == k - 1 and p.group(1) == acc and p.group(2) == a, (src, ins[i - 2:i + 1])
    assert n_blocks >= 20           # every instantiation of both kernels carries at least one block


def _big_scene(seed=0, n_sph=200, n_box=32):
    from motion_planning_baselines_amd import geometry as G
    return G.env_spheres_boxes_3d(seed, n_sph, n_box)


def test_list_grid_pack_check_flags_and_candidate_sets():
    """Geometry version 7 (round 6): a field with more than 63 obstacle spheres carries a LIST grid -- any number of candidates per
    cell, boxes culled like spheres.  pack_geometry builds it, mpb_geom_check accepts it (and refuses forged ranges), mpb_geom_flags
    reports bit 13 (and not bit 8: the compact-grid kernels must not take it), the persistent launcher takes it up to H = 64; and the
    candidate sets are CONSERVATIVE: for random query points every sphere / box within (margin + largest collision sphere) of the
    point is listed in the point's cell, found the way the kernels find it (cell = round(x / h) - K on the lattice)."""
    from motion_planning_baselines_amd import geometry as G, _lib
    robot, field = G.RobotPanda(), _big_scene()
    buf = G.pack_geometry(robot, field)
    gi = buf.view(np.int32)
    assert gi[1] == G.GEOM_VERSION_LIST and gi[6] == 200 and gi[7] == 32
    _lib.geom_check(buf)
    fl = _lib.geom_flags(buf)
    assert (fl & 0x2000) and not (fl & 0x100) and (fl >> 16) == gi[26]
    lib = _lib.lib()
    assert lib.mpb_stomp_run_path(fl, 1 << 30, 128, 32, 64, 14) != 0 and lib.mpb_stomp_run_path(fl, 1 << 30, 128, 32, 48, 7) != 0
    assert lib.mpb_stomp_run_path(fl, 1 << 30, 128, 32, 128, 14) == 0           # two horizon chunks leave no LDS for the tables
    # a compact-grid scene is untouched; two fields: one needs the list grid -> both take it
    small = G.env_spheres_3d()
    assert G.pack_geometry(robot, small).view(np.int32)[1] == G.GEOM_VERSION
    two = G.pack_geometry(robot, [small, field])
    assert two.view(np.int32)[1] == G.GEOM_VERSION_LIST and (_lib.geom_flags(two) & 0x2100) == 0x2000
    _lib.geom_check(two)
    # forged ranges are refused
    off_grid, n_cells, total = int(gi[16]), int(gi[26]), int(gi[13])
    off_cand = off_grid + (n_cells + 1023) // 1024 * 1024
    bad = buf.copy()
    w = bad.view(np.uint32)
    i = int(np.argmax((w[off_grid:off_grid + n_cells] >> 15) & 0x7F))
    w[off_grid + i] = (4 * (total - off_cand) - 1) | (5 << 15)                  # range runs past the candidate bytes
    with pytest.raises(_lib.MPBError):
        _lib.geom_check(bad)
    bad = buf.copy()
    bad.view(np.uint8)[4 * off_cand] = 250                                       # a sphere index that does not exist
    w = bad.view(np.uint32)
    w[off_grid] = 0 | (1 << 15)
    with pytest.raises(_lib.MPBError):
        _lib.geom_check(bad)
    # ---- conservative candidate sets
    fs, rs = field.spec(), robot.spec()
    a_max = float(fs['margin']) + float(np.max(rs['link_radius']))
    inv = buf[23:26].astype(np.float64)
    dims = gi[17:20].astype(np.int64)
    lo = buf[20:23].astype(np.float64)
    K = np.rint(lo * inv + 0.5).astype(np.int64)
    words = buf.view(np.uint32)[off_grid:off_grid + n_cells]
    cand = buf.view(np.uint8)[4 * off_cand:]
    rng = np.random.RandomState(1)
    pts = rng.uniform(-1.0, 1.0, (4000, 3))
    sph, box = fs['spheres'].astype(np.float64), fs['boxes'].astype(np.float64)
    checked = 0
    for p in pts:
        cell = np.rint(p * inv).astype(np.int64) - K
        d_s = np.linalg.norm(p - sph[:, :3], axis=1) - sph[:, 3]
        q = np.abs(p - box[:, :3]) - box[:, 3:6]
        d_b = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(1), 0)
        near_s, near_b = set(np.nonzero(d_s < a_max)[0]), set(np.nonzero(d_b < a_max)[0])
        if np.any(cell < 0) or np.any(cell >= dims):
            assert not near_s and not near_b, 'a point outside the grid must be beyond every threshold'
            continue
        wd = int(words[cell[0] + dims[0] * (cell[1] + dims[1] * cell[2])])
        if wd & 0x80000000:
            continue
        st, ns, nb = wd & 0x7FFF, (wd >> 15) & 0x7F, (wd >> 22) & 0x3F
        assert near_s <= set(cand[st:st + ns].tolist()), (p, near_s)
        assert near_b <= set(cand[st + ns:st + ns + nb].tolist()), (p, near_b)
        checked += bool(near_s or near_b)
    assert checked > 500


@pytest.mark.parametrize('offset_m', [0.0, 2500.0, 5000.0, 40000.0])
def test_scene_far_from_the_origin_packs_and_checks(offset_m):
    """mpb_geom_check accepted the grid origin only within an ABSOLUTE 1e-3 of a lattice point, which the fp32 rounding
    of lo and 1 / h exceeds beyond |K| ~ 16 700 cells (~2 km at C3's cell size) -- scenes that build_grid packs (it allows |K| + dims <
    65 536) were refused by the C check.  The tolerance now scales with |K|: whatever pack_geometry produces passes the check, with a
    grid where fp32 still resolves the cells and without one beyond."""
    from motion_planning_baselines_amd import geometry as G, _lib
    base = G.env_spheres_3d()
    sph = base.spheres.copy()
    sph[:, 0] += offset_m
    field = G.CollisionField(spheres=sph, margin=base.margin)
    buf = G.pack_geometry(G.RobotPanda(), field)
    _lib.geom_check(buf)                                   # must not raise
    n_cells = int(buf.view(np.int32)[26])
    if offset_m <= 5000.0:
        assert n_cells > 0 and (_lib.geom_flags(buf) & 0x100)
    else:
        assert n_cells == 0 and not (_lib.geom_flags(buf) & 0x100)      # too far for the fp32 cell index: the exhaustive evaluators serve it
