"""Geometry back-end specification: robots, collision primitives, packed device layout.

The reference delegates forward kinematics and signed-distance fields to the
un-vendored ``torch_robotics`` package (reference call sites:
mp_baselines/planners/costs/cost_functions.py:50-52 ``robot.get_position /
get_velocity / fk_map_collision`` and costs/factors/field_factor.py:39
``field.compute_cost``).  That package is not part of the reference tree, so the
geometry here is BUILD-DEFINED (parity with torch_robotics is unpinned, see
DESIGN.md).  This module holds only *data*: the analytic definitions are
evaluated on the GPU by csrc/mpb_collision.h and, for checking, on the CPU by
oracle/geometry_ref.py from the very same arrays.

Definitions (all fp32):
  * point robot, D in {2,3}: one collision sphere at (q0, q1, q2|0), radius r.
  * serial revolute chain (Panda): frame_0 = I; frame_{j+1} = frame_j * P_j * Rz(q_j)
    for j < D; an optional fixed tool frame_{D+1} = frame_D * P_D.  P_j are constant
    3x4 transforms (modified-DH: Rx(alpha_{j-1}) Tx(a_{j-1}) Tz(d_j)).
    Collision sphere l sits at frame_{link_frame[l]} * offset_l with radius r_l.
  * obstacle sphere: sdf(x) = |x - c| - r.
  * axis-aligned box: q = |x - c| - h; sdf = |max(q,0)| + min(max(qx,qy,qz), 0).
  * per-waypoint collision cost: sum_l relu(margin + r_l - min_o sdf_o(x_l)).
"""
import math
import os

import numpy as np
import torch

GEOM_MAGIC = 0x4D504247  # 'MPBG'
GEOM_VERSION = 6
MAX_FIELDS = 4           # collision fields chained in one buffer (csrc/mpb_geom.h MPB_MAX_FIELDS)
GEOM_HEADER_WORDS = 32
GRID_MAX_DIM = 64       # cells per axis of the broad-phase grid
GRID_MAX_CELLS = 4096   # = MPB_GRID_MAX_CELLS of csrc/mpb_geom.h: 16 KB of LDS
GRID_CELL = 0.14        # coarsest target cell edge [m]; refined by GRID_REFINE steps down to GRID_CELL_MIN while the grid fits
GRID_CELL_MIN = 0.05
GRID_REFINE = 0.97
GRID_PAD = 1024         # the grid section is padded to a multiple of this many words (one round of a 256-thread block's uint4 loads)
GRID_OVERFLOW = 0xFFFFFFFE  # more than 4 candidates in the cell: the kernel tests every obstacle
# geometry version 7 (round 6): fields whose obstacle set does not fit the compact grid (more than 63 spheres) carry a LIST grid --
# a cell word holds (start, sphere count, box count) into a byte array of candidate indices that follows the cell words: any number of
# candidates per cell, boxes culled like spheres (csrc/mpb_geom.h, spheres_hinge_list).  Limits = what the persistent kernel keeps in LDS
GEOM_VERSION_LIST = 7
LIST_MAX_SPH = 255        # 8-bit candidate indices; the kernels' sphere table holds 255 + the far dummy
LIST_MAX_BOX = 127
LIST_MAX_CAND = 16384     # bytes of candidate indices
LIST_CELL_MAX_SPH = 126   # per cell (7-bit field; 127 marks an overflowing cell: exhaustive loop)
LIST_CELL_MAX_BOX = 62    # per cell (6-bit field; 63 marks overflow)
LIST_OVERFLOW = 0x80000000
KIND_POINT = 0
KIND_CHAIN = 1
MAX_DOF = 12
BOX_2D_HALF_Z = 1.0e6  # 2-D boxes are 3-D boxes that are effectively infinite in z


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def _mdh(alpha, a, d):
    """Modified-DH constant part: Rx(alpha) * Tx(a) * Tz(d) as a 3x4 matrix."""
    R = _rot_x(alpha)
    t = R @ np.array([a, 0.0, d])
    # Rx(alpha) Tx(a) Tz(d): translation = Rx * (a,0,d)  (Tx and Tz act in the rotated frame)
    out = np.zeros((3, 4), dtype=np.float64)
    out[:, :3] = R
    out[:, 3] = t
    # cos(+-pi/2) evaluates to 6.1e-17, not 0: the link twists of a modified-DH table are exact quarter turns, so such
    # residues are rounding noise of the table's construction -- snapped to exact zeros (the compile-time robot models
    # of model_gen.py then fold those terms away, and the generic kernels / the oracle see the same exact tables)
    out[np.abs(out) < 1e-12] = 0.0
    return out


class Robot:
    """Common robot interface mirroring what the reference consumes
    (cost_functions.py:21 ``q_dim``; :50-51 ``get_position``/``get_velocity``;
    :412-420 ``q_min``/``q_max``; :380 ``dt``)."""

    kind = None

    def __init__(self, q_dim, q_min, q_max, dt=None):
        self.q_dim = int(q_dim)
        self.q_min_np = np.asarray(q_min, dtype=np.float32)
        self.q_max_np = np.asarray(q_max, dtype=np.float32)
        self.dt = dt

    @property
    def q_min(self):
        return torch.from_numpy(self.q_min_np)

    @property
    def q_max(self):
        return torch.from_numpy(self.q_max_np)

    # state slicing -- same meaning as torch_robotics' RobotBase (positions first, velocities last)
    def get_position(self, x):
        return x[..., :self.q_dim]

    def get_velocity(self, x):
        return x[..., self.q_dim:2 * self.q_dim]

    def spec(self):
        raise NotImplementedError


class RobotPointMass(Robot):
    """Point-mass robot in 2-D or 3-D with one collision sphere of radius ``radius``."""

    kind = KIND_POINT

    def __init__(self, q_dim=2, radius=0.01, q_limits=(-1.0, 1.0), dt=None):
        super().__init__(q_dim, [q_limits[0]] * q_dim, [q_limits[1]] * q_dim, dt=dt)
        assert q_dim in (2, 3)
        self.radius = float(radius)

    def spec(self):
        return dict(
            kind=KIND_POINT, n_dof=self.q_dim,
            joint_tf=np.zeros((0, 3, 4), np.float32),
            link_frame=np.zeros((1,), np.int32),
            link_offset=np.zeros((1, 3), np.float32),
            link_radius=np.array([self.radius], np.float32),
        )


class RobotSerialChain(Robot):
    """Serial chain of revolute joints with a fixed set of collision spheres."""

    kind = KIND_CHAIN

    def __init__(self, joint_tf, link_frame, link_offset, link_radius, q_min, q_max, dt=None):
        joint_tf = np.asarray(joint_tf, dtype=np.float32)
        n_dof = len(q_min)
        assert joint_tf.shape in ((n_dof, 3, 4), (n_dof + 1, 3, 4))
        if joint_tf.shape[0] == n_dof:  # no tool frame: append identity
            eye = np.zeros((1, 3, 4), np.float32)
            eye[0, :, :3] = np.eye(3)
            joint_tf = np.concatenate([joint_tf, eye], 0)
        assert n_dof <= MAX_DOF
        super().__init__(n_dof, q_min, q_max, dt=dt)
        self.joint_tf = joint_tf
        self.link_frame = np.asarray(link_frame, dtype=np.int32)
        self.link_offset = np.asarray(link_offset, dtype=np.float32).reshape(-1, 3)
        self.link_radius = np.asarray(link_radius, dtype=np.float32)
        assert self.link_frame.min() >= 1 and self.link_frame.max() <= n_dof + 1
        # kernels walk the chain once and emit spheres frame by frame
        assert np.all(np.diff(self.link_frame) >= 0), "collision spheres must be sorted by frame"

    def spec(self):
        return dict(
            kind=KIND_CHAIN, n_dof=self.q_dim, joint_tf=self.joint_tf,
            link_frame=self.link_frame, link_offset=self.link_offset, link_radius=self.link_radius,
        )


# Franka Emika Panda, public modified-DH table (Franka Control Interface documentation):
#   joint : a_{i-1}   d_i    alpha_{i-1}
_PANDA_MDH = [
    (0.0, 0.333, 0.0),
    (0.0, 0.0, -math.pi / 2),
    (0.0, 0.316, math.pi / 2),
    (0.0825, 0.0, math.pi / 2),
    (-0.0825, 0.384, -math.pi / 2),
    (0.0, 0.0, math.pi / 2),
    (0.088, 0.0, math.pi / 2),
]
_PANDA_FLANGE_D = 0.107
_PANDA_Q_MIN = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]
_PANDA_Q_MAX = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]

# Build-defined collision-sphere set: (frame, x, y, z, radius); frame j = link j (after joint j),
# frame 8 = flange + hand (Tz(0.107) Rz(-pi/4)).
_PANDA_SPHERES = [
    (1, 0.0, 0.0, -0.20, 0.08), (1, 0.0, 0.0, -0.10, 0.08), (1, 0.0, -0.03, 0.0, 0.08),
    (2, 0.0, 0.0, 0.03, 0.08), (2, 0.0, -0.07, 0.0, 0.07), (2, 0.0, -0.14, 0.0, 0.07),
    (3, 0.0, 0.0, -0.12, 0.07), (3, 0.0, 0.0, -0.05, 0.07), (3, 0.04, 0.0, -0.02, 0.07),
    (3, 0.0825, 0.03, 0.0, 0.07),
    (4, 0.0, 0.0, 0.03, 0.07), (4, -0.04, 0.02, 0.0, 0.07), (4, -0.0825, 0.06, 0.0, 0.07),
    (4, -0.0825, 0.10, 0.02, 0.06),
    (5, 0.0, 0.0, -0.26, 0.07), (5, 0.0, 0.04, -0.20, 0.06), (5, 0.0, 0.07, -0.14, 0.055),
    (5, 0.0, 0.08, -0.08, 0.055), (5, 0.0, 0.05, -0.02, 0.06), (5, 0.0, 0.0, 0.0, 0.07),
    (6, 0.0, 0.0, 0.01, 0.07), (6, 0.06, 0.0, 0.01, 0.06), (6, 0.088, 0.02, 0.0, 0.06),
    (7, 0.0, 0.0, 0.07, 0.06), (7, 0.03, 0.03, 0.09, 0.045), (7, -0.03, -0.03, 0.09, 0.045),
    (8, 0.0, 0.0, 0.03, 0.06), (8, 0.0, 0.06, 0.04, 0.045), (8, 0.0, -0.06, 0.04, 0.045),
    (8, 0.0, 0.04, 0.09, 0.03), (8, 0.0, -0.04, 0.09, 0.03),
]


class RobotPanda(RobotSerialChain):
    """7-DoF Franka Panda: public DH kinematics + a build-defined set of 31 collision spheres."""

    def __init__(self, dt=None):
        tfs = [_mdh(alpha, a, d) for (a, d, alpha) in _PANDA_MDH]
        tool = np.zeros((3, 4))
        tool[:, :3] = _rot_z(-math.pi / 4)
        tool[:, 3] = [0.0, 0.0, _PANDA_FLANGE_D]
        tfs.append(tool)
        sph = np.array(_PANDA_SPHERES, dtype=np.float64)
        super().__init__(
            joint_tf=np.stack(tfs), link_frame=sph[:, 0].astype(np.int32),
            link_offset=sph[:, 1:4], link_radius=sph[:, 4],
            q_min=_PANDA_Q_MIN, q_max=_PANDA_Q_MAX, dt=dt)


class CollisionField:
    """Set of sphere / axis-aligned-box obstacles plus the hinge margin.

    Stands in for torch_robotics' collision distance fields (reference call site
    field_factor.py:39 ``field.compute_cost(q_pos, link_pos, **kwargs)``).
    """

    def __init__(self, spheres=None, boxes=None, margin=0.05):
        def _arr(x, w):
            a = np.zeros((0, w), np.float32) if x is None else np.asarray(x, dtype=np.float32)
            return a.reshape(-1, w)
        sp = np.asarray(spheres, np.float32) if spheres is not None and len(spheres) else None
        if sp is not None and sp.shape[-1] == 3:  # 2-D circles (cx, cy, r)
            sp = np.concatenate([sp[:, :2], np.zeros((len(sp), 1), np.float32), sp[:, 2:3]], 1)
        bx = np.asarray(boxes, np.float32) if boxes is not None and len(boxes) else None
        if bx is not None and bx.shape[-1] == 4:  # 2-D boxes (cx, cy, hx, hy)
            n = len(bx)
            bx = np.concatenate([bx[:, :2], np.zeros((n, 1), np.float32), bx[:, 2:4],
                                 np.full((n, 1), BOX_2D_HALF_Z, np.float32)], 1)
        self.spheres = _arr(sp, 4)
        self.boxes = _arr(bx, 6)
        self.margin = float(margin)
        assert len(self.spheres) + len(self.boxes) > 0, "empty collision field"

    def spec(self):
        return dict(spheres=self.spheres, boxes=self.boxes, margin=np.float32(self.margin))

    def zero_grad(self):  # reference calls field.zero_grad() (field_factor.py:56); nothing to clear here
        pass


def _fit_axis(a, b, edge):
    """(n, 1/h as fp32, K): fewest cells of a lattice-aligned edge <= `edge` that cover [a, b] (cells on a lattice through the world
    origin: cell ix of an axis is [(K + ix - 1/2) h, (K + ix + 1/2) h), geometry version 6)"""
    n = max(int(np.ceil((b - a) / edge)), 1)
    while n <= GRID_MAX_DIM:
        h = (b - a) / n
        while h <= edge * (1.0 + 1e-9):
            inv32 = np.float32(1.0 / h)
            he = 1.0 / float(inv32)                      # the edge the kernels effectively use
            K = int(np.floor(a / he + 0.5))              # (K - 1/2) he <= a
            if (K - 0.5 + n) * he >= b:
                return n, inv32, K
            h *= 1.0 + 2e-4
        n += 1
    return None


def build_list_grid(spheres, boxes, a_max, slack=1e-4, planar=False):
    """Broad-phase LIST grid (geometry version 7) over the inflated obstacle spheres AND boxes, for scenes beyond the compact grid's 63
    spheres: per cell the indices of every sphere / box a collision sphere of radius <= a_max - margin centred in the cell could be
    within its hinge threshold of (conservative, fp64).  Cell words: bits 0-14 start into the candidate bytes, 15-21 sphere count,
    22-27 box count (boxes follow the spheres), bit 31 = the cell overflows a count field (the kernels then test every obstacle).
    The finest lattice cell (from GRID_CELL_MIN up) whose grid fits GRID_MAX_CELLS words and LIST_MAX_CAND candidate bytes.
    None when the scene does not fit the kernels' tables (LIST_MAX_SPH / LIST_MAX_BOX) or no cell size fits."""
    ns, nb = len(spheres), len(boxes)
    if ns + nb == 0 or ns > LIST_MAX_SPH or nb > LIST_MAX_BOX:
        return None
    c = spheres[:, :3].astype(np.float64).reshape(-1, 3)
    R = spheres[:, 3].astype(np.float64).reshape(-1) + a_max + slack
    bc = boxes[:, 0:3].astype(np.float64).reshape(-1, 3)
    bh = boxes[:, 3:6].astype(np.float64).reshape(-1, 3)
    los = [c - R[:, None]] if ns else []
    his = [c + R[:, None]] if ns else []
    if nb:
        los.append(bc - bh - (a_max + slack))
        his.append(bc + bh + (a_max + slack))
    lo, hi = np.concatenate(los).min(0), np.concatenate(his).max(0)
    if planar:                                             # 2-D robots query z = 0 only: one layer of cells (2-D boxes are "infinite" in z)
        lo[2], hi[2] = -0.5 * GRID_CELL, 0.5 * GRID_CELL
    if np.any(hi - lo > 1.0e5):
        return None
    pad = 1e-5 + 1e-6 * float(np.abs(np.concatenate([lo, hi])).max())     # the kernels take the cell from fp32 arithmetic on x
    edge = GRID_CELL_MIN
    while edge < 4.0:
        fits = [(1, np.float32(1.0 / GRID_CELL), 0) if (planar and ax == 2) else _fit_axis(float(lo[ax]), float(hi[ax]), edge) for ax in range(3)]
        if all(f is not None for f in fits) and int(np.prod([f[0] for f in fits])) <= GRID_MAX_CELLS:
            dims = np.array([f[0] for f in fits], dtype=np.int64)
            inv32 = np.array([f[1] for f in fits], dtype=np.float32)
            K = np.array([f[2] for f in fits], dtype=np.int64)
            cell_eff = 1.0 / inv32.astype(np.float64)
            if not (np.abs(K).max() + dims.max() >= (1 << 16) or abs(int(K[0] + dims[0] * (K[1] + dims[1] * K[2]))) + int(dims.prod()) >= (1 << 21)):
                lo_s = (K - 0.5) * cell_eff
                ix = [np.arange(d) for d in dims]
                X, Y, Z = np.meshgrid(ix[0], ix[1], ix[2], indexing='ij')
                cmin = lo_s + np.stack([X, Y, Z], -1) * cell_eff          # (nx, ny, nz, 3)
                cmax = cmin + cell_eff
                hit_s = np.zeros((*dims, ns), dtype=bool)
                for o in range(ns):
                    d = np.maximum(np.maximum(cmin - c[o], c[o] - cmax), 0.0)
                    hit_s[..., o] = (d * d).sum(-1) < (R[o] + pad) ** 2
                hit_b = np.zeros((*dims, nb), dtype=bool)
                for o in range(nb):                                           # distance between the cell and the box, both axis aligned
                    d = np.maximum(np.maximum(cmin - (bc[o] + bh[o]), (bc[o] - bh[o]) - cmax), 0.0)
                    hit_b[..., o] = (d * d).sum(-1) < (a_max + slack + pad) ** 2
                cs, cb = hit_s.sum(-1), hit_b.sum(-1)
                over = (cs > LIST_CELL_MAX_SPH) | (cb > LIST_CELL_MAX_BOX)
                total = int((cs + cb)[~over].sum())
                if total <= LIST_MAX_CAND and total < (1 << 15):
                    # cell words in x-fastest order, candidates in ascending index order (spheres, then boxes)
                    order = (2, 1, 0)
                    ncl = int(dims.prod())
                    hs = np.ascontiguousarray(hit_s.transpose(*order, 3)).reshape(ncl, ns)
                    hbx = np.ascontiguousarray(hit_b.transpose(*order, 3)).reshape(ncl, nb)
                    ov = np.ascontiguousarray(over.transpose(*order)).reshape(-1)
                    words = np.zeros(hs.shape[0], dtype=np.uint32)
                    cand = []
                    pos = 0
                    for i in range(hs.shape[0]):
                        if ov[i]:
                            words[i] = LIST_OVERFLOW
                            continue
                        si, bi = np.nonzero(hs[i])[0], np.nonzero(hbx[i])[0]
                        words[i] = pos | (len(si) << 15) | (len(bi) << 22)
                        cand.extend(si.tolist())
                        cand.extend(bi.tolist())
                        pos += len(si) + len(bi)
                    cand = np.asarray(cand + [0] * ((-len(cand)) % 16 or (16 if not cand else 0)), dtype=np.uint8)
                    k_lin = int(K[0] + dims[0] * (K[1] + dims[1] * K[2]))
                    return dict(dims=dims.astype(np.int32), lo=lo_s.astype(np.float32), inv=inv32, words=words, cand=cand, k_lin=k_lin, K=K,
                                stats=dict(cell=float(cell_eff.max()), mean_sph=float(cs.mean()), max_sph=int(cs.max()), mean_box=float(cb.mean()),
                                           max_box=int(cb.max()), overflow=int(over.sum()), cand_bytes=total))
        edge /= GRID_REFINE
    return None


def build_grid(spheres, a_max, slack=1e-4, planar=False):
    """Uniform broad-phase grid over the inflated obstacle spheres (host, fp64).  None when there are no
    spheres or more than 254 of them (8-bit indices)."""
    n = len(spheres)
    if n == 0 or n > 254:
        return None
    c = spheres[:, :3].astype(np.float64)
    R = spheres[:, 3].astype(np.float64) + a_max + slack      # a sphere at x can matter only if |x - c| < R
    lo = (c - R[:, None]).min(0)
    hi = (c + R[:, None]).max(0)
    # Cells on a LATTICE through the world origin (geometry version 6): cell ix of an axis is [(K + ix - 1/2) h, (K + ix + 1/2) h)
    # with integer K, i.e. round-to-nearest(x / h) - K -- which a kernel gets from ONE fma(x, 1/h, 1.5 * 2^23) (the integer lands
    # in the low mantissa bits) where floor((x - lo) / h) took an fma with three register sources and a convert
    # (csrc/mpb_geom.h, grid_cell_rel).  Per axis the smallest cell count n whose lattice-aligned edge h (the smallest h >=
    # extent / n for which some lattice position covers [lo, hi]) does not exceed the target edge; the target goes from
    # GRID_CELL down to GRID_CELL_MIN while the grid still fits GRID_MAX_CELLS words of LDS: finer cells list fewer
    # obstacles each, and the kernels' candidate loop runs max-over-the-wave(candidates) times (C3: 17 x 15 x 16 cells of
    # ~0.12 m -> 1.27 trips per group of four collision spheres); planar problems get one layer of cells along z.
    flat = [planar and ax == 2 and np.ptp(c[:, ax]) == 0.0 for ax in range(3)]     # 2-D robots: every query point has z = 0

    def fit_axis(a, b, edge):
        """(n, h32, K): fewest cells of a lattice-aligned edge <= `edge` that cover [a, b]"""
        n = max(int(np.ceil((b - a) / edge)), 1)
        while n <= GRID_MAX_DIM:
            h = (b - a) / n
            while h <= edge * (1.0 + 1e-9):
                inv32 = np.float32(1.0 / h)
                he = 1.0 / float(inv32)                      # the edge the kernels effectively use
                K = int(np.floor(a / he + 0.5))              # (K - 1/2) he <= a
                if (K - 0.5 + n) * he >= b:
                    return n, inv32, K
                h *= 1.0 + 2e-4
            n += 1
        return None

    def lattice(edge):
        fits = []
        for ax in range(3):
            if flat[ax]:                                     # one layer of cells, the one that holds the plane z = 0
                fits.append((1, np.float32(1.0 / edge), 0))
                continue
            f = fit_axis(float(lo[ax]), float(hi[ax]), edge)
            if f is None:
                return None
            fits.append(f)
        return fits
    edge = GRID_CELL
    fits = lattice(edge)
    while edge * GRID_REFINE >= GRID_CELL_MIN:
        t = lattice(edge * GRID_REFINE)
        if t is None or int(np.prod([f[0] for f in t])) > GRID_MAX_CELLS:
            break
        edge *= GRID_REFINE
        fits = t
    while fits is None or int(np.prod([f[0] for f in fits])) > GRID_MAX_CELLS:      # (GRID_CELL itself too fine for a very large scene: coarsen)
        edge /= GRID_REFINE
        fits = lattice(edge)
    dims = np.array([f[0] for f in fits], dtype=np.int64)
    inv32 = np.array([f[1] for f in fits], dtype=np.float32)
    K = np.array([f[2] for f in fits], dtype=np.int64)
    cell_eff = 1.0 / inv32.astype(np.float64)
    # the kernels hold round(x / h) and the linear lattice index exactly in fp32 (integers below 2^22) and resolve a cell at
    # |x| / h only while fp32 does: a scene that far from the origin gets no grid (the exhaustive evaluators serve it)
    if np.abs(K).max() + dims.max() >= (1 << 16) or abs(int(K[0] + dims[0] * (K[1] + dims[1] * K[2]))) + int(dims.prod()) >= (1 << 21):
        return None
    lo_s = (K - 0.5) * cell_eff
    lo32 = lo_s.astype(np.float32)
    # the kernels take the cell from fp32 arithmetic on x: a point within ~1e-5 of a cell face may land in the neighbour --
    # added to R, with the fp32 rounding of large coordinates
    Rg = R + 1e-5 + 1e-6 * np.abs(np.concatenate([lo, hi])).max()
    ix = [np.arange(d) for d in dims]
    X, Y, Z = np.meshgrid(ix[0], ix[1], ix[2], indexing='ij')
    cmin = lo_s + np.stack([X, Y, Z], -1) * cell_eff          # (nx,ny,nz,3)
    cmax = cmin + cell_eff
    counts = np.zeros(dims, dtype=np.int64)
    lists = np.full((*dims, 4), n, dtype=np.uint32)      # empty slot = n: the far dummy the kernels append to the table
    for o in range(n):
        d = np.maximum(np.maximum(cmin - c[o], c[o] - cmax), 0.0)
        hit = (d * d).sum(-1) < Rg[o] ** 2
        idx = np.nonzero(hit)
        k = counts[idx]
        ok = k < 4
        sel = tuple(a[ok] for a in idx)
        lists[sel + (k[ok],)] = o
        counts[idx] += 1
    w = lists[..., 0] | (lists[..., 1] << 8) | (lists[..., 2] << 16) | (lists[..., 3] << 24)
    w = np.where(counts > 4, np.uint32(GRID_OVERFLOW), w.astype(np.uint32))
    words = np.ascontiguousarray(w.transpose(2, 1, 0)).reshape(-1).astype(np.uint32)   # x fastest
    k_lin = int(K[0] + dims[0] * (K[1] + dims[1] * K[2]))     # linear index of the lattice point (Kx, Ky, Kz): header word 31
    return dict(dims=dims.astype(np.int32), lo=lo32, inv=inv32, words=words, k_lin=k_lin, K=K,
                stats=dict(mean=float(counts.mean()), max=int(counts.max()), overflow=int((counts > 4).sum())))


def links_that_can_touch(rs, fs, slack=1e-4):
    """Static broad phase at pack time (serial chains): a collision sphere riding on frame 1 moves on a circle about the
    first joint axis whatever the trajectory does; when that circle keeps farther than margin + r_link (+ slack) from
    every obstacle its hinge is exactly zero for every configuration, and the sphere can be left out of the link table
    -- cost and gradient are unchanged bit for bit (x + 0).  Base links are the typical case (nothing is placed inside
    the robot's pedestal).  Returns a boolean keep-mask over the robot's collision spheres; spheres on later frames are
    always kept (their reach depends on several joints)."""
    n_links = len(rs['link_radius'])
    keep = np.ones(n_links, dtype=bool)
    if rs['kind'] != KIND_CHAIN:
        return keep
    P0 = np.asarray(rs['joint_tf'][0], dtype=np.float64)            # frame 1 = P0 * Rz(q0)
    R0, t0 = P0[:, :3], P0[:, 3]
    n = R0[:, 2]                                                     # joint axis in the world
    sph = np.asarray(fs['spheres'], dtype=np.float64).reshape(-1, 4)
    box = np.asarray(fs['boxes'], dtype=np.float64).reshape(-1, 6)
    cen = np.concatenate([sph[:, :3], box[:, :3]], 0)
    rad = np.concatenate([sph[:, 3], np.linalg.norm(box[:, 3:6], axis=1)], 0)     # boxes: bounding sphere
    for l in range(n_links):
        if int(rs['link_frame'][l]) != 1:
            continue
        o = np.asarray(rs['link_offset'][l], dtype=np.float64)
        a0 = t0 + R0 @ np.array([0.0, 0.0, o[2]])                   # centre of the circle
        rho = float(np.hypot(o[0], o[1]))
        v = cen - a0
        h = v @ n
        rperp = np.linalg.norm(v - h[:, None] * n[None, :], axis=1)
        dist = np.sqrt((rperp - rho) ** 2 + h ** 2)                 # obstacle centre to the circle
        thr = float(fs['margin']) + float(rs['link_radius'][l]) + slack
        if np.all(dist - rad > thr):
            keep[l] = False
    if not keep.any():
        keep[-1] = True                                              # the kernels want at least one link
    return keep


def hinge_bound(rs, fs):
    """Upper bound of a hinge margin + r_l - sdf of the scene: the signed distance to a sphere is >= -radius, to a box >= -(its
    smallest half extent)."""
    deepest = 0.0
    if len(fs['spheres']):
        deepest = max(deepest, float(np.max(np.asarray(fs['spheres'])[:, 3])))
    if len(fs['boxes']):
        deepest = max(deepest, float(np.max(np.min(np.asarray(fs['boxes'])[:, 3:6], axis=1))))
    return float(fs['margin']) + float(np.max(rs['link_radius'])) + deepest


def field_needs_list_grid(field):
    """The compact broad-phase grid (geometry version 6) serves up to 63 obstacle spheres (csrc/mpb_geom.h MPB_GRID_MAX_SPH); beyond
    that a field takes the list grid of version 7 where it fits."""
    return len(field.spec()['spheres']) > 63


def pack_geometry(robot, field, scales=None, prune_static=True, use_model=True, list_grid=None):
    """Pack robot + collision field(s) into the flat fp32 word buffer the HIP kernels read.

    use_model: tag the buffer with the compile-time robot model its tables equal (model_gen.py), which lets the
    kernels take their unrolled chain walk; False keeps the generic table-driven walk (same bits; tests compare them).

    prune_static: leave out the collision spheres that can never come within their hinge threshold of this field's
    obstacles (links_that_can_touch); the per-sphere entry points (mpb_fk_collision_points, mpb_field_cost_points) need
    the full table: pack with prune_static=False for them.

    `field` may be a list of up to MAX_FIELDS CollisionFields (the reference builds one CostCollision per field,
    gpmp2.py:70-78, and sums them): the per-field buffers are chained, header word [27] of each holding the word
    offset to the next one (0 = last) and word [28] a per-field scale s_f -- every kernel evaluates
    sum_f s_f * cost_f(q) (and its gradient) by walking the chain.

    Layout (32-bit words; ints stored bit-exact), mirrored by csrc/mpb_geom.h:
      [0] magic [1] version [2] kind [3] n_dof [4] n_frames_tf (0 or n_dof+1) [5] n_links
      [6] n_spheres [7] n_boxes [8] margin(f32) [9] off_tf [10] off_links [11] off_spheres
      [12] off_boxes [13] total_words [14] off_cull [15] off_frame_start
      [16] off_grid [17..19] grid dims nx,ny,nz [20..22] grid origin (f32) [23..25] 1/cell size (f32)
      [26] n_cells (0: no grid) [27] words to the next chained field (0: none) [28] field scale s_f (f32)
      (version 7: the grid section is a LIST grid -- build_list_grid -- and n_cand_words = total_words - off_grid - pad1024(n_cells) words of
      candidate bytes follow the padded cell words)
      [29] compile-time robot model id (model_gen.py; 0: none -- set only when the robot's tables equal the model's
      bit for bit) [30] keep mask over the MODEL's collision spheres (bit l: sphere l is in the link table) [31] reserved
      joint_tf    : n_frames_tf x 12   (row-major 3x4)
      links       : n_links x 8        (frame:int, ox, oy, oz, radius, 0, 0, 0)
      spheres     : n_spheres x 4      (cx, cy, cz, r)
      boxes       : n_boxes x 8        (cx, cy, cz, 0, hx, hy, hz, 0)
      cull        : ceil4(n_spheres) x 8 (-2cx, -2cy, -2cz, rhs, cx, cy, cz, r): a link sphere at x can
                    touch obstacle o only if |x|^2 - 2 x.c < rhs_o = (margin + max_l r_l + r_o)^2 - |c|^2 + slack
                    (3 fma + 1 compare per pair; the exact distance is evaluated only when a lane passes).
                    Padding entries never pass (rhs = -1e30).
      frame_start : n_frames + 1 ints (padded to 4): links [fs[j], fs[j+1]) ride on frame j+1
      grid        : nx*ny*nz uint32 words, x fastest (section zero-padded to a multiple of 1024 words).  Broad phase for the obstacle spheres: a word packs up
                    to four 8-bit obstacle indices (n_spheres = none: the far dummy) -- exactly the obstacles whose ball inflated
                    by (margin + max_l r_l + slack) touches the cell; GRID_OVERFLOW when more than four do.
                    A collision sphere at x can only be within its hinge threshold of the obstacles listed
                    in the cell containing x (none outside the grid), so per-LANE culling is exact.
    """
    if isinstance(field, (list, tuple)):
        fields = list(field)
        assert 1 <= len(fields) <= MAX_FIELDS, f'1..{MAX_FIELDS} collision fields per geometry buffer'
        scales = [1.0] * len(fields) if scales is None else [float(v) for v in scales]
        assert len(scales) == len(fields)
        # (the persistent kernel stages ONE grid format per launch: when any field needs the list grid, all of them take it)
        want_list = any(field_needs_list_grid(f) for f in fields) if list_grid is None else bool(list_grid)
        parts = [pack_geometry(robot, f, scales=[sc], prune_static=prune_static, use_model=use_model, list_grid=want_list)
                 for f, sc in zip(fields, scales)]
        for i, part in enumerate(parts[:-1]):
            part.view(np.int32)[27] = part.size
        return np.concatenate(parts)
    rs, fs = robot.spec(), field.spec()
    from . import model_gen
    model_id, keep_mask = 0, 0
    for name, mid in model_gen.MODEL_IDS.items():
        if use_model and model_gen.matches_model(rs, name):
            model_id, keep_mask = mid, (1 << len(rs['link_radius'])) - 1
    if model_id and hinge_bound(rs, fs) >= 1.0:
        # the model kernels take relu(hinge) as the [0, 1] clamp of the instruction encoding (csrc/mpb_geom.h, UNIT): a scene
        # whose hinges could reach 1 m keeps the table-driven walk
        model_id, keep_mask = 0, 0
    if prune_static and rs['kind'] == KIND_CHAIN:
        keep = links_that_can_touch(rs, fs)
        if model_id:
            keep_mask = int(sum(1 << l for l in range(len(keep)) if keep[l]))
            if np.any(np.asarray(rs['link_frame'])[~keep] != 1):      # the model kernels only expect frame-1 spheres to go
                model_id, keep_mask = 0, 0
        if not keep.all():
            rs = dict(rs, link_frame=np.asarray(rs['link_frame'])[keep], link_offset=np.asarray(rs['link_offset'])[keep],
                      link_radius=np.asarray(rs['link_radius'])[keep])
    n_tf = rs['joint_tf'].shape[0]
    n_links = len(rs['link_radius'])
    n_sph, n_box = len(fs['spheres']), len(fs['boxes'])
    n_sph_pad = (n_sph + 3) // 4 * 4
    n_frames = max(n_tf, 1)
    n_fs = (n_frames + 1 + 3) // 4 * 4
    off_tf = GEOM_HEADER_WORDS
    off_links = off_tf + 12 * n_tf
    off_sph = off_links + 8 * n_links
    off_box = off_sph + 4 * n_sph
    off_cull = off_box + 8 * n_box
    off_fs = off_cull + 8 * n_sph_pad
    off_grid = off_fs + n_fs
    planar = (rs['kind'] == KIND_POINT and rs['n_dof'] == 2 and (len(fs['spheres']) == 0 or bool(np.all(np.asarray(fs['spheres'])[:, 2] == 0.0))))
    a_max_g = float(fs['margin']) + float(np.max(rs['link_radius']))
    want_list = field_needs_list_grid(field) if list_grid is None else bool(list_grid)
    grid, version = None, GEOM_VERSION
    if want_list:
        grid = build_list_grid(np.asarray(fs['spheres']).reshape(-1, 4), np.asarray(fs['boxes']).reshape(-1, 6), a_max_g, planar=planar)
        if grid is not None:
            version = GEOM_VERSION_LIST
    if grid is None:
        grid = build_grid(fs['spheres'], a_max_g, planar=planar)
    n_cells = 0 if grid is None else int(grid['words'].size)
    n_cand_words = int(grid['cand'].size) // 4 if version == GEOM_VERSION_LIST else 0
    off_cand = off_grid + (n_cells + GRID_PAD - 1) // GRID_PAD * GRID_PAD   # the cell words are staged by the kernels in whole 16-byte rounds
    total = off_cand + n_cand_words
    buf = np.zeros((total,), dtype=np.float32)
    ibuf = buf.view(np.int32)
    ibuf[0:8] = [GEOM_MAGIC, version, rs['kind'], rs['n_dof'], n_tf, n_links, n_sph, n_box]
    buf[8] = fs['margin']
    ibuf[9:17] = [off_tf, off_links, off_sph, off_box, total, off_cull, off_fs, off_grid]
    if grid is not None:
        ibuf[17:20] = grid['dims']
        buf[20:23] = grid['lo']
        buf[23:26] = grid['inv']
        ibuf[26] = n_cells
        ibuf[31] = grid['k_lin']
        buf.view(np.uint32)[off_grid:off_grid + n_cells] = grid['words']
        if version == GEOM_VERSION_LIST:
            buf.view(np.uint8)[4 * off_cand:4 * off_cand + grid['cand'].size] = grid['cand']
    buf[off_tf:off_links] = rs['joint_tf'].astype(np.float32).reshape(-1)
    links = np.zeros((n_links, 8), np.float32)
    links.view(np.int32)[:, 0] = rs['link_frame'] if rs['kind'] == KIND_CHAIN else 1
    links[:, 1:4] = rs['link_offset']
    links[:, 4] = rs['link_radius']
    buf[off_links:off_sph] = links.reshape(-1)
    buf[off_sph:off_box] = fs['spheres'].reshape(-1)
    boxes = np.zeros((n_box, 8), np.float32)
    if n_box:
        boxes[:, 0:3] = fs['boxes'][:, 0:3]
        boxes[:, 4:7] = fs['boxes'][:, 3:6]
    buf[off_box:off_cull] = boxes.reshape(-1)
    # conservative cull table (fp64 on the host, rounded up)
    cull = np.zeros((n_sph_pad, 8), np.float32)
    cull[:, 3] = -1.0e30
    cull[:, 4:7] = -1.0e9   # (parked link slots sit at +1e9: never near a padding obstacle)
    if n_sph:
        c = fs['spheres'][:, :3].astype(np.float64)
        r = fs['spheres'][:, 3].astype(np.float64)
        a_max = float(fs['margin']) + float(np.max(rs['link_radius']))
        T = a_max + r
        cc = (c * c).sum(1)
        cmax = float(np.sqrt(cc.max()))
        slack = 2.0 ** -20 * (2.0 * cmax + float(T.max())) ** 2 + 1e-7   # 4x the fp32 error of the test value
        rhs = T * T - cc + slack
        cull[:n_sph, 0:3] = (-2.0 * c).astype(np.float32)
        cull[:n_sph, 3] = np.nextafter(rhs.astype(np.float32), np.float32(np.inf))
        cull[:n_sph, 4:7] = fs['spheres'][:, :3]
        cull[:n_sph, 7] = fs['spheres'][:, 3]
    buf[off_cull:off_fs] = cull.reshape(-1)
    fstart = np.zeros((n_fs,), np.int32)
    if rs['kind'] == KIND_CHAIN:
        lf = np.asarray(rs['link_frame'])
        for j in range(n_frames + 1):
            fstart[j] = int(np.searchsorted(lf, j + 1, side='left'))
        fstart[n_frames + 1:] = n_links
    else:
        fstart[0], fstart[1:] = 0, n_links
    ibuf[off_fs:off_grid] = fstart
    ibuf[27] = 0
    buf[28] = 1.0 if scales is None else float(scales[0])
    ibuf[29] = model_id
    buf.view(np.uint32)[30] = keep_mask
    return buf


def count_fields(packed):
    """Number of chained fields in a packed geometry buffer."""
    gi = np.asarray(packed).view(np.int32)
    n, off = 1, 0
    while gi[off + 27] != 0:
        off += int(gi[off + 27])
        n += 1
    return n


# ----------------------------------------------------------------------------------------------
# Synthetic environments for BASELINE.md configs (stand-ins for torch_robotics environments)
# ----------------------------------------------------------------------------------------------

def env_grid_circles_2d(margin=0.005, radius=0.05, n=7, extent=0.75):
    """C1: n x n grid of circles on [-extent, extent]^2 (stand-in for EnvGridCircles2D)."""
    xs = np.linspace(-extent, extent, n)
    c = np.array([(x, y, radius) for x in xs for y in xs], dtype=np.float32)
    return CollisionField(spheres=c, margin=margin)


def env_dense_2d(seed=3, n_circles=32, n_boxes=8, margin=0.01):
    """C2: random circles + boxes in [-1,1]^2, sizes U[0.05,0.15] (stand-in for EnvDense2D)."""
    rng = np.random.RandomState(seed)
    circles = np.concatenate([rng.uniform(-1, 1, (n_circles, 2)), rng.uniform(0.05, 0.15, (n_circles, 1))], 1)
    boxes = np.concatenate([rng.uniform(-1, 1, (n_boxes, 2)), rng.uniform(0.05, 0.15, (n_boxes, 2))], 1)
    return CollisionField(spheres=circles.astype(np.float32), boxes=boxes.astype(np.float32), margin=margin)


def env_spheres_3d(seed=0, n_spheres=16, margin=0.05):
    """C3/C4/C5: random spheres r U[0.05,0.15], centres U[-0.8,0.8]^3, none within 0.2 of the base axis."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n_spheres:
        c = rng.uniform(-0.8, 0.8, 3)
        r = rng.uniform(0.05, 0.15)
        if math.hypot(c[0], c[1]) - r < 0.2:
            continue
        out.append((c[0], c[1], c[2], r))
    return CollisionField(spheres=np.array(out, np.float32), margin=margin)


def env_spheres_boxes_3d(seed=0, n_spheres=200, n_boxes=32, margin=0.04):
    """A crowded 3-D scene (round 6): spheres r U[0.04, 0.12] and boxes with half extents U[0.03, 0.10], centres U[-0.8, 0.8]^3, none
    within 0.25 of the base axis -- beyond the compact broad-phase grid (63 spheres), so pack_geometry gives it the LIST grid of
    geometry version 7."""
    rng = np.random.RandomState(seed)
    sph, box = [], []
    while len(sph) < n_spheres:
        c, r = rng.uniform(-0.8, 0.8, 3), rng.uniform(0.04, 0.12)
        if math.hypot(c[0], c[1]) - r > 0.25:
            sph.append((*c, r))
    while len(box) < n_boxes:
        c, h = rng.uniform(-0.8, 0.8, 3), rng.uniform(0.03, 0.10, 3)
        if math.hypot(c[0], c[1]) - math.hypot(h[0], h[1]) > 0.25:
            box.append((*c, *h))
    return CollisionField(spheres=np.array(sph, np.float32), boxes=np.array(box, np.float32) if n_boxes else None, margin=margin)
