"""Tensor-level wrappers over the C-ABI: take PyTorch-ROCm tensors, pass raw device pointers + the
current HIP stream.  PyTorch is plumbing here (device memory, streams); all arithmetic is in
csrc/*.hip.  Every wrapper validates device / dtype / contiguity / shape on the host before
launching (a kernel that faults can reset the whole GPU host).
"""
import ctypes
import functools

import numpy as np
import torch

from . import _lib
from .geometry import count_fields, pack_geometry


# the raw handle of a device's current stream: torch.cuda.current_stream(dev).cuda_stream builds a Stream object per call (~2.5 us,
# a quarter of the host side of a persistent STOMP call); torch keeps the raw getter its own launchers use
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def raw_stream(device_index):
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    # called inside _on_tensor_device: the current device is the tensors' device
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_tensor_device(fn):
    """Every wrapper launches on the device its tensors live on, on THAT device's current stream: all GPU tensor
    arguments (a DeviceGeometry counts through its buffer) must share one device, which becomes the current device for
    the duration of the call.  Without this a planner built for cuda:1 while cuda:0 is current would launch on GPU 0
    with GPU-1 pointers."""
    @functools.wraps(fn)
    def run(*args, **kw):
        dev = None
        for a in list(args) + list(kw.values()):
            t = a.buf if isinstance(a, DeviceGeometry) else a
            if isinstance(t, torch.Tensor) and t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise ValueError(f'{fn.__name__}: tensor arguments live on different devices ({dev} and {t.device})')
        if dev is None:
            return fn(*args, **kw)       # no GPU tensor: the shape / device checks of the wrapper raise
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return run


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk(t, shape, name, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise ValueError(f'{name} is required')
    if not t.is_cuda:
        raise ValueError(f'{name} must live on the GPU (got {t.device}); there is no CPU path')
    if t.dtype != torch.float32:
        raise ValueError(f'{name} must be float32 (got {t.dtype})')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name} has shape {tuple(t.shape)}, expected {tuple(shape)}')


class DeviceGeometry:
    """Packed robot + field geometry resident in HBM (one small fp32 buffer)."""

    def __init__(self, robot, field, device, scales=None, keep_all_links=False, use_model=True):
        """`field`: one CollisionField or a list of up to 4 (evaluated as sum_f scales[f] * cost_f).
        keep_all_links: pack every collision sphere of the robot (needed by the per-sphere entry points
        fk_collision_points / field_cost_points); by default spheres that can never reach an obstacle of the field are
        left out of the link table (geometry.links_that_can_touch: exact, cost and gradient unchanged)."""
        self.robot, self.field = robot, field
        host = pack_geometry(robot, field, scales=scales, prune_static=not keep_all_links, use_model=use_model)
        self.all_links = int(host.view(np.int32)[5]) == len(robot.spec()['link_radius']) and (
            not isinstance(field, (list, tuple)) or keep_all_links or len(field) == 1)
        _lib.geom_check(host)
        self.flags = _lib.geom_flags(host)
        self.host = host
        self.n_dof = robot.q_dim
        self.n_fields = count_fields(host)
        self.buf = torch.from_numpy(host.copy()).to(device)

    @classmethod
    def from_packed(cls, packed, device):
        self = cls.__new__(cls)
        host = np.ascontiguousarray(packed, dtype=np.float32)
        _lib.geom_check(host)
        self.flags = _lib.geom_flags(host)
        self.robot = self.field = None
        self.all_links = True
        self.host = host
        self.n_dof = int(host.view(np.int32)[3])
        self.n_fields = count_fields(host)
        self.buf = torch.from_numpy(host.copy()).to(device)
        return self


@_on_tensor_device
def cost_collision_eval(trajs, geom, k_sigma, weight=1.0, h_begin=1, per_waypoint=False):
    B, H, d = trajs.shape
    _chk(trajs, (B, H, d), 'trajs')
    out = torch.empty(B, device=trajs.device, dtype=torch.float32)
    pw = torch.empty(B, H, device=trajs.device, dtype=torch.float32) if per_waypoint else None
    _lib.check(_lib.lib().mpb_cost_collision_eval(_ptr(trajs), _ptr(geom.buf), _ptr(out), _ptr(pw), B, H, d, h_begin,
                                                 float(k_sigma), float(weight), _stream()), 'mpb_cost_collision_eval')
    return (out, pw) if per_waypoint else out


@_on_tensor_device
def cost_collision_grad(trajs, geom, k_sigma, weight=1.0, h_begin=1, grad=None):
    B, H, d = trajs.shape
    _chk(trajs, (B, H, d), 'trajs')
    out = torch.empty(B, device=trajs.device, dtype=torch.float32)
    if grad is None:
        grad = torch.empty_like(trajs)
    else:
        _chk(grad, (B, H, d), 'grad')
    _lib.check(_lib.lib().mpb_cost_collision_grad(_ptr(trajs), _ptr(geom.buf), int(geom.flags), _ptr(out), _ptr(grad), B, H, d, h_begin,
                                                 float(k_sigma), float(weight), _stream()), 'mpb_cost_collision_grad')
    return out, grad


TERM_GP, TERM_START, TERM_GOAL, TERM_SMOOTH, TERM_JLIM, TERM_VEL_FD = 1, 2, 4, 8, 16, 32   # include/mpb.h MPB_TERM_*


@_on_tensor_device
def cost_terms_eval(trajs, n_dof, dt=0.0, k_gp=0.0, vel_fd=False, k_start=0.0, start_state=None, k_goal=0.0,
                    goal_states=None, trajs_per_goal=1, k_smooth=0.0, k_jlim=0.0, q_min=None, q_max=None, jl_eps=0.0,
                    out=None, accumulate=False, broadcast_jlim=True, terms=None):
    """One pass over trajs (B,H,d) evaluating the enabled trajectory-only cost terms (mpb_cost_terms_eval).
    A term is enabled by naming it in `terms` (iterable of 'gp','start','goal','smooth','jlim').
    Returns (out (B,), jl_total 0-dim fp64 tensor or None)."""
    B, H, d = trajs.shape
    _chk(trajs, (B, H, d), 'trajs')
    terms = set(terms or ())
    unknown = terms - {'gp', 'start', 'goal', 'smooth', 'jlim'}
    if unknown:
        raise ValueError(f'unknown cost terms {sorted(unknown)}')
    flags = 0
    if 'gp' in terms:
        flags |= TERM_GP | (TERM_VEL_FD if vel_fd else 0)
    if 'start' in terms:
        flags |= TERM_START
        _chk(start_state, (2 * n_dof,), 'start_state')
    if 'goal' in terms:
        flags |= TERM_GOAL
        if goal_states is None or goal_states.ndim != 2:
            raise ValueError('goal_states (G, 2*n_dof) is required')
        _chk(goal_states, (goal_states.shape[0], 2 * n_dof), 'goal_states')
        if trajs_per_goal < 1 or goal_states.shape[0] * trajs_per_goal < B:
            raise ValueError(f'{goal_states.shape[0]} goals x {trajs_per_goal} trajectories do not cover B={B}')
    if 'smooth' in terms:
        flags |= TERM_SMOOTH
    jl_total = None
    if 'jlim' in terms:
        flags |= TERM_JLIM
        _chk(q_min, (n_dof,), 'q_min')
        _chk(q_max, (n_dof,), 'q_max')
        jl_total = torch.zeros((), device=trajs.device, dtype=torch.float64)
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs an existing out buffer')
        out = torch.empty(B, device=trajs.device, dtype=torch.float32)
    else:
        if out.numel() != B:
            raise ValueError(f'out has {out.numel()} elements, expected {B}')
        _chk(out, out.shape, 'out')
    _lib.check(_lib.lib().mpb_cost_terms_eval(
        _ptr(trajs), _ptr(out), _ptr(jl_total), _ptr(start_state), _ptr(goal_states), _ptr(q_min), _ptr(q_max),
        B, H, d, int(n_dof), int(trajs_per_goal), flags, float(dt), float(k_gp), float(k_start), float(k_goal),
        float(k_smooth), float(k_jlim), float(jl_eps), int(bool(accumulate)), int(bool(broadcast_jlim)), _stream()),
        'mpb_cost_terms_eval')
    return out, jl_total


def _need_all_links(geom):
    if not getattr(geom, 'all_links', True):
        raise ValueError('this geometry leaves out collision spheres that cannot reach an obstacle; the per-sphere entry '
                         'points need DeviceGeometry(..., keep_all_links=True)')


@_on_tensor_device
def fk_collision_points(q, geom):
    """q (B,H,d) -> positions of the robot's collision spheres (B,H,L,3) (mpb_fk_collision_points)."""
    B, H, d = q.shape
    _chk(q, (B, H, d), 'q')
    _need_all_links(geom)
    L = int(geom.host.view(np.int32)[5])
    pts = torch.empty(B, H, L, 3, device=q.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_fk_collision_points(_ptr(q), _ptr(geom.buf), _ptr(pts), B, H, d, _stream()), 'mpb_fk_collision_points')
    return pts


@_on_tensor_device
def fk_collision_points_vjp(q, geom, grad_pts):
    B, H, d = q.shape
    _need_all_links(geom)
    L = int(geom.host.view(np.int32)[5])
    _chk(q, (B, H, d), 'q')
    _chk(grad_pts, (B, H, L, 3), 'grad_pts')
    gq = torch.empty(B, H, geom.n_dof, device=q.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_fk_collision_points_vjp(_ptr(q), _ptr(geom.buf), _ptr(grad_pts), _ptr(gq), B, H, d, _stream()),
               'mpb_fk_collision_points_vjp')
    return gq


@_on_tensor_device
def field_cost_points(pts, geom):
    """Collision-sphere positions (B,H,L,3) -> hinge cost per waypoint (B,H) (mpb_field_cost_points)."""
    B, H, L, _ = pts.shape
    _need_all_links(geom)
    _chk(pts, (B, H, int(geom.host.view(np.int32)[5]), 3), 'pts')
    cost = torch.empty(B, H, device=pts.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_field_cost_points(_ptr(pts), _ptr(geom.buf), _ptr(cost), B, H, _stream()), 'mpb_field_cost_points')
    return cost


@_on_tensor_device
def field_cost_points_vjp(pts, geom, grad_cost):
    B, H, L, _ = pts.shape
    _need_all_links(geom)
    _chk(pts, (B, H, int(geom.host.view(np.int32)[5]), 3), 'pts')
    _chk(grad_cost, (B, H), 'grad_cost')
    gp = torch.empty_like(pts)
    _lib.check(_lib.lib().mpb_field_cost_points_vjp(_ptr(pts), _ptr(geom.buf), _ptr(grad_cost), _ptr(gp), B, H, _stream()),
               'mpb_field_cost_points_vjp')
    return gp


@_on_tensor_device
def gp_factor_error(x, D, dt):
    """(B,H,2D) -> (B,H-1,2D): x_{t+1} - Phi x_t (mpb_gp_factor_error)."""
    B, H, dim = x.shape
    _chk(x, (B, H, 2 * D), 'x')
    out = torch.empty(B, H - 1, dim, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_gp_factor_error(_ptr(x), _ptr(out), B, H, D, float(dt), _stream()), 'mpb_gp_factor_error')
    return out


@_on_tensor_device
def cost_terms_grad(trajs, n_dof, grad_in=None, grad_out=None, apply=False, R=None, prior_bw=0.0, lr=0.0, grad_clip=0.0,
                    jl_scale=1.0, dt=0.0, k_gp=0.0, vel_fd=False, k_start=0.0, start_state=None, k_goal=0.0,
                    goal_states=None, trajs_per_goal=1, k_smooth=0.0, k_jlim=0.0, q_min=None, q_max=None, jl_eps=0.0,
                    terms=None):
    """Analytic gradient of the trajectory-only cost terms (+ CHOMP's smoothness prior, + an incoming gradient such as
    the collision one), written to grad_out or -- apply=True -- consumed by CHOMP's clamped, end-masked step on
    `trajs` in place (mpb_cost_terms_grad).  Term arguments as cost_terms_eval."""
    B, H, d = trajs.shape
    _chk(trajs, (B, H, d), 'trajs')
    terms = set(terms or ())
    unknown = terms - {'gp', 'start', 'goal', 'smooth', 'jlim'}
    if unknown:
        raise ValueError(f'unknown cost terms {sorted(unknown)}')
    flags = 0
    if 'gp' in terms:
        flags |= TERM_GP | (TERM_VEL_FD if vel_fd else 0)
    if 'start' in terms:
        flags |= TERM_START
        _chk(start_state, (2 * n_dof,), 'start_state')
    if 'goal' in terms:
        flags |= TERM_GOAL
        _chk(goal_states, (goal_states.shape[0], 2 * n_dof), 'goal_states')
    if 'smooth' in terms:
        flags |= TERM_SMOOTH
    if 'jlim' in terms:
        flags |= TERM_JLIM
        _chk(q_min, (n_dof,), 'q_min')
        _chk(q_max, (n_dof,), 'q_max')
    if grad_in is not None:
        _chk(grad_in, (B, H, d), 'grad_in')
    if not apply:
        if grad_out is None:
            grad_out = torch.empty(B, H, d, device=trajs.device, dtype=torch.float32)
        _chk(grad_out, (B, H, d), 'grad_out')
    if prior_bw != 0.0:
        _chk(R, (H, H), 'R')
    _lib.check(_lib.lib().mpb_cost_terms_grad(
        _ptr(trajs), _ptr(grad_in), _ptr(grad_out), _ptr(R), _ptr(start_state), _ptr(goal_states), _ptr(q_min), _ptr(q_max),
        B, H, d, int(n_dof), int(trajs_per_goal), flags, float(dt), float(k_gp), float(k_start), float(k_goal),
        float(k_smooth), float(k_jlim), float(jl_eps), float(jl_scale), float(prior_bw), float(lr), float(grad_clip),
        int(bool(apply)), _stream()), 'mpb_cost_terms_grad')
    return grad_out


@_on_tensor_device
def traj_interpolate(trajs, n_interp):
    """(B,H,d) -> (B,(H-1)(n+1)+1,d): n evenly spaced joint-space points per segment (mpb_traj_interpolate)."""
    B, H, d = trajs.shape
    _chk(trajs, (B, H, d), 'trajs')
    n = int(n_interp)
    out = torch.empty(B, (H - 1) * (n + 1) + 1, d, device=trajs.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_traj_interpolate(_ptr(trajs), _ptr(out), B, H, d, n, _stream()), 'mpb_traj_interpolate')
    return out


@_on_tensor_device
def traj_resample(paths, lengths, H, dt):
    """N padded polylines (N,Lmax,D) with `lengths` (N,) int32 valid rows -> (N,H,2D) support points uniform in
    arc length + average-velocity channel (mpb_traj_resample)."""
    N, Lmax, D = paths.shape
    _chk(paths, (N, Lmax, D), 'paths')
    if not (lengths.is_cuda and lengths.dtype == torch.int32 and lengths.is_contiguous() and tuple(lengths.shape) == (N,)):
        raise ValueError('lengths must be a contiguous int32 GPU tensor of shape (N,)')
    out = torch.empty(N, H, 2 * D, device=paths.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_traj_resample(_ptr(paths), _ptr(lengths), _ptr(out), N, Lmax, int(H), D, float(dt), _stream()),
               'mpb_traj_resample')
    return out


@_on_tensor_device
def traj_finite_difference(pos, dt):
    """(B,H,D) positions -> (B,H,2D) [pos, central-difference velocities] (mpb_traj_finite_difference)."""
    B, H, D = pos.shape
    _chk(pos, (B, H, D), 'pos')
    out = torch.empty(B, H, 2 * D, device=pos.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_traj_finite_difference(_ptr(pos), _ptr(out), B, H, D, float(dt), _stream()),
               'mpb_traj_finite_difference')
    return out


@_on_tensor_device
def stomp_step(means, eps, samples, costs, weights, L, Sigma, geom, S, D, k_sigma, weight, lr, temperature,
               n_iters=1, seed=0, iter0=0, particle_offset=0):
    P, H, d = means.shape
    _chk(means, (P, H, d), 'means')
    _chk(samples, (P, S, H, d), 'samples')
    _chk(costs, (P, S), 'costs')
    _chk(weights, (P, S), 'weights')
    _chk(L, (H, H), 'L')
    _chk(Sigma, (H, H), 'Sigma')
    if eps is not None:
        _chk(eps, (n_iters, S, d, P, H), 'eps')
    _lib.check(_lib.lib().mpb_stomp_step(
        _ptr(means), _ptr(eps), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(L), _ptr(Sigma), _ptr(geom.buf),
        int(geom.flags), P, S, H, d, D, float(k_sigma), float(weight), float(lr), float(temperature), int(n_iters),
        int(seed) & (2 ** 64 - 1), int(iter0), int(particle_offset), _stream()), 'mpb_stomp_step')


def stomp_workspace(P, S, H, d, device):
    """Exchange buffer of the persistent STOMP kernel (mpb_stomp_run): header zeroed (mpb_stomp_workspace_init), the
    rest need not be initialised.  One workspace serves one call at a time."""
    n = int(_lib.lib().mpb_stomp_workspace_bytes(int(P), int(S), int(H), int(d)))
    ws = torch.empty((n + 3) // 4, device=device, dtype=torch.float32)
    with torch.cuda.device(ws.device):
        _lib.check(_lib.lib().mpb_stomp_workspace_init(_ptr(ws), ws.numel() * 4, _stream()), 'mpb_stomp_workspace_init')
    return ws


STOMP_PATH_TWO_KERNEL, STOMP_PATH_PERSISTENT_EXCHANGE, STOMP_PATH_PERSISTENT = 0, 1, 2


def stomp_run_path(geom, workspace, P, S, H, d):
    """Which form of the loop mpb_stomp_run takes for this call (STOMP_PATH_*), without launching anything."""
    nbytes = 0 if workspace is None else workspace.numel() * 4
    with torch.cuda.device(geom.buf.device):
        return int(_lib.lib().mpb_stomp_run_path(int(geom.flags), nbytes, int(P), int(S), int(H), int(d)))


class StompRunStatus:
    """Host-visible status block of mpb_stomp_run_checked: 8 words of pinned host memory the kernel writes directly
    ([0] tag of the last completed call, [1] tag of the last LOST call, [2] why, [4..7] device real-time stamps of the
    launch's begin / end), read here without synchronising."""

    def __init__(self):
        self.buf = torch.zeros(8, dtype=torch.int32).pin_memory()
        self._view = self.buf.numpy().view(np.uint32)
        self.issued = []             # tags of the persistent launches not yet known to be complete, in launch order
        self.tag_c = ctypes.c_uint32(0)

    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr())

    def note_launch(self):
        tag = int(self.tag_c.value)
        if tag:
            self.issued.append(tag)
        return tag

    def lost(self):
        """(tag, why) of a lost call among the ones issued through this block, else None.  Reads host memory only."""
        lost, done = int(self._view[1]), int(self._view[0])
        if lost and lost in self.issued:
            return lost, int(self._view[2])
        if done in self.issued:      # calls complete in launch order: everything up to `done` is over, and was fine
            del self.issued[:self.issued.index(done) + 1]
        return None

    def device_span_ms(self):
        """Duration of the last COMPLETED persistent launch as the device saw it (100 MHz real-time counter: first unit
        started -> last workgroup left), in milliseconds; None if no launch has completed.  Host memory only."""
        v = self._view
        if int(v[0]) == 0:
            return None
        t0, t1 = int(v[4]) | (int(v[5]) << 32), int(v[6]) | (int(v[7]) << 32)
        return (t1 - t0) * 1e-5 if t1 > t0 else None

    def acknowledge(self, tag):
        """Forget a lost call (after it has been reported)."""
        if tag in self.issued:
            del self.issued[:self.issued.index(tag) + 1]


@_on_tensor_device
def stomp_run(means, eps, samples, costs, weights, L, Sigma, geom, S, D, k_sigma, weight, lr, temperature, workspace,
              n_iters=1, seed=0, iter0=0, particle_offset=0, status=None, means_copy=None):
    """stomp_step as ONE persistent launch where the shape allows it (H = 64, S <= 64, grid-backed fields), the
    two-kernel loop otherwise (the C side decides).  status: a StompRunStatus the kernel reports a lost call to
    (include/mpb.h, "Failure contract"); means_copy: a second (P,H,d) destination of the final means, written by the same
    launch.  Returns the call's tag (0: two-kernel loop)."""
    P, H, d = means.shape
    _chk(means, (P, H, d), 'means')
    _chk(samples, (P, S, H, d), 'samples')
    _chk(costs, (P, S), 'costs')
    _chk(weights, (P, S), 'weights')
    _chk(L, (H, H), 'L')
    _chk(Sigma, (H, H), 'Sigma')
    if eps is not None:
        _chk(eps, (n_iters, S, d, P, H), 'eps')
    if workspace is not None:
        _chk(workspace, tuple(workspace.shape), 'workspace')
    if means_copy is not None:
        _chk(means_copy, (P, H, d), 'means_copy')
    _lib.check(_lib.lib().mpb_stomp_run_checked(
        _ptr(means), _ptr(eps), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(L), _ptr(Sigma), _ptr(geom.buf),
        int(geom.flags), _ptr(workspace), 0 if workspace is None else workspace.numel() * 4,
        P, S, H, d, D, float(k_sigma), float(weight), float(lr), float(temperature), int(n_iters),
        int(seed) & (2 ** 64 - 1), int(iter0), int(particle_offset),
        None if status is None else status.ptr(), None if status is None else ctypes.byref(status.tag_c), _ptr(means_copy),
        _stream()), 'mpb_stomp_run')
    return 0 if status is None else status.note_launch()


class StompRunPlan:
    """The arguments of stomp_run for a planner whose buffers do not change between optimize() calls, validated ONCE and
    kept as ctypes values: launch() converts nothing but the iteration count and the stream.  (stomp_run's per-call
    checks and conversions are ~15 us of host time -- 4 % of a 20-iteration call at C3.)  Device-noise calls only."""

    def __init__(self, means, samples, costs, weights, L, Sigma, geom, S, D, k_sigma, weight, lr, temperature, workspace,
                 seed, particle_offset, status):
        P, H, d = means.shape
        _chk(means, (P, H, d), 'means')
        _chk(samples, (P, S, H, d), 'samples')
        _chk(costs, (P, S), 'costs')
        _chk(weights, (P, S), 'weights')
        _chk(L, (H, H), 'L')
        _chk(Sigma, (H, H), 'Sigma')
        _chk(workspace, tuple(workspace.shape), 'workspace')
        devs = {t.device for t in (means, samples, costs, weights, L, Sigma, geom.buf, workspace)}
        if len(devs) != 1:
            raise ValueError(f'StompRunPlan: tensors live on different devices ({devs})')
        self.device = means.device
        self.shape = (P, H, d)
        self.key = (means.data_ptr(), samples.data_ptr(), costs.data_ptr(), weights.data_ptr(), L.data_ptr(),
                    Sigma.data_ptr(), geom.buf.data_ptr(), workspace.data_ptr(), S, D, float(k_sigma), float(weight),
                    float(lr), float(temperature), int(seed), int(particle_offset))
        self._keep = (means, samples, costs, weights, L, Sigma, geom, workspace, status)     # the pointers below stay valid
        c = ctypes
        self._head = (_ptr(means), c.c_void_p(0), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(L), _ptr(Sigma),
                      _ptr(geom.buf), c.c_int(int(geom.flags)), _ptr(workspace), c.c_size_t(workspace.numel() * 4),
                      c.c_int(P), c.c_int(S), c.c_int(H), c.c_int(d), c.c_int(D), c.c_float(k_sigma), c.c_float(weight),
                      c.c_float(lr), c.c_float(temperature))
        self._seed = c.c_uint64(int(seed) & (2 ** 64 - 1))
        self._poff = c.c_uint32(int(particle_offset))
        self._status = status
        self._status_ptr = status.ptr()
        self._tag_ref = c.byref(status.tag_c)
        # the same arguments kept on the library's side (mpb_stomp_plan_*): launch() hands over four values per call
        h = c.c_void_p(0)
        _lib.check(_lib.lib().mpb_stomp_plan_create(
            c.byref(h), _ptr(means), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(L), _ptr(Sigma), _ptr(geom.buf), int(geom.flags),
            _ptr(workspace), workspace.numel() * 4, P, S, H, d, D, float(k_sigma), float(weight), float(lr), float(temperature),
            int(seed) & (2 ** 64 - 1), int(particle_offset), self._status_ptr), 'mpb_stomp_plan_create')
        self._handle = h
        self._launch = _lib.lib().mpb_stomp_plan_launch
        self._destroy = _lib.lib().mpb_stomp_plan_destroy

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h is not None and h.value:
            self._destroy(h)
            self._handle = None

    def launch_timed(self, n_iters, iter0, means_copy=None):
        """launch() with the kernel's own duration measured on the dispatch (mpb_stomp_run_timed): synchronises; returns ms."""
        ms = ctypes.c_float(0.0)
        rc = _lib.lib().mpb_stomp_run_timed(*self._head, int(n_iters), self._seed, int(iter0), self._poff, self._status_ptr,
                                            self._tag_ref, None if means_copy is None else means_copy.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream, ctypes.byref(ms))
        if rc != 0:
            _lib.check(rc, 'mpb_stomp_run_timed')
        self._status.note_launch()
        return float(ms.value)

    def launch(self, n_iters, iter0, means_copy=None):
        """Enqueue the call on the current stream of the plan's device (which must be the current device)."""
        rc = self._launch(self._handle, n_iters, iter0, None if means_copy is None else means_copy.data_ptr(),
                          raw_stream(self.device.index), self._tag_ref)
        if rc != 0:
            _lib.check(rc, 'mpb_stomp_run')
        return self._status.note_launch()


def stomp_run_state(workspace):
    """State of the last persistent stomp_run on this workspace (synchronises): 0 fine, 1 a workgroup gave up waiting
    for its partner (the call is lost), 2 the workspace header was not zeroed."""
    out = ctypes.c_int(0)
    with torch.cuda.device(workspace.device):
        _lib.check(_lib.lib().mpb_stomp_run_status(_ptr(workspace), _stream(), ctypes.cast(ctypes.pointer(out), ctypes.c_void_p)),
                   'mpb_stomp_run_status')
    return int(out.value)


def stomp_run_timed_out(workspace):
    """Was the last persistent stomp_run on this workspace lost?  (synchronises)"""
    return stomp_run_state(workspace) != 0


def debug_philox(ctr, key, rounds):
    """Test aid: raw Philox4x32-`rounds` words for (n,4) int64/uint32 counters and (n,2) keys (host arrays) -> (n,4) uint32."""
    ctr = np.ascontiguousarray(np.asarray(ctr, dtype=np.uint32).reshape(-1, 4))
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(-1, 2))
    n = ctr.shape[0]
    dev = torch.device('cuda', torch.cuda.current_device())
    c = torch.from_numpy(ctr.view(np.int32)).to(dev)
    k = torch.from_numpy(key.view(np.int32)).to(dev)
    out = torch.empty(n, 4, dtype=torch.int32, device=dev)
    _lib.debug_check(_lib.debug_lib().mpb_debug_philox(_ptr(c), _ptr(k), _ptr(out), n, int(rounds), _stream()), 'mpb_debug_philox')
    return out.cpu().numpy().view(np.uint32)


def debug_stomp_normals(P, S, d, n_iters, device, seed=0, iter0=0, particle_offset=0, H=64):
    """Test aid: the standard normals the STOMP kernels draw in throughput mode, (n_iters, P, S, d, 64 * ceil(H / 64)) fp32
    (columns k >= H are drawn by the kernels as well and meet zero columns of L)."""
    Hp = 64 * ((int(H) + 63) // 64)
    out = torch.empty(n_iters, P, S, d, Hp, device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.debug_check(_lib.debug_lib().mpb_debug_stomp_normals_h(_ptr(out), int(P), int(S), int(d), int(H), int(n_iters),
                                                                    int(seed) & (2 ** 64 - 1), int(iter0), int(particle_offset), _stream()),
                         'mpb_debug_stomp_normals_h')
    return out


def debug_mppi_normals(NP, S, T, c, n_iters, device, seed=0, iter0=0):
    """Test aid: the standard normals mppi_step draws in throughput mode, in the layout of its injected eps (n_iters, NP, c, S, T)."""
    out = torch.empty(n_iters, NP, c, S, T, device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.debug_check(_lib.debug_lib().mpb_debug_mppi_normals(_ptr(out), int(NP), int(S), int(T), int(c), int(n_iters),
                                                                 int(seed) & (2 ** 64 - 1), int(iter0), _stream()), 'mpb_debug_mppi_normals')
    return out


def debug_occupy(n_blocks, usec, device):
    """Test aid: n_blocks workgroups that each take a CU's LDS and idle for `usec` microseconds on the current stream."""
    sink = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(sink.device):
        _lib.debug_check(_lib.debug_lib().mpb_debug_occupy(int(n_blocks), int(usec), _ptr(sink), _stream()), 'mpb_debug_occupy')
    return sink


@_on_tensor_device
def stomp_step_profile(means, samples, costs, weights, L, Sigma, geom, S, D, k_sigma, weight, lr, temperature,
                       n_iters=50, seed=0, iter0=0, particle_offset=0):
    """Measurement aid: n_iters iterations of stomp_step (device noise) with per-dispatch HIP events; returns the average
    duration in ms of (sample+cost kernel, update kernel) inside that loop.  Synchronises the stream."""
    P, H, d = means.shape
    _chk(means, (P, H, d), 'means')
    _chk(samples, (P, S, H, d), 'samples')
    _chk(costs, (P, S), 'costs')
    _chk(weights, (P, S), 'weights')
    _chk(L, (H, H), 'L')
    _chk(Sigma, (H, H), 'Sigma')
    ka, kb = ctypes.c_float(0.0), ctypes.c_float(0.0)
    _lib.check(_lib.lib().mpb_stomp_step_profile(
        _ptr(means), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(L), _ptr(Sigma), _ptr(geom.buf), int(geom.flags),
        P, S, H, d, D, float(k_sigma), float(weight), float(lr), float(temperature), int(n_iters),
        int(seed) & (2 ** 64 - 1), int(iter0), int(particle_offset), _stream(),
        ctypes.cast(ctypes.pointer(ka), ctypes.c_void_p), ctypes.cast(ctypes.pointer(kb), ctypes.c_void_p)),
        'mpb_stomp_step_profile')
    return float(ka.value), float(kb.value)


@_on_tensor_device
def stomp_sample(means, eps, samples, L, S, seed=0, it=0, particle_offset=0, geom=None, costs=None, k_sigma=0.0,
                 weight=1.0):
    """First kernel of an iteration: draw + write samples; with geom/costs also the fused collision cost."""
    P, H, d = means.shape
    _chk(means, (P, H, d), 'means')
    _chk(samples, (P, S, H, d), 'samples')
    _chk(L, (H, H), 'L')
    if eps is not None:
        _chk(eps, (S, d, P, H), 'eps')
    if (geom is None) != (costs is None):
        raise ValueError('geom and costs must be given together')
    if costs is not None:
        _chk(costs, (P, S), 'costs')
    _lib.check(_lib.lib().mpb_stomp_sample(_ptr(means), _ptr(eps), _ptr(samples), _ptr(L),
                                          _ptr(None if geom is None else geom.buf),
                                          0 if geom is None else int(geom.flags), _ptr(costs), P, S, H, d,
                                          float(k_sigma), float(weight), int(seed) & (2 ** 64 - 1), int(it),
                                          int(particle_offset), _stream()), 'mpb_stomp_sample')


@_on_tensor_device
def stomp_update(means, samples, costs, weights, Sigma, lr, temperature):
    P, S, H, d = samples.shape
    _chk(means, (P, H, d), 'means')
    _chk(samples, (P, S, H, d), 'samples')
    _chk(costs, (P, S), 'costs')
    _chk(weights, (P, S), 'weights')
    _chk(Sigma, (H, H), 'Sigma', allow_none=True)   # None: update without the covariance product (StochGPMP)
    _lib.check(_lib.lib().mpb_stomp_update(_ptr(means), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(Sigma),
                                          P, S, H, d, float(lr), float(temperature), _stream()), 'mpb_stomp_update')


@_on_tensor_device
def chomp_step(means, R, geom, D, k_sigma, weight, w_prior, lr, grad_clip, n_iters=1, B_global=None, costs_out=None):
    B, H, d = means.shape
    _chk(means, (B, H, d), 'means')
    _chk(R, (H, H), 'R')
    _chk(costs_out, (B,), 'costs_out', allow_none=True)
    _lib.check(_lib.lib().mpb_chomp_step(_ptr(means), _ptr(R), _ptr(geom.buf), int(geom.flags), _ptr(costs_out), B,
                                        B if B_global is None else int(B_global), H, d, D, float(k_sigma), float(weight),
                                        float(w_prior), float(lr), float(grad_clip), int(n_iters), _stream()),
               'mpb_chomp_step')


@_on_tensor_device
def gpmp2_workspace(B, H, D, device):
    n = int(_lib.lib().mpb_gpmp2_workspace_bytes(B, H, D))
    if n == 0:
        raise ValueError(f'unsupported GPMP2 shape B={B} H={H} D={D}')
    return torch.empty(n, dtype=torch.uint8, device=device)


@_on_tensor_device
def gpmp2_step(x, start, goal, geom, workspace, sigmas, dt, delta, trust_region, step_size, n_iters=1, costs_out=None,
               n_interp=0):
    """n_iters Gauss-Newton iterations on one GPU.  sigmas = (start, gp, goal, coll)."""
    B, H, dim = x.shape
    D = dim // 2
    _chk(x, (B, H, dim), 'x')
    _chk(start, (B, dim), 'start')
    _chk(goal, (B, dim), 'goal')
    _chk(costs_out, (B,), 'costs_out', allow_none=True)
    assert workspace.numel() >= _lib.lib().mpb_gpmp2_workspace_bytes(B, H, D)
    _lib.check(_lib.lib().mpb_gpmp2_step(
        _ptr(x), _ptr(start), _ptr(goal), _ptr(geom.buf), int(geom.flags), _ptr(workspace), _ptr(costs_out), B, H, D, float(dt),
        float(sigmas[0]), float(sigmas[1]), float(sigmas[2]), float(sigmas[3]), float(delta), int(bool(trust_region)),
        float(step_size), int(n_iters), int(n_interp or 0), int(geom.n_fields), _stream()), 'mpb_gpmp2_step')


@_on_tensor_device
def gpmp2_linearize(x, geom, workspace, n_interp=0):
    B, H, dim = x.shape
    _chk(x, (B, H, dim), 'x')
    _lib.check(_lib.lib().mpb_gpmp2_linearize(_ptr(x), _ptr(geom.buf), int(geom.flags), _ptr(workspace), B, H, dim // 2,
                                              int(n_interp or 0), _stream()), 'mpb_gpmp2_linearize')


@_on_tensor_device
def gpmp2_collision_rows(x, geom, n_interp=0):
    """The collision factor's rows as the GPMP2 solve consumes them: (F, B, H, D+1) fp32 with [..., :D] = h_t =
    -d c_t / d q_t (with n_interp > 0: of the INTERPOLATED trajectory's summed cost, cost_functions.py:115-119,
    field_factor.py:42-54) and [..., D] = c_t, one set per chained field, each scaled by sqrt(s_f); row 0 is zero
    (traj_range [1, None]).  Runs mpb_gpmp2_linearize into a scratch buffer that holds only the Jacobian section of
    the GPMP2 workspace (the kernel writes nothing else)."""
    B, H, dim = x.shape
    D = dim // 2
    _chk(x, (B, H, dim), 'x')
    if _lib.lib().mpb_gpmp2_workspace_bytes(B, H, D) == 0:
        raise ValueError(f'unsupported GPMP2 shape B={B} H={H} D={D}')
    MAX_FIELDS = 4                                   # MPB_MAX_FIELDS: the section is laid out for four fields
    jac = torch.zeros(MAX_FIELDS, B, H, D + 1, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_gpmp2_linearize(_ptr(x), _ptr(geom.buf), int(geom.flags), _ptr(jac), B, H, D, int(n_interp or 0), _stream()),
               'mpb_gpmp2_linearize')
    return jac[:geom.n_fields]


@_on_tensor_device
def gpmp2_diag(workspace, B, H, D, sigmas, dt, n_fields=1):
    """Local SUM over particles of diag(A^T K A) as an (H*2D,) fp64 tensor."""
    out = torch.empty(H * 2 * D, dtype=torch.float64, device=workspace.device)
    _lib.check(_lib.lib().mpb_gpmp2_diag(_ptr(workspace), _ptr(out), B, H, D, int(n_fields), float(dt), float(sigmas[0]),
                                        float(sigmas[1]), float(sigmas[2]), float(sigmas[3]), _stream()), 'mpb_gpmp2_diag')
    return out


@_on_tensor_device
def gpmp2_solve(x, start, goal, diag_mean, workspace, sigmas, dt, delta, trust_region, step_size, costs_out=None,
                n_fields=1):
    B, H, dim = x.shape
    _chk(x, (B, H, dim), 'x')
    _chk(start, (B, dim), 'start')
    _chk(goal, (B, dim), 'goal')
    if diag_mean is not None:
        assert diag_mean.dtype == torch.float64 and diag_mean.is_cuda and diag_mean.numel() == H * dim
    _lib.check(_lib.lib().mpb_gpmp2_solve(
        _ptr(x), _ptr(start), _ptr(goal), _ptr(diag_mean), _ptr(workspace), _ptr(costs_out), B, H, dim // 2, int(n_fields),
        float(dt), float(sigmas[0]), float(sigmas[1]), float(sigmas[2]), float(sigmas[3]), float(delta), int(bool(trust_region)),
        float(step_size), _stream()), 'mpb_gpmp2_solve')


@_on_tensor_device
def mppi_step(mean, eps, scale_tril, cov_inv, state0, goal, ctrl_min, ctrl_max, discount, c_weights, geom, controls,
              states, costs, weights, dt, k_sigma=0.0, weight=1.0, temp=1.0, step_size=1.0, n_iters=1, seed=0, iter0=0,
              best_cost=None, best_states=None):
    NP, T, c = mean.shape
    S = controls.shape[1]
    _chk(mean, (NP, T, c), 'mean')
    if eps is not None:
        _chk(eps, (n_iters, NP, c, S, T), 'eps')
    _chk(scale_tril, (c, T, T), 'scale_tril')
    _chk(cov_inv, (c, T, T), 'cov_inv')
    _chk(state0, (NP, c), 'state0')
    _chk(goal, (NP, c), 'goal')
    _chk(ctrl_min, (c,), 'ctrl_min')
    _chk(ctrl_max, (c,), 'ctrl_max')
    _chk(discount, (T,), 'discount')
    _chk(c_weights, (4,), 'c_weights')
    _chk(controls, (NP, S, T, c), 'controls')
    _chk(states, (NP, S, T, c), 'states')
    _chk(costs, (NP, S), 'costs')
    _chk(weights, (NP, S), 'weights')
    if (best_cost is None) != (best_states is None):
        raise ValueError('best_cost and best_states must be given together')
    _chk(best_cost, (NP,), 'best_cost', allow_none=True)
    _chk(best_states, (NP, T, c), 'best_states', allow_none=True)
    _lib.check(_lib.lib().mpb_mppi_step(
        _ptr(mean), _ptr(eps), _ptr(scale_tril), _ptr(cov_inv), _ptr(state0), _ptr(goal), _ptr(ctrl_min), _ptr(ctrl_max),
        _ptr(discount), _ptr(c_weights), _ptr(None if geom is None else geom.buf), 0 if geom is None else int(geom.flags),
        _ptr(controls), _ptr(states),
        _ptr(costs), _ptr(weights), _ptr(best_cost), _ptr(best_states), NP, S, T, c, 0, float(dt), float(k_sigma),
        float(weight), float(temp),
        float(step_size), int(n_iters), int(seed) & (2 ** 64 - 1), int(iter0), _stream()), 'mpb_mppi_step')


@_on_tensor_device
def point_dynamics(x, u, ctrl_min, ctrl_max, dt, dyn_std=None, noise=None):
    """PointParticleDynamics.dynamics (point.py:102-140): x, u (..., dim) contiguous fp32 of the same shape -> x_next."""
    dim = x.shape[-1]
    _chk(x, tuple(x.shape), 'x')
    _chk(u, tuple(x.shape), 'u')
    _chk(ctrl_min, (dim,), 'ctrl_min')
    _chk(ctrl_max, (dim,), 'ctrl_max')
    if noise is not None:
        _chk(noise, tuple(x.shape), 'noise')
        _chk(dyn_std, (dim,), 'dyn_std')
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mpb_point_dynamics(_ptr(x), _ptr(u), _ptr(ctrl_min), _ptr(ctrl_max), _ptr(dyn_std if noise is not None else None),
                                            _ptr(noise), _ptr(out), x.numel() // dim, dim, float(dt), _stream()), 'mpb_point_dynamics')
    return out


@_on_tensor_device
def point_traj_cost(X, U, goal, discount, w_pos, w_vel, w_ctrl, w_pos_T, energy=0.0):
    """PointParticleDynamics.traj_cost (point.py:154-226): X (T,B,sd), U (T,B,cd), goal (sd), discount (T) -> costs (B)."""
    T, B, sd = X.shape
    cd = U.shape[-1]
    _chk(X, (T, B, sd), 'X')
    _chk(U, (T, B, cd), 'U')
    _chk(goal, (sd,), 'goal')
    _chk(discount, (T,), 'discount')
    out = torch.empty(B, device=X.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_point_traj_cost(_ptr(X), _ptr(U), _ptr(goal), _ptr(discount), float(w_pos), float(w_vel), float(w_ctrl),
                                             float(w_pos_T), float(energy), _ptr(out), T, B, sd, cd, _stream()), 'mpb_point_traj_cost')
    return out


@_on_tensor_device
def mvn_sample_dense(means, eps, tril_t, n, seed=0):
    """x = mean + L eps from a dense scale_tril handed over transposed (mpb_mvn_sample_dense): means (G,M) fp64, eps None or
    (n,G,M) fp64, tril_t (M,M) fp64 with tril_t[k,m] = L[m,k] -> (G*n, M) fp32, row mode * n + sample."""
    G, M = means.shape
    for t, nm in ((means, 'means'), (tril_t, 'tril_t')):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError(f'{nm} must be a contiguous CUDA float64 tensor')
    if tuple(tril_t.shape) != (M, M):
        raise ValueError('tril_t must be (M, M)')
    if eps is not None and not (eps.is_cuda and eps.dtype == torch.float64 and eps.is_contiguous() and tuple(eps.shape) == (n, G, M)):
        raise ValueError('eps must be a contiguous CUDA float64 tensor of shape (n, G, M)')
    out = torch.empty(G * n, M, device=means.device, dtype=torch.float32)
    _lib.check(_lib.lib().mpb_mvn_sample_dense(_ptr(out), _ptr(means), _ptr(eps), _ptr(tril_t), G, n, M,
                                              int(seed) & (2 ** 64 - 1), _stream()), 'mpb_mvn_sample_dense')
    return out


@_on_tensor_device
def gp_prior_sample(means, eps, Udiag, Uoff, n, D, seed=0, scale_tril=None, out=None):
    """Initial particles from the GP prior: means (G,H,2D) fp64, eps None or (n,G,H*2D) fp64 -> (G*n,H,2D) fp32.
    With `scale_tril` (2H,2H fp64, planners.base.gp_prior_scale_tril) and H <= 128 the product runs as a GEMM on
    the matrix cores; otherwise as the per-chain forward substitution."""
    G, H, dim = means.shape
    for t, nm in ((means, 'means'), (Udiag, 'Udiag'), (Uoff, 'Uoff')):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError(f'{nm} must be a contiguous CUDA float64 tensor')
    assert tuple(Udiag.shape) == (H, 3) and tuple(Uoff.shape) == (H - 1, 4) and dim == 2 * D
    if eps is not None:
        if not (eps.is_cuda and eps.dtype == torch.float64 and eps.is_contiguous() and tuple(eps.shape) == (n, G, H * dim)):
            raise ValueError('eps must be a contiguous CUDA float64 tensor of shape (n, G, H*2D)')
    if out is None:
        out = torch.empty(G * n, H, dim, device=means.device, dtype=torch.float32)
    else:
        _chk(out, (G * n, H, dim), 'out')
    if scale_tril is not None and H <= 128:
        if not (scale_tril.is_cuda and scale_tril.dtype == torch.float64 and scale_tril.is_contiguous()
                and tuple(scale_tril.shape) == (2 * H, 2 * H)):
            raise ValueError('scale_tril must be a contiguous CUDA float64 tensor of shape (2H, 2H)')
        _lib.check(_lib.lib().mpb_gp_prior_sample_dense(_ptr(out), _ptr(means), _ptr(eps), _ptr(scale_tril), G, n, H, D,
                                                       int(seed) & (2 ** 64 - 1), _stream()), 'mpb_gp_prior_sample_dense')
        return out
    _lib.check(_lib.lib().mpb_gp_prior_sample(_ptr(out), _ptr(means), _ptr(eps), _ptr(Udiag), _ptr(Uoff), G, n, H, D,
                                             int(seed) & (2 ** 64 - 1), _stream()), 'mpb_gp_prior_sample')
    return out


@_on_tensor_device
def stoch_gpmp_step(means, means64, samples, costs, weights, Udiag, Uoff, scale_tril, start, goal, geom, S, sig_cost,
                    sig_sample, dt, temperature, step_size, n_iters=1, seed=0):
    """n_iters StochGPMP iterations (device noise) enqueued by one C call: sample -> costs -> update."""
    P, H, dim = means.shape
    _chk(means, (P, H, dim), 'means')
    _chk(samples, (P * S, H, dim), 'samples')
    _chk(costs, (P, S), 'costs')
    _chk(weights, (P, S), 'weights')
    _chk(start, (P, dim), 'start')
    _chk(goal, (P, dim), 'goal')
    for t, nm, shp in ((means64, 'means64', (P, H, dim)), (Udiag, 'Udiag', (H, 3)), (Uoff, 'Uoff', (H - 1, 4))):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == shp):
            raise ValueError(f'{nm} must be a contiguous CUDA float64 tensor of shape {shp}')
    if scale_tril is not None and not (scale_tril.is_cuda and scale_tril.dtype == torch.float64 and scale_tril.is_contiguous()
                                       and tuple(scale_tril.shape) == (2 * H, 2 * H)):
        raise ValueError('scale_tril must be a contiguous CUDA float64 tensor of shape (2H, 2H)')
    _lib.check(_lib.lib().mpb_stoch_gpmp_step(
        _ptr(means), _ptr(means64), _ptr(samples), _ptr(costs), _ptr(weights), _ptr(Udiag), _ptr(Uoff), _ptr(scale_tril),
        _ptr(start), _ptr(goal), _ptr(geom.buf), P, S, H, dim // 2, float(dt), float(sig_cost[0]), float(sig_cost[1]),
        float(sig_cost[2]), float(sig_cost[3]), float(sig_sample[0]), float(sig_sample[1]), float(sig_sample[2]),
        float(temperature), float(step_size), int(n_iters), int(seed) & (2 ** 64 - 1), _stream()), 'mpb_stoch_gpmp_step')


@_on_tensor_device
def stoch_gpmp_costs(samples, means, start, goal, geom, costs, S, sig_cost, sig_sample, dt, temperature):
    """costs (P,S) of StochGPMP samples (P*S,H,2D): composite cost + importance term.
    sig_cost = (start, gp, goal_prior, coll); sig_sample = (start, gp, goal)."""
    B, H, dim = samples.shape
    P = B // S
    _chk(samples, (P * S, H, dim), 'samples')
    _chk(means, (P, H, dim), 'means')
    _chk(start, (P, dim), 'start')
    _chk(goal, (P, dim), 'goal')
    _chk(costs, (P, S), 'costs')
    _lib.check(_lib.lib().mpb_stoch_gpmp_costs(
        _ptr(samples), _ptr(means), _ptr(start), _ptr(goal), _ptr(geom.buf), _ptr(costs), P, S, H, dim // 2, float(dt),
        float(sig_cost[0]), float(sig_cost[1]), float(sig_cost[2]), float(sig_cost[3]), float(sig_sample[0]),
        float(sig_sample[1]), float(sig_sample[2]), float(temperature), _stream()), 'mpb_stoch_gpmp_costs')
