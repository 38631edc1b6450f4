"""STOMP with the reference's class surface (mp_baselines/planners/stomp.py), iterations on the GPU.

Host side (this file): constants R / Sigma / scale_tril computed with the same torch calls as the
reference (stomp.py:63-64, :68-95 -- SURVEY.md H2: they are ill-conditioned in fp32 and must be
bit-identical), buffer ownership, noise-source selection.  Device side: mpb_stomp_step
(csrc/mpb_kernels.hip) runs sample -> cost -> softmax -> covariance-weighted update for all
``opt_iters`` iterations without returning to Python.
"""
import time

import torch

from .. import ops
from .._lib import MPBError
from .base import OptimizationPlanner
from .costs.cost_functions import device_plan, fusable_collision


def stomp_precision_matrix(n_support_points, dt, sigma_spectral, tensor_args):
    """R = A^T A with A the (H+2) x H second-difference operator (unit corner entries) scaled by
    sigma_spectral / dt^2 -- STOMP._get_R_mat (stomp.py:68-86)."""
    H = n_support_points
    inner = -2 * torch.eye(H) + torch.diag(torch.ones(H - 1), 1) + torch.diag(torch.ones(H - 1), -1)
    A = torch.cat((torch.zeros(1, H), inner, torch.zeros(1, H)), dim=0)
    A[0, 0] = 1.
    A[-1, -1] = 1.
    A = A * 1. / dt ** 2 * sigma_spectral
    return (A.t() @ A).to(**tensor_args)


def precision_to_scale_tril(P):
    """scale_tril L (L L^T = P^-1) exactly as MultivariateNormal(precision_matrix=P) derives it
    (torch/distributions/multivariate_normal.py:80-86)."""
    Lf = torch.linalg.cholesky(torch.flip(P, (-2, -1)))
    L_inv = torch.transpose(torch.flip(Lf, (-2, -1)), -2, -1)
    Id = torch.eye(P.shape[-1], dtype=P.dtype, device=P.device)
    return torch.linalg.solve_triangular(L_inv, Id, upper=False).contiguous()   # (some LAPACK back-ends hand back a transposed view)


_current_device = getattr(torch._C, '_cuda_getDevice', torch.cuda.current_device)     # (the raw getter: no lazy-init bookkeeping per call)
_perf_counter = time.perf_counter
_SYNC_SPIN_S = 1e-3          # check='sync': how long optimize() polls (the launch's status block, or the stream) before it blocks in the runtime


class PersistentLaunchLost(MPBError):
    """A persistent (one-launch) optimisation loop was abandoned by the device side: see STOMP's `check` argument."""


class STOMP(OptimizationPlanner):
    """Drop-in for mp_baselines.planners.stomp.STOMP (ctor kwargs stomp.py:10-32).

    Extra keyword arguments (not in the reference):
      noise: 'philox' (default) -- standard normals generated inside the kernel (counter-based, result
             independent of sharding); 'torch_cpu' -- drawn per iteration with the CPU generator in the
             reference's (S,d,P,H) order, so that `torch.manual_seed(s)` reproduces a CPU reference run
             bit for bit in the noise; 'torch' -- same draw on the planner's device.
      seed / particle_offset: Philox key and global index of this shard's first particle.
      persistent: run a collision-only cost's loop as ONE persistent launch (mpb_stomp_run) where the shape allows it
             (default); False keeps the two-kernels-per-iteration path (mpb_stomp_step).
      check: (the default changed from 'deferred' to 'sync' with ABI 4: optimize() now waits for its launch -- it polls the
             launch's status block for at most 1 ms, then blocks in the runtime, GIL released.)  What happens when a persistent launch is LOST (include/mpb.h, "Failure contract": its workgroups wait for
             each other, every wait is bounded, and a wait that runs out -- the device stopped starting this launch's
             workgroups for seconds, e.g. another context holds every CU -- abandons the call).  'sync' (default):
             optimize() waits until the launch has PUBLISHED its completion tag (host spin on the launch's pinned status block, no sleep;
             every result store of the launch is ordered before the tag, but the kernel may not have retired yet: work on
             the current stream is ordered behind it as always, a consumer on another stream must wait on this stream) and
             raises PersistentLaunchLost before it returns -- a caller never holds the result of a lost call.  'deferred' (opt-in, for callers that queue more
             work behind optimize() and synchronise themselves, e.g. bench.py): optimize() returns without waiting and the
             loss is raised at the NEXT call into the planner (optimize / reset / sample / get_traj / persistent_timed_out);
             the kernel reports into pinned host memory, so that check costs no synchronisation -- but a program whose
             LAST planner call is optimize() must call persistent_timed_out() (or get_traj()) itself before it trusts the
             returned tensor, which after a lost call holds the means from BEFORE that call.
             After the exception the means of the particles the lost call did not finish are the ones from before that
             call; samples / costs / weights are undefined: reset() the planner (or re-run from saved means).
    """

    def __init__(self, n_dof, n_support_points, num_particles_per_goal, num_samples, opt_iters, dt, start_state,
                 cost=None, initial_particle_means=None, multi_goal_states=None, sigma_start_init=0.001,
                 sigma_goal_init=0.001, sigma_gp_init=10., temperature=1., step_size=1., sigma_spectral=0.1,
                 goal_state=None, pos_only=True, tensor_args=None, noise='philox', seed=0, particle_offset=0,
                 persistent=True, check='sync', **kwargs):
        super().__init__(name='STOMP', n_dof=n_dof, n_support_points=n_support_points,
                         num_particles_per_goal=num_particles_per_goal, opt_iters=opt_iters, dt=dt,
                         start_state=start_state, cost=cost, initial_particle_means=initial_particle_means,
                         multi_goal_states=multi_goal_states, sigma_start_init=sigma_start_init,
                         sigma_goal_init=sigma_goal_init, sigma_gp_init=sigma_gp_init, pos_only=pos_only,
                         tensor_args=tensor_args)
        assert noise in ('philox', 'torch', 'torch_cpu')
        assert check in ('deferred', 'sync')
        self.check = check
        self._status = None              # host-visible status block of the persistent launches (allocated on first use)
        self._plan = None                # validated, pre-converted arguments of the persistent launch (ops.StompRunPlan)
        self._last_tag = 0               # tag of the persistent launch the current optimize() call has made (0: none)
        self._spare_copy = None          # the next call's return buffer (allocated while the previous launch runs)
        self._eps_ring = None            # noise='torch_cpu': pinned host / device double buffer of one iteration's normals
        self._eps_dev = None             # noise='torch': device chunk buffer of normals
        self._traj_out = None
        self.lr = step_size
        self.sigma_spectral = sigma_spectral
        self.start_state = start_state          # quirk Q10: overwrites the zero-velocity-extended state
        self.goal_state = goal_state
        self.num_samples = num_samples
        self.temperature = temperature
        self.noise = noise
        self.seed = int(seed)
        self.particle_offset = int(particle_offset)
        self._iter = 0
        self._weights = None
        self._run_ws = None              # exchange buffer of the persistent kernel (allocated on first use)
        self.persistent = bool(persistent)
        # constants on the CPU with the reference's own op sequence, then moved (H2)
        cpu = dict(device='cpu', dtype=torch.float32)
        R = stomp_precision_matrix(n_support_points, dt, sigma_spectral, cpu)
        self.Sigma_inv = R.to(self.device)
        self.Sigma = torch.inverse(R).to(self.device).contiguous()
        self.scale_tril = precision_to_scale_tril(R).to(self.device).contiguous()
        P, S, H, d = self.num_particles, num_samples, n_support_points, self.d_state_opt
        self.state_particles = torch.empty(P, S, H, d, device=self.device, dtype=torch.float32)
        self.costs = torch.zeros(P, S, device=self.device, dtype=torch.float32)
        self._weights_buf = torch.empty(P, S, device=self.device, dtype=torch.float32)
        self.reset(initial_particle_means=initial_particle_means)
        self.best_cost = torch.inf

    # ---- constants ---------------------------------------------------------------------------
    def _get_R_mat(self):
        """stomp.py:68-86."""
        return stomp_precision_matrix(self.n_support_points, self.dt, self.sigma_spectral,
                                      dict(device='cpu', dtype=torch.float32)).to(self.device)

    def set_noise_dist(self):
        """stomp.py:88-95: the noise distribution is N(0, Sigma_inv^-1); here that is its scale_tril, derived exactly
        as MultivariateNormal(precision_matrix=...) derives it."""
        self.scale_tril = precision_to_scale_tril(self.Sigma_inv.detach().cpu().float()).to(self.device).contiguous()

    def const_vel_trajectory(self, start_state, goal_state):
        """stomp.py:122-135: straight line start -> goal, (H, d_state_opt)."""
        H, D = self.n_support_points, self.n_dof
        start_state, goal_state = start_state.detach().cpu().float(), goal_state.detach().cpu().float()
        n = H - 1
        traj = torch.zeros(H, self.d_state_opt)
        for i in range(H):
            traj[i, :D] = start_state[:D] * (n - i) * 1. / n + goal_state[:D] * i * 1. / n
        if not self.pos_only:
            traj[:, D:] = ((goal_state[:D] - start_state[:D]) / (n * self.dt)).unsqueeze(0)
        return traj.to(self.device)

    def _calc_sample_weights(self, costs):
        """stomp.py:219-220: softmax(-costs / T) over the samples of each particle, (P, S, 1, 1).  Served by the
        update kernel with a zero step (the means stay untouched)."""
        costs = costs.reshape(self.num_particles, self.num_samples).to(torch.float32).contiguous()
        w = torch.empty_like(costs)
        ops.stomp_update(self._particle_means, self.state_particles, costs, w, self.Sigma, 0.0, self.temperature)
        return w.reshape(self.num_particles, self.num_samples, 1, 1)

    # ---- noise -------------------------------------------------------------------------------
    def _draw_eps(self, n_iters):
        """Standard normals in the reference's draw order: one (S,d,P,H) block per iteration."""
        if self.noise == 'philox':
            return None
        S, d, P, H = self.num_samples, self.d_state_opt, self.num_particles, self.n_support_points
        dev = 'cpu' if self.noise == 'torch_cpu' else self.device
        blocks = [torch.empty(S, d, P, H, device=dev, dtype=torch.float32).normal_() for _ in range(n_iters)]
        return torch.stack(blocks).to(self.device).contiguous()

    def sample(self):
        """stomp.py:97-108: fresh state_particles (P,S,H,d) around the current means."""
        self._raise_if_lost()
        eps = self._draw_eps(1)
        ops.stomp_sample(self._particle_means, None if eps is None else eps[0], self.state_particles,
                         self.scale_tril, self.num_samples, seed=self.seed, it=self._iter,
                         particle_offset=self.particle_offset)
        self._iter += 1
        return self.state_particles

    def reset(self, initial_particle_means=None):
        """stomp.py:110-120."""
        self._raise_if_lost()
        if initial_particle_means is not None:
            m = initial_particle_means.clone()
        else:
            m = self.get_random_trajs()
        self._particle_means = m.to(device=self.device, dtype=torch.float32).contiguous()
        self.state_particles = self.sample()

    # ---- optimisation ------------------------------------------------------------------------
    def optimize(self, opt_iters=None, **observation):
        """stomp.py:137-148: run the iterations, return the current trajectory (P,H,d) -- a copy of the means
        (base.py:204-213); where the persistent launch runs, the copy is written by that launch itself."""
        self._traj_out = None
        self._last_tag = 0
        self._run_optimization(opt_iters, **observation)
        if self.check == 'sync':
            self._raise_if_lost(synchronize=True, tag=self._last_tag)
        if self._traj_out is not None:
            out, self._traj_out = self._traj_out, None
            return out
        return self._get_traj()

    def get_traj(self):
        self._raise_if_lost()
        return self._get_traj()

    def _raise_if_lost(self, synchronize=False, tag=0):
        """Raise PersistentLaunchLost if a persistent launch issued by this planner has been abandoned (class docstring,
        `check`).  Reads pinned host memory the kernel writes: no device synchronisation unless asked for.
        synchronize: wait for the call that has just been enqueued -- `tag`: the persistent launch it made (0: none)."""
        if self._status is None:
            return
        if synchronize:
            # host spin for a BOUNDED time (the runtime's blocking wait wakes up tens of microseconds late, which is what a
            # short call feels), then the blocking wait: a long optimisation must not pin a core and hold the GIL against
            # the caller's other threads (thread-per-GPU drivers, watchdogs) for its whole duration.
            # A persistent launch is waited for on its own STATUS BLOCK: the last workgroup out publishes the call's tag to
            # pinned host memory (csrc/mpb_stomp_fused.h, fused_leave: word 0 completed, word 1 lost) -- no marker packet
            # behind the kernel (an event costs ~3 us of queue processing after the kernel's end, and ~3 us of host calls to
            # create, record and query).  Work enqueued later on the stream is ordered behind the kernel as always.
            t_end = _perf_counter() + _SYNC_SPIN_S
            stream = torch.cuda.current_stream(self.device)
            if tag:
                view = self._status._view
                while int(view[0]) != tag and int(view[1]) != tag:
                    if _perf_counter() > t_end:
                        stream.synchronize()     # (releases the GIL while it blocks)
                        break
            else:
                while not stream.query():
                    if _perf_counter() > t_end:
                        stream.synchronize()
                        break
        lost = self._status.lost()
        if lost is not None:
            tag, why = lost
            self._status.acknowledge(tag)
            raise PersistentLaunchLost(
                'the persistent STOMP launch (tag 0x%08x) was abandoned: %s.  The means of the particles it did not finish '
                'are unchanged from before that optimize() call; samples, costs and weights are undefined.  reset() the '
                'planner (or restore saved means) and run again; persistent=False selects the two-kernel path, which '
                'has no inter-workgroup waits.' % (tag, {
                    1: 'a workgroup waited longer than the bound for the partner workgroups of its particle (the device did '
                       'not start them: is another context or stream holding the whole GPU?)',
                    2: 'the workspace header was not zero before the first call'}.get(why, 'reason %d' % why)))

    def _run_optimization(self, opt_iters, **observation):
        if opt_iters is None:
            opt_iters = self.opt_iters
        self._raise_if_lost()
        fused = fusable_collision(self.cost)
        if fused is not None and not observation:
            cc, weight = fused
            geom = cc.device_geometry(self.device)
            if self.persistent and self._run_ws is None:
                self._run_ws = ops.stomp_workspace(self.num_particles, self.num_samples, self.n_support_points,
                                                   self.d_state_opt, self.device)
                self._status = ops.StompRunStatus()
            # the copy optimize() returns (base.py:204-213) is written by the launch itself (pos_only: _get_traj appends
            # finite-difference velocities, from the means)
            copy = None
            if not self.pos_only:
                # (a buffer allocated behind the PREVIOUS launch, while that kernel ran: every call still returns a tensor of its
                # own, and the allocator's ~2 us are not in front of this launch)
                copy, self._spare_copy = self._spare_copy, None
                m = self._particle_means
                if copy is None or copy.shape != m.shape or copy.device != m.device or copy.dtype != m.dtype:
                    copy = torch.empty_like(m)
            if (self.persistent and self.noise == 'philox' and self._particle_means.is_cuda
                    and _current_device() == self._particle_means.device.index):
                # device noise: nothing changes between calls but the iteration counter -- arguments validated once
                key = (self._particle_means.data_ptr(), self.state_particles.data_ptr(), self.costs.data_ptr(),
                       self._weights_buf.data_ptr(), self.scale_tril.data_ptr(), self.Sigma.data_ptr(), geom.buf.data_ptr(),
                       self._run_ws.data_ptr(), self.num_samples, self.n_dof, float(cc.k_sigma), float(weight), float(self.lr),
                       float(self.temperature), self.seed, self.particle_offset)
                plan = self._plan
                if plan is None or plan.key != key:
                    plan = self._plan = ops.StompRunPlan(
                        self._particle_means, self.state_particles, self.costs, self._weights_buf, self.scale_tril,
                        self.Sigma, geom, self.num_samples, self.n_dof, cc.k_sigma, weight, self.lr, self.temperature,
                        self._run_ws, self.seed, self.particle_offset, self._status)
                self._last_tag = plan.launch(opt_iters, self._iter, copy)
                if copy is not None:
                    self._spare_copy = torch.empty_like(copy)
            else:
                # the whole loop as one persistent launch where the shape allows it (H = 64, S <= 64, grid-backed fields);
                # mpb_stomp_run falls back to the two-kernel loop by itself otherwise
                self._last_tag = self._run_fused_injected(opt_iters, geom, cc, weight, copy)
            self._traj_out = copy
            self._iter += opt_iters
        elif not observation and device_plan(self.cost, self.device) is not None:
            # composite of HIP-served members: sample(+collision) kernel -> trajectory-terms kernel(s) -> update
            # kernel, all queued on the stream without a host round trip
            cc, weight, groups = device_plan(self.cost, self.device)
            flat = self.state_particles.view(-1, self.n_support_points, self.d_state_opt)
            for _ in range(opt_iters):
                eps = self._draw_eps(1)
                ops.stomp_sample(self._particle_means, None if eps is None else eps[0], self.state_particles,
                                 self.scale_tril, self.num_samples, seed=self.seed, it=self._iter,
                                 particle_offset=self.particle_offset,
                                 geom=None if cc is None else cc.device_geometry(self.device),
                                 costs=None if cc is None else self.costs,
                                 k_sigma=0.0 if cc is None else cc.k_sigma, weight=weight)
                self._iter += 1
                for gi, spec in enumerate(groups):
                    ops.cost_terms_eval(flat, self.n_dof, out=self.costs, accumulate=(cc is not None or gi > 0), **spec)
                ops.stomp_update(self._particle_means, self.state_particles, self.costs, self._weights_buf,
                                 self.Sigma, self.lr, self.temperature)
        else:
            # caller-supplied cost callable: sample kernel -> user cost on device tensors -> update kernel
            for _ in range(opt_iters):
                self.costs = self._sample_and_eval(**observation)
                self._update_distribution(self.costs, self.state_particles)
        self._weights = self._weights_buf.reshape(self.num_particles, self.num_samples, 1, 1)

    def _run_fused_injected(self, opt_iters, geom, cc, weight, copy):
        """The fused loop (mpb_stomp_run) outside the device-noise fast path: the IDENTICAL-SEED modes noise = 'torch_cpu' / 'torch'
        (stomp.py:97-108 draws one (S,d,P,H) block of standard normals per iteration through the torch generator: the same
        `normal_()` calls in the same order here, so `torch.manual_seed(s)` reproduces a reference run's noise bit for bit), and
        device noise on the two-kernel path.  Returns the tag of the last persistent launch (0: none).

        torch_cpu: the reference's generator runs on the HOST -- 3.7 M normals per C3 iteration, serial by construction (one
        Mersenne-Twister stream) -- so the loop is a two-stage pipeline: block k of a two-deep ring of PINNED host buffers is
        drawn while the GPU copies and consumes block k - 1 (async H2D on the launch stream, one n_iters = 1 launch per
        iteration); an iteration costs max(host draw, H2D + kernel) = the host draw.  (Until round 5: all K blocks drawn,
        stacked, copied from pageable memory, then one launch -- draw + stack + copy + kernel in series.)
        torch: the draw is a device kernel per iteration into a chunk buffer, one launch per chunk of up to 16 iterations."""
        run = lambda eps, n, it0, cp: ops.stomp_run(
            self._particle_means, eps, self.state_particles, self.costs, self._weights_buf, self.scale_tril, self.Sigma, geom,
            self.num_samples, self.n_dof, cc.k_sigma, weight, self.lr, self.temperature, self._run_ws if self.persistent else None,
            n_iters=n, seed=self.seed, iter0=it0, particle_offset=self.particle_offset,
            status=self._status if self.persistent else None, means_copy=cp)
        if self.noise == 'philox' or opt_iters <= 0:
            return run(None, opt_iters, self._iter, copy)
        S, d, P, H = self.num_samples, self.d_state_opt, self.num_particles, self.n_support_points
        tag = 0
        if self.noise == 'torch':
            chunk = min(opt_iters, 16)
            buf = self._eps_dev
            if buf is None or buf.shape != (chunk, S, d, P, H) or buf.device != self.device:
                buf = self._eps_dev = torch.empty(chunk, S, d, P, H, device=self.device, dtype=torch.float32)
            done = 0
            while done < opt_iters:
                n = min(chunk, opt_iters - done)
                for i in range(n):
                    buf[i].normal_()                    # one generator call per iteration, like the reference
                tag = run(buf[:n], n, self._iter + done, copy if done + n == opt_iters else None)
                done += n
            return tag
        ring = self._eps_ring
        if ring is None or ring['host'][0].shape != (1, S, d, P, H) or ring['dev'][0].device != self.device:
            ring = self._eps_ring = dict(
                host=[torch.empty(1, S, d, P, H, dtype=torch.float32).pin_memory() for _ in range(2)],
                dev=[torch.empty(1, S, d, P, H, device=self.device, dtype=torch.float32) for _ in range(2)],
                copied=[torch.cuda.Event() for _ in range(2)])
        for it in range(opt_iters):
            k = it & 1
            ring['copied'][k].synchronize()             # the copy that last read host block k has finished (no-op before its first use)
            ring['host'][k].normal_()                   # the reference's draw, on the CPU generator
            ring['dev'][k].copy_(ring['host'][k], non_blocking=True)
            ring['copied'][k].record()
            tag = run(ring['dev'][k], 1, self._iter + it, copy if it == opt_iters - 1 else None)
        return tag

    def persistent_timed_out(self):
        """Was the last persistent launch lost (class docstring, `check`)?  Synchronises the stream and reads the
        workspace header; False when no persistent launch was made.  Does not raise -- and a loss reported here is
        not raised again later."""
        if self._run_ws is None or not self.persistent:
            return False
        lost = ops.stomp_run_timed_out(self._run_ws)
        if lost and self._status is not None:
            hit = self._status.lost()
            if hit is not None:
                self._status.acknowledge(hit[0])
        return lost

    def optimize_timed(self, opt_iters):
        """Measurement aid (bench.py): optimize(opt_iters) on the persistent path with the kernel's own duration taken on the
        dispatch (mpb_stomp_run_timed); synchronises and returns milliseconds (None when the planner is not on that path)."""
        self._run_optimization(0)                      # builds the workspace / plan exactly as optimize() would
        if self._plan is None:
            return None
        ms = self._plan.launch_timed(opt_iters, self._iter)
        self._iter += opt_iters
        return ms if ms > 0.0 else None

    def run_path(self):
        """Which form of the loop optimize() takes for this planner's shape and cost (ops.STOMP_PATH_*): the persistent
        launch with or without an exchange between workgroups, or the two-kernel loop."""
        fused = fusable_collision(self.cost)
        if fused is None or not self.persistent:
            return ops.STOMP_PATH_TWO_KERNEL
        ws = self._run_ws if self._run_ws is not None else ops.stomp_workspace(
            self.num_particles, self.num_samples, self.n_support_points, self.d_state_opt, self.device)
        return ops.stomp_run_path(fused[0].device_geometry(self.device), ws, self.num_particles, self.num_samples,
                                  self.n_support_points, self.d_state_opt)

    def _sample_and_eval(self, **observation):
        """stomp.py:162-197."""
        self.state_particles = self.sample()
        costs = self._get_costs(self.state_particles.flatten(0, 1), **observation)
        return costs.reshape(self.num_particles, self.num_samples).to(torch.float32).contiguous()

    def _update_distribution(self, costs, traj_particles):
        """stomp.py:199-211 (softmax over samples, covariance-weighted mean update)."""
        ops.stomp_update(self._particle_means, traj_particles, costs, self._weights_buf, self.Sigma, self.lr,
                         self.temperature)
        self._weights = self._weights_buf.reshape(self.num_particles, self.num_samples, 1, 1)
