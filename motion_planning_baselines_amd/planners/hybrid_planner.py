"""HybridPlanner with the reference's class surface (mp_baselines/planners/hybrid_planner.py:10-89).

A sample-based planner proposes position-only paths (one polyline per particle, each with its own number
of waypoints, or None when it found nothing); they are turned into position+velocity support points and
handed to an optimisation-based planner (GPMP2 / StochGPMP of this package) that runs on the GPU.

The reference converts the paths with torch_robotics' ``smoothen_trajectory`` / ``tensor_linspace_v1``
(external, CPU, one path at a time in a Python loop: hybrid_planner.py:44-59).  Here ALL paths are converted
by one launch of mpb_traj_resample (arc-length-uniform linear resampling, average velocity on the interior
points -- build-defined); the sample-based planner itself is the caller's (duck-typed:
``optimize(refill_samples_buffer=True, ...) -> list of (n_i, D) tensors or None``, ``start_state_pos``,
``goal_state_pos``) -- RRT is outside this build's scope.
"""
import torch

from .. import ops
from .base import MPPlanner, require_cuda


class HybridPlanner(MPPlanner):

    def __init__(self, sample_based_planner, opt_based_planner, **kwargs):
        super().__init__("HybridSampleAndOptimizationPlanner", **kwargs)
        self.sample_based_planner = sample_based_planner
        self.opt_based_planner = opt_based_planner
        self.device = require_cuda(self.tensor_args)

    def render(self, ax, **kwargs):
        raise NotImplementedError

    def paths_to_initial_means(self, traj_l):
        """list of (n_i, D) position paths (or None) -> (1, N, H, 2D) initial particle means
        (hybrid_planner.py:42-66; a missing path becomes the straight line start -> goal, :47-51)."""
        H, dt = self.opt_based_planner.n_support_points, self.opt_based_planner.dt
        paths = []
        for traj in traj_l:
            if traj is None:
                traj = torch.stack((torch.as_tensor(self.sample_based_planner.start_state_pos),
                                    torch.as_tensor(self.sample_based_planner.goal_state_pos)))
            paths.append(torch.as_tensor(traj, dtype=torch.float32).reshape(-1, traj.shape[-1]))
        N, D = len(paths), paths[0].shape[-1]
        Lmax = max(p.shape[0] for p in paths)
        padded = torch.zeros(N, Lmax, D, dtype=torch.float32)
        for i, p in enumerate(paths):
            padded[i, :p.shape[0]] = p.to('cpu') if not p.is_cuda else p.cpu()
        lengths = torch.tensor([p.shape[0] for p in paths], dtype=torch.int32)
        out = ops.traj_resample(padded.to(self.device), lengths.to(self.device), H, dt)
        return out.unsqueeze(0)                                   # 'n h d -> 1 n h d' (:64)

    def optimize(self, debug=False, print_times=False, return_iterations=False, **kwargs):
        traj_l = self.sample_based_planner.optimize(refill_samples_buffer=True, debug=debug, **kwargs)
        self.opt_based_planner.reset(initial_particle_means=self.paths_to_initial_means(traj_l))
        trajs_0 = self.opt_based_planner.get_traj()
        n = self.opt_based_planner.opt_iters
        trajs_iters = torch.empty((n + 1, *trajs_0.shape), device=trajs_0.device, dtype=trajs_0.dtype)
        trajs_iters[0] = trajs_0
        for i in range(n):
            trajs_iters[i + 1] = self.opt_based_planner.optimize(opt_iters=1, debug=debug, **kwargs)
        return trajs_iters if return_iterations else trajs_iters[-1]
