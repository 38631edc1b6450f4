"""UnaryFactor: isotropic Gaussian prior on ONE state (the start / goal anchors of GPMP2 and StochGPMP).

Drop-in for mp_baselines/planners/costs/factors/unary_factor.py:4-32 -- same constructor, the public attributes `sigma`,
`mean`, `K`, `dim`, `tensor_args`, `get_error`, `set_mean`.  The planners here never call it in their loops (the anchors are
folded into the block-tridiagonal solve, csrc/mpb_gpmp2.hip); it exists for callers that assemble factors themselves.

Design differences from the reference class: the weight matrix is held as its diagonal (`k_diag`; the dense `K` the
reference's callers index is materialised once from it), and the Jacobian d(error)/d(mean) = I is ONE resident identity
handed out as a broadcast view over the batch -- the reference builds and repeats a fresh (B, dim, dim) identity per call.
"""
import torch


class UnaryFactor:

    def __init__(self, dim, sigma, mean=None, tensor_args=None):
        ta = {} if tensor_args is None else tensor_args
        self.dim, self.sigma, self.tensor_args = dim, sigma, tensor_args
        self.k_diag = torch.ones(dim, **ta) / sigma ** 2             # precision per component: 1 / sigma^2
        self.K = torch.diag(self.k_diag)                              # (dim, dim), the values of eye / sigma^2
        self._identity = torch.eye(dim, **ta)
        self.mean = torch.zeros(dim, **ta) if mean is None else mean  # (aliases the caller's tensor, like the reference)

    def get_error(self, x, calc_jacobian=True):
        """x (B, 1, dim) or (B, dim) -> mean - x; with calc_jacobian also the (B, dim, dim) Jacobian (read-only view)."""
        residual = self.mean - x
        if not calc_jacobian:
            return residual
        batch = x.shape[0]
        return residual.reshape(batch, self.dim, 1), self._identity.expand(batch, self.dim, self.dim)

    def set_mean(self, x):
        self.mean = x.detach().clone()
