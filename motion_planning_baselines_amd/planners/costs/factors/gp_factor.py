"""GPFactor: constant-velocity Gauss-Markov prior between consecutive states -- the GP-prior finite-difference smoothness
term.  Drop-in for mp_baselines/planners/costs/factors/gp_factor.py:4-65: same constructor, public attributes (`phi`,
`Q_c_inv`, `Q_inv`, `H1`, `H2`, `idx1`, `idx2`, `state_dim`, ...) and `get_error`.

    Phi = [[I, dt I], [0, I]],   Q^-1 = [[12 / dt^3, -6 / dt^2], [-6 / dt^2, 4 / dt]] (x) Q_c^-1,   e_t = x_{t+1} - Phi x_t

Design differences from the reference class: the error is one HIP pass over the batch (mpb_gp_factor_error: no index_select
copies, no batched 2D x 2D matmul); Q^-1 is formed as the Kronecker product it is -- one broadcast multiply of the 2 x 2
coefficient block with the per-factor Q_c^-1 instead of three scaled copies and three concatenations (the same single rounding
per entry: fl(coefficient) * q); the constant Jacobians are broadcast VIEWS of one (2D, 2D) matrix each, not per-factor copies.
"""
import torch

from .... import ops


class GPFactor:

    def __init__(self, dim, sigma, d_t, num_factors, tensor_args=None, Q_c_inv=None):
        self.dim, self.state_dim = dim, 2 * dim
        self.d_t, self.num_factors, self.tensor_args = d_t, num_factors, tensor_args
        steps = torch.arange(num_factors + 1, device=tensor_args['device'])
        self.idx1, self.idx2 = steps[:-1], steps[1:]                      # factor t joins states t and t + 1
        self.phi = self.calc_phi()
        power = torch.diag(torch.ones(dim, **tensor_args) / sigma ** 2) if Q_c_inv is None else torch.as_tensor(Q_c_inv, **tensor_args)
        self.Q_c_inv = power.expand(num_factors, dim, dim).contiguous()   # one power-spectral block per factor
        self.Q_inv = self.calc_Q_inv()                                    # (num_factors, 2D, 2D)
        n = self.state_dim
        self.H1 = self.phi.expand(num_factors, n, n)                      # d e_t / d x_t     (sign convention of the reference)
        self.H2 = (-torch.eye(n, **tensor_args)).expand(num_factors, n, n)     # d e_t / d x_{t+1}

    def calc_phi(self):
        """State transition of the constant-velocity model over one step (gp_factor.py:34-40)."""
        D = self.dim
        phi = torch.eye(2 * D, **self.tensor_args)
        phi[:D, D:].fill_diagonal_(self.d_t)
        return phi

    def calc_Q_inv(self):
        """Inverse process noise of one step (gp_factor.py:42-50) as coefficient block (x) Q_c^-1."""
        dt, D = self.d_t, self.dim
        coef = torch.tensor([[12. * (dt ** -3.), -6. * (dt ** -2.)], [-6. * (dt ** -2.), 4. * (dt ** -1.)]], **self.tensor_args)
        blocks = coef.view(1, 2, 1, 2, 1) * self.Q_c_inv.view(-1, 1, D, 1, D)          # [f, a, i, b, j] = coef[a, b] Q_c^-1[f, i, j]
        return blocks.reshape(-1, 2 * D, 2 * D)

    def get_error(self, x_traj, calc_jacobian=True):
        """(B, num_factors + 1, 2D) -> e (B, num_factors, 2D, 1) [, H1, H2]  (gp_factor.py:52-65), on the GPU."""
        assert x_traj.shape[1] == self.num_factors + 1 and x_traj.shape[2] == self.state_dim
        error = ops.gp_factor_error(x_traj.to(torch.float32).contiguous(), self.dim, self.d_t).unsqueeze(-1)
        return (error, self.H1, self.H2) if calc_jacobian else error
