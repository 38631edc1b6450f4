"""GPFactor (mp_baselines/planners/costs/factors/gp_factor.py:4-65): constant-velocity GP prior between consecutive
states -- Phi, Q^-1, the error x_{t+1} - Phi x_t and its constant Jacobians."""
import torch

from .... import ops


class GPFactor:

    def __init__(self, dim, sigma, d_t, num_factors, tensor_args=None, Q_c_inv=None):
        self.dim = dim
        self.d_t = d_t
        self.tensor_args = tensor_args
        self.state_dim = self.dim * 2
        self.num_factors = num_factors
        dev = tensor_args['device']
        self.idx1 = torch.arange(0, self.num_factors, device=dev)
        self.idx2 = torch.arange(1, self.num_factors + 1, device=dev)
        self.phi = self.calc_phi()
        if Q_c_inv is None:
            Q_c_inv = torch.eye(dim, **tensor_args) / sigma ** 2
        self.Q_c_inv = torch.zeros(num_factors, dim, dim, **tensor_args) + Q_c_inv
        self.Q_inv = self.calc_Q_inv()                      # (num_factors, 2D, 2D)
        self.H1 = self.phi.unsqueeze(0).repeat(self.num_factors, 1, 1)
        self.H2 = -1. * torch.eye(self.state_dim, **self.tensor_args).unsqueeze(0).repeat(self.num_factors, 1, 1)

    def calc_phi(self):
        """gp_factor.py:34-40 (set-up constants: a handful of torch calls at construction)."""
        I = torch.eye(self.dim, **self.tensor_args)
        Z = torch.zeros(self.dim, self.dim, **self.tensor_args)
        return torch.cat((torch.cat((I, self.d_t * I), dim=1), torch.cat((Z, I), dim=1)), dim=0)

    def calc_Q_inv(self):
        """gp_factor.py:42-50."""
        m1 = 12. * (self.d_t ** -3.) * self.Q_c_inv
        m2 = -6. * (self.d_t ** -2.) * self.Q_c_inv
        m3 = 4. * (self.d_t ** -1.) * self.Q_c_inv
        return torch.cat((torch.cat((m1, m2), dim=-1), torch.cat((m2, m3), dim=-1)), dim=-2)

    def get_error(self, x_traj, calc_jacobian=True):
        """gp_factor.py:52-65: error (B, num_factors, 2D, 1) on the GPU (mpb_gp_factor_error)."""
        assert x_traj.shape[1] == self.num_factors + 1 and x_traj.shape[2] == self.state_dim
        error = ops.gp_factor_error(x_traj.to(torch.float32).contiguous(), self.dim, self.d_t).unsqueeze(-1)
        if calc_jacobian:
            return error, self.H1, self.H2
        return error
