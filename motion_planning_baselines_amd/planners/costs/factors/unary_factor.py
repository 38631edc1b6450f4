"""UnaryFactor (mp_baselines/planners/costs/factors/unary_factor.py:4-32): Gaussian prior on one state."""
import torch


class UnaryFactor:

    def __init__(self, dim, sigma, mean=None, tensor_args=None):
        self.sigma = sigma
        self.mean = torch.zeros(dim, **tensor_args) if mean is None else mean
        self.tensor_args = tensor_args
        self.K = torch.eye(dim, **tensor_args) / sigma ** 2
        self.dim = dim

    def get_error(self, x, calc_jacobian=True):
        """mean - x (a single element-wise difference of a (B, 1, dim) slice: device-tensor plumbing)."""
        error = self.mean - x
        if calc_jacobian:
            H = torch.eye(self.dim, **self.tensor_args).unsqueeze(0).repeat(x.shape[0], 1, 1)
            return error.view(x.shape[0], self.dim, 1), H
        return error

    def set_mean(self, x):
        self.mean = x.clone().detach()
