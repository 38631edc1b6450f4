"""Control-trajectory covariance priors of MPPI (mp_baselines/planners/priors/gaussian.py:143-216): host-side set-up
constants, computed with the same torch calls as the reference."""
import numpy as np
import torch


def diag_Cov(sigma, length, ctrl_dim, tensor_args):
    """Time-independent diagonal covariance (gaussian.py:143-163); (T,T,c)."""
    Cov = torch.eye(length, **tensor_args).unsqueeze(-1).repeat(1, 1, ctrl_dim)
    if isinstance(sigma, (list, tuple)):
        return Cov * torch.Tensor(np.array(sigma)).to(**tensor_args) ** 2
    return Cov * sigma ** 2


def const_ctrl_Cov(sigma, length, ctrl_dim, tensor_args):
    """Constant-control covariance prior (gaussian.py:166-198); (T,T,c)."""
    if isinstance(sigma, (list, tuple)):
        sigma = torch.from_numpy(np.array(sigma)).to(**tensor_args)
    L = torch.tril(torch.ones(length, length - 1, **tensor_args), diagonal=-1)
    LL_t = torch.matmul(L, L.transpose(0, 1)) + torch.ones(length, length, **tensor_args)
    return LL_t.unsqueeze(-1).repeat(1, 1, ctrl_dim) * sigma ** 2


def check_Cov_is_valid(Cov):
    """gaussian.py:201-216: refuse covariances whose determinant would underflow the Gaussian's normaliser."""
    Cov_np = Cov.cpu().numpy()
    for i in range(Cov.shape[-1]):
        if np.linalg.det(Cov_np[:, :, i]) < 1.e-7:
            raise ZeroDivisionError('Covariance-determinant too small, potential for underflow.  Consider increasing sigma.')
