"""Control-trajectory priors of MPPI with the reference's class surface (mp_baselines/planners/priors/gaussian.py).

The covariance builders are host-side set-up (the reference's own torch calls); `ControlTrajectoryGaussian.sample` draws
mean + scale_tril @ eps on the GPU (mpb_mvn_sample_dense, fp64 inside, one launch per control dimension).  The MPPI
planner fuses the same sampling into its one-launch loop (mpb_mppi_step)."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from ... import ops
from ..base import require_cuda


class GMM:
    """priors/gaussian.py:7-33: mixture of independent Gaussians over control trajectories -- means / sigmas
    (num_particles, steps, ctrl_dim), weights (num_particles) -- on whatever device the tensors live (no planner of the
    reference uses it; kept for code written against the module)."""

    def __init__(self, means, sigmas, weights):
        import torch.distributions as dist
        self.num_particles, self.rollout_steps, self.ctrl_dim = means.shape
        components = dist.Independent(dist.Normal(means, sigmas), 2)
        self.dist = dist.mixture_same_family.MixtureSameFamily(dist.Categorical(weights), components)

    def sample(self, num_samples):
        """(steps-first transpose as the reference, :29-30; num_samples: a torch.Size / tuple like there, or an int)"""
        shape = (num_samples,) if isinstance(num_samples, int) else tuple(num_samples)
        return self.dist.sample(shape).transpose(0, 1)

    def log_prob(self, x):
        return self.dist.log_prob(x)


def get_indep_gaussian_prior(sigma_init, rollout_steps, control_dim, mu_init=None, tensor_args=None):
    """priors/gaussian.py:63-82: Normal(mu, sigma_init) per (step, control dimension); tensor_args (extra) places it on a device."""
    import torch.distributions as dist
    ta = tensor_args or {}
    mu = torch.zeros(rollout_steps, control_dim, **ta)
    if mu_init is not None:
        mu[:, :] = torch.as_tensor(mu_init, **ta)
    return dist.Normal(mu, torch.ones(rollout_steps, control_dim, **ta) * sigma_init)


def avg_ctrl_to_goal(state, target, rollout_steps, dt, max_ctrl=100, control_type='velocity'):
    """gaussian.py:36-58 (host-side helper, velocity control)."""
    assert control_type in ['velocity', 'acceleration']
    if control_type == 'velocity':
        return ((target - state) / rollout_steps / dt).clamp(max=max_ctrl)
    pos_dim = int(state.dim() / 2)
    return ((target[:pos_dim] - state[:pos_dim]) / rollout_steps / dt ** 2).clamp(max=max_ctrl)


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


class ControlTrajectoryPrior(ABC):
    """Prior on control trajectories, independent across control dimensions (gaussian.py:218-298)."""

    def __init__(self, rollout_steps, ctrl_dim, tensor_args=None):
        self.rollout_steps = rollout_steps
        self.ctrl_dim = ctrl_dim
        self.device = require_cuda(tensor_args)
        self.tensor_args = dict(device=self.device, dtype=torch.float32)

    @abstractmethod
    def make_dist(self):
        """Build the per-dimension sampling constants."""

    def log_prob(self, samples, cond_inputs=None):
        raise NotImplementedError('log_prob is not served on the GPU (no planner of the reference calls it)')

    def update_means(self, means):
        """gaussian.py:270-273."""
        self.mu = means.detach().clone().to(**self.tensor_args).reshape(self.rollout_steps, self.ctrl_dim)
        self._mu64 = self.mu.t().double().contiguous()                    # (c, T)

    def sample(self, num_samples, cond_inputs=None, eps=None):
        """gaussian.py:276-298: (num_samples, rollout_steps, ctrl_dim) control samples, one multivariate normal per control
        dimension.  eps: optional standard normals (ctrl_dim, num_samples, rollout_steps) in the reference's draw order
        (parity runs); default: device Philox, a fresh stream every call."""
        n, T, c = int(num_samples), self.rollout_steps, self.ctrl_dim
        U_s = torch.empty(n, T, c, **self.tensor_args)
        for i in range(c):
            e = None if eps is None else eps[i].to(device=self.device, dtype=torch.float64).reshape(n, 1, T).contiguous()
            out = ops.mvn_sample_dense(self._mu64[i:i + 1].contiguous(), e, self._tril_t[i], n, seed=self.seed + self._calls * c + i)
            U_s[:, :, i] = out
        self._calls += 1
        return U_s


class ControlTrajectoryGaussian(ControlTrajectoryPrior):
    """Multivariate Gaussian per control dimension (gaussian.py:301-333): mu (T, c), Cov (T, T, c)."""

    def __init__(self, rollout_steps, ctrl_dim, mu=None, Cov=None, tensor_args=None, seed=0):
        assert mu.size(0) == rollout_steps
        assert mu.size(1) == ctrl_dim
        super().__init__(rollout_steps, ctrl_dim, tensor_args=tensor_args)
        self.seed, self._calls = int(seed), 0
        self.Cov = Cov.to(**self.tensor_args)
        self.update_means(mu)
        self.make_dist()

    def make_dist(self):
        """MultivariateNormal(covariance_matrix=C).scale_tril == cholesky(C) (host fp64 set-up), kept transposed for the
        sampling kernel."""
        C = self.Cov.detach().cpu().double()
        self._tril_t = [torch.linalg.cholesky(C[..., i]).t().contiguous().to(self.device) for i in range(self.ctrl_dim)]


def get_multivar_gaussian_prior(sigma, rollout_steps, control_dim, Cov_type='indep_ctrl', mu_init=None, tensor_args=None, seed=0):
    """gaussian.py:85-140."""
    assert Cov_type in ['indep_ctrl', 'const_ctrl'], 'Invalid type for control prior dist.'
    cpu = dict(device='cpu', dtype=torch.float32)
    mu = torch.zeros(rollout_steps, control_dim, **cpu)
    if mu_init is not None:
        mu[:, :] = mu_init.detach().cpu()
    Cov = (const_ctrl_Cov if Cov_type == 'const_ctrl' else diag_Cov)(sigma, rollout_steps, control_dim, tensor_args=cpu)
    return ControlTrajectoryGaussian(rollout_steps, control_dim, mu, Cov, tensor_args=tensor_args, seed=seed)
