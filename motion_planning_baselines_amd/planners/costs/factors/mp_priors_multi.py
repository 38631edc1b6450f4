"""MultiMPPrior (mp_baselines/planners/costs/factors/mp_priors_multi.py:15-259): multi-modal constant-velocity GP
trajectory prior.

The reference materialises the dense M x M precision K^-1 = A^T Q^-1 A (M = state_dim * (num_steps+1)) in fp64 and
hands it to MultivariateNormal.  With isotropic factors (K_s_inv = I/s_s^2, K_gp_inv = Q^-1(sigma_gp), K_g_inv =
I/s_g^2 -- what every planner of the reference passes) the precision is block tridiagonal with (2x2) (x) I_dof
blocks; this class keeps that structured form (planners.base.gp_prior_factor), samples on the GPU
(mpb_gp_prior_sample / _dense) and builds the dense matrix only when `Sigma_inv` is actually read.
"""
import numpy as np
import torch

from .... import ops
from ...base import const_vel_mean, gp_prior_factor, gp_prior_scale_tril, require_cuda


class MultiMPPrior:

    def __init__(self, num_steps, dt, state_dim, dof, K_s_inv, K_gp_inv, start_state, means=None, K_g_inv=None,
                 goal_states=None, use_numpy=False, tensor_args=None, noise='torch_cpu', seed=0):
        self.state_dim, self.dof, self.num_steps, self.dt = state_dim, dof, num_steps, dt
        assert state_dim == 2 * dof
        self.M = state_dim * (num_steps + 1)
        self.tensor_args = tensor_args
        self.device = require_cuda(dict(tensor_args, dtype=torch.float32))
        self.goal_directed = goal_states is not None
        self.noise, self.seed = noise, int(seed)
        H = num_steps + 1
        # isotropic factors only: recover the sigmas and check the structure
        Ks = torch.as_tensor(K_s_inv).detach().cpu().double()
        Kgp = torch.as_tensor(K_gp_inv).detach().cpu().double()
        self.sigma_start = float(Ks[0, 0]) ** -0.5
        self.sigma_gp = (float(Kgp[0, 0]) * dt ** 3 / 12.0) ** -0.5
        I = torch.eye(dof, dtype=torch.float64)
        qa, qb, qc = 12.0 / dt ** 3, -6.0 / dt ** 2, 4.0 / dt
        Qi = torch.cat((torch.cat((qa * I, qb * I), 1), torch.cat((qb * I, qc * I), 1)), 0) / self.sigma_gp ** 2
        ok = torch.allclose(Ks, torch.eye(state_dim, dtype=torch.float64) / self.sigma_start ** 2, rtol=1e-6) and \
            torch.allclose(Kgp, Qi, rtol=1e-6, atol=1e-12 * float(Qi.abs().max()))
        self.sigma_goal = None
        if self.goal_directed:
            Kg = torch.as_tensor(K_g_inv).detach().cpu().double()
            self.sigma_goal = float(Kg[0, 0]) ** -0.5
            ok = ok and torch.allclose(Kg, torch.eye(state_dim, dtype=torch.float64) / self.sigma_goal ** 2, rtol=1e-6)
        self._general = not ok
        self._K = (Ks, Kgp, torch.as_tensor(K_g_inv).detach().cpu().double() if self.goal_directed else None)
        if means is None:
            self.num_modes = goal_states.shape[0] if self.goal_directed else 1
            s = torch.as_tensor(start_state).detach().cpu().double()
            if self.goal_directed:
                g = torch.as_tensor(goal_states).detach().cpu().double()
                means = torch.stack([const_vel_mean(s[:dof], g[i, :dof], H, dt) for i in range(self.num_modes)])
            else:
                means = s.repeat(H, 1).unsqueeze(0)                     # mp_priors_multi.py:176
        else:
            self.num_modes = means.shape[0]
        self.means = torch.as_tensor(means).detach().reshape(self.num_modes, -1).to(self.device, torch.float64)
        self._Sigma_inv = None
        self._modal = None               # (num_modes, M, M): one precision per mode, after set_Sigma_invs / update_dist
        self._U_d = None                 # the structured factor of one degree of freedom on the device (log_prob)
        self._draws = 0
        if self._general:
            # arbitrary precisions (mp_priors_multi.py:213-251 takes any matrices): no (2x2) (x) I structure -- the dense
            # K^-1 = A^T Q^-1 A and its scale_tril (as MultivariateNormal(precision_matrix=...) derives it,
            # multivariate_normal.py:80-86) are built once on the host in fp64, sampling is a dense product on the GPU
            Kinv = self._dense_precision(dt)
            self._Sigma_inv = torch.from_numpy(Kinv).to(**self.tensor_args)
            self._set_dense_factors([Kinv])
            return
        self._Ud, self._Uo = gp_prior_factor(H, dt, self.sigma_start, self.sigma_gp, self.sigma_goal)
        f64 = lambda a: torch.as_tensor(a, dtype=torch.float64).to(self.device).contiguous()
        self._Ud_d, self._Uo_d = f64(self._Ud), f64(self._Uo)
        self._tril = f64(gp_prior_scale_tril(self._Ud, self._Uo)) if H <= 128 else None

    def _dense_precision(self, dt, K=None):
        """K^-1 = A^T Q^-1 A (mp_priors_multi.py:213-251) for arbitrary K_s_inv / K_gp_inv / K_g_inv, block by block in fp64:
        diag block t: [t=0] K_s + [t<H-1] Phi^T Q Phi + [t>0] Q + [t=H-1] K_g; block (t, t+1): -Phi^T Q."""
        H, sd, D = self.num_steps + 1, self.state_dim, self.dof
        Ks, Q, Kg = (None if k is None else k.numpy() for k in (self._K if K is None else K))
        Phi = np.eye(sd)
        Phi[:D, D:] = np.eye(D) * dt
        PQP, off = Phi.T @ Q @ Phi, -Phi.T @ Q
        K = np.zeros((self.M, self.M))
        for t in range(H):
            blk = np.zeros((sd, sd))
            if t == 0:
                blk += Ks
            if t < H - 1:
                blk += PQP
            if t > 0:
                blk += Q
            if t == H - 1 and Kg is not None:
                blk += Kg
            K[t * sd:(t + 1) * sd, t * sd:(t + 1) * sd] = blk
            if t < H - 1:
                K[t * sd:(t + 1) * sd, (t + 1) * sd:(t + 2) * sd] = off
                K[(t + 1) * sd:(t + 2) * sd, t * sd:(t + 1) * sd] = off.T
        return K

    def _set_dense_factors(self, Kinvs):
        """Dense path: per distinct precision matrix (one shared by all modes, or one per mode after set_Sigma_invs) the
        scale_tril as MultivariateNormal(precision_matrix=...) derives it (multivariate_normal.py:80-86), transposed for
        mpb_mvn_sample_dense, the precision itself on the device for log_prob, and log det K^-1 -- all fp64, built once on the host."""
        from scipy.linalg import solve_triangular
        self._tril_t, self._Kinv_d, self._logdet = [], [], []
        for Kinv in Kinvs:
            Kinv = np.asarray(Kinv, dtype=np.float64)
            Lf = np.linalg.cholesky(Kinv[::-1, ::-1])
            L_inv = np.ascontiguousarray(Lf[::-1, ::-1].T)
            tril = solve_triangular(L_inv, np.eye(self.M), lower=True)
            self._tril_t.append(torch.from_numpy(np.ascontiguousarray(tril.T)).to(self.device))
            self._Kinv_d.append(torch.from_numpy(np.ascontiguousarray(Kinv)).to(self.device))
            self._logdet.append(-2.0 * float(np.log(np.diag(tril)).sum()))          # log det K^-1 = -2 sum log diag(scale_tril)

    # ---- the rest of the reference's public surface (mp_priors_multi.py:100-176, :213-259) -----------
    def update_dist(self, means, Sigma_invs):
        """mp_priors_multi.py:100-110: (re)define the distribution -- means (num_modes, M), one precision per mode.  The
        reference builds a dense MultivariateNormal; here the factors of the dense path are rebuilt (host fp64) unless the
        precisions are the ones this object already holds."""
        self.means = torch.as_tensor(means).detach().reshape(self.num_modes, -1).to(self.device, torch.float64)
        S = torch.as_tensor(Sigma_invs).detach().cpu().double()
        assert tuple(S.shape) == (self.num_modes, self.M, self.M)
        cur = self.Sigma_inv.detach().cpu().double()
        if self._modal is None and all(torch.equal(S[i], cur) for i in range(self.num_modes)):
            return
        same = all(torch.equal(S[i], S[0]) for i in range(1, self.num_modes))
        self._general = True
        self._modal = None if same else S.to(**self.tensor_args)
        self._Sigma_inv = S[0].to(**self.tensor_args)
        self._set_dense_factors([S[0].numpy()] if same else [S[i].numpy() for i in range(self.num_modes)])

    def set_Sigma_invs(self, Sigma_invs_new):
        """mp_priors_multi.py:124-128."""
        assert tuple(Sigma_invs_new.shape) == tuple(self.Sigma_invs.shape)
        self.update_dist(self.means, Sigma_invs_new)

    @classmethod
    def const_vel_trajectory(cls, start_state, goal_state, dt, num_steps, dof, set_initial_final_vel_to_zero=True,
                             tensor_args=None):
        """mp_priors_multi.py:130-152: straight line start -> goal with the mean velocity on the interior points (and on the
        end points too unless set_initial_final_vel_to_zero)."""
        ta = tensor_args or dict(device=start_state.device, dtype=start_state.dtype)
        a = (torch.arange(num_steps + 1, **ta) / num_steps).unsqueeze(1)
        s, g = start_state[:dof].to(**ta), goal_state[:dof].to(**ta)
        traj = torch.zeros(num_steps + 1, 2 * dof, **ta)
        traj[:, :dof] = s * (1.0 - a) + g * a
        vel = ((g - s) / (num_steps * dt)).unsqueeze(0)
        if set_initial_final_vel_to_zero:
            traj[1:-1, dof:] = vel
        else:
            traj[:, dof:] = vel
        return traj

    def get_const_vel_mean(self, start_state, goal_states, dt, num_steps, dof):
        """mp_priors_multi.py:154-176: (num_modes, num_steps + 1, 2 dof) straight-line means, one per goal; without goals the
        start state repeated."""
        if self.goal_directed:
            return torch.stack([self.const_vel_trajectory(start_state, goal_states[i], dt, num_steps, dof, tensor_args=self.tensor_args)
                                for i in range(self.num_modes)], dim=0)
        return start_state.repeat(num_steps + 1, 1)

    def get_const_vel_covariance(self, dt, K_s_inv, K_gp_inv, K_g_inv, precision_matrix=True):
        """mp_priors_multi.py:213-251: K^-1 = A^T Q^-1 A (or its inverse) from the factor precisions, in fp64, returned in
        the object's dtype."""
        K = tuple(None if k is None else torch.as_tensor(k).detach().cpu().double()
                  for k in (K_s_inv, K_gp_inv, K_g_inv if self.goal_directed else None))
        Kinv = self._dense_precision(dt, K)
        out = Kinv if precision_matrix else np.linalg.inv(Kinv)
        return torch.from_numpy(np.ascontiguousarray(out)).to(**self.tensor_args)

    def log_prob(self, x):
        """mp_priors_multi.py:258-259 (MultivariateNormal.log_prob): x (..., num_modes, M) -> (..., num_modes),
        -1/2 e^T K^-1 e + 1/2 log det K^-1 - M/2 log 2 pi with e = x - mean, in fp64 on the device.  Isotropic factors: through
        the structured factor K^-1 = U U^T per degree of freedom (U upper triangular, 2H x 2H, band width 4) -- e^T K^-1 e =
        sum_dof |U^T e_dof|^2, log det K^-1 = 2 dof sum log diag U; the dense M x M matrix is never formed.  Arbitrary
        precisions: the dense quadratic form."""
        H, D, sd = self.num_steps + 1, self.dof, self.state_dim
        x = torch.as_tensor(x).to(self.device, torch.float64)
        e = x.reshape(*x.shape[:-1], self.M) - self.means                                   # broadcast over the leading dims
        const = -0.5 * self.M * float(np.log(2.0 * np.pi))
        if self._general:
            outs = []
            for i in range(self.num_modes):
                j = i if self._modal is not None else 0
                ei = e[..., i, :]
                outs.append(-0.5 * ((ei @ self._Kinv_d[j]) * ei).sum(-1) + 0.5 * self._logdet[j] + const)
            out = torch.stack(outs, dim=-1)
        else:
            if self._U_d is None:
                U = np.zeros((2 * H, 2 * H))
                for t in range(H):
                    U[2 * t, 2 * t], U[2 * t, 2 * t + 1], U[2 * t + 1, 2 * t + 1] = self._Ud[t]
                    if t < H - 1:
                        U[2 * t:2 * t + 2, 2 * t + 2:2 * t + 4] = self._Uo[t].reshape(2, 2)
                self._U_d = torch.from_numpy(U).to(self.device)
                self._logdet_s = 2.0 * D * float(np.log(np.abs(np.diag(U))).sum())
            # e (..., modes, H, {pos, vel}, dof) -> per dof the 2H-vector [pos_0, vel_0, pos_1, vel_1, ...]
            ed = e.reshape(*e.shape[:-1], H, 2, D).movedim(-1, -3).reshape(*e.shape[:-1], D, 2 * H)
            y = ed @ self._U_d                                                              # (U^T e)^T, row vectors
            out = -0.5 * (y * y).sum((-1, -2)) + 0.5 * self._logdet_s + const
        return out.to(self.tensor_args.get('dtype', torch.float32))

    # ---- dense views (built on demand, fp64 on the host like the reference) --------------------------
    @property
    def Sigma_inv(self):
        """Dense precision K^-1 (M x M), index t*state_dim + c (mp_priors_multi.py:213-251)."""
        if self._Sigma_inv is None:
            H, D = self.num_steps + 1, self.dof
            U = np.zeros((2 * H, 2 * H))
            for t in range(H):
                U[2 * t, 2 * t], U[2 * t, 2 * t + 1], U[2 * t + 1, 2 * t + 1] = self._Ud[t]
                if t < H - 1:
                    U[2 * t:2 * t + 2, 2 * t + 2:2 * t + 4] = self._Uo[t].reshape(2, 2)
            P1 = U @ U.T                                                # one degree of freedom, index 2t + {pos, vel}
            K = np.zeros((self.M, self.M))
            for d in range(D):
                idx = np.array([[t * 2 * D + d, t * 2 * D + D + d] for t in range(H)]).reshape(-1)
                K[np.ix_(idx, idx)] = P1
            self._Sigma_inv = torch.from_numpy(K).to(**self.tensor_args)
        return self._Sigma_inv

    @property
    def Sigma_invs(self):
        return self._modal if self._modal is not None else self.Sigma_inv.repeat(self.num_modes, 1, 1)

    def get_mean(self, reshape=True):
        m = self.means.clone().detach()
        return m.reshape(self.num_modes, self.num_steps + 1, self.state_dim) if reshape else m

    def set_mean(self, means_new):
        assert means_new.shape == self.means.shape
        self.means = means_new.clone().detach().to(self.device, torch.float64)

    def sample(self, num_samples):
        """(num_modes, num_samples, H, state_dim) like the reference's `.transpose(1, 0)` view (:253-256)."""
        H, dim = self.num_steps + 1, self.state_dim
        eps = None
        if self.noise != 'philox':
            eps = torch.empty(num_samples, self.num_modes, H * dim, dtype=torch.float64).normal_().to(self.device)
        if self._general and self._modal is None:
            out = ops.mvn_sample_dense(self.means.contiguous(), eps, self._tril_t[0], num_samples, seed=self.seed + self._draws)
            self._draws += 1
            return out.reshape(self.num_modes, num_samples, H, dim).to(self.tensor_args.get('dtype', torch.float32))
        if self._general:                # one precision per mode: one dense product per mode
            outs = []
            for i in range(self.num_modes):
                ei = None if eps is None else eps[:, i:i + 1].contiguous()
                outs.append(ops.mvn_sample_dense(self.means[i:i + 1].contiguous(), ei, self._tril_t[i], num_samples,
                                                 seed=self.seed + self._draws + 7919 * i))
            self._draws += 1
            return torch.stack(outs).reshape(self.num_modes, num_samples, H, dim).to(self.tensor_args.get('dtype', torch.float32))
        out = ops.gp_prior_sample(self.means.reshape(self.num_modes, H, dim).contiguous(), eps, self._Ud_d, self._Uo_d,
                                  num_samples, self.dof, seed=self.seed + self._draws, scale_tril=self._tril)
        self._draws += 1
        return out.reshape(self.num_modes, num_samples, H, dim).to(self.tensor_args.get('dtype', torch.float32))
