"""Point-particle system of the MPPI examples with the reference's class surface (mp_baselines/planners/dynamics/point.py).

`dynamics(x, u)` and `traj_cost(X, U, ...)` are callable as in the reference and run on the GPU (mpb_point_dynamics,
mpb_point_traj_cost, csrc/mpb_mppi.hip); the MPPI planner itself fuses the same arithmetic into its one-launch loop
(mpb_mppi_step) and only reads this object's parameters."""
import numpy as np
import torch

from ... import ops
from ..base import require_cuda


class PointParticleDynamics:
    """Drop-in for mp_baselines.planners.dynamics.point.PointParticleDynamics (ctor kwargs point.py:6-23)."""

    def __init__(self, rollout_steps=None, control_dim=2, state_dim=2, dt=0.01, discount=1.0, deterministic=True,
                 start_state=None, goal_state=None, ctrl_min=None, ctrl_max=None, control_type='velocity',
                 dyn_std=np.zeros(4, ), c_weights=None, verbose=False, tensor_args=None):
        self.device = require_cuda(tensor_args)
        self.tensor_args = dict(device=self.device, dtype=torch.float32)
        self.control_dim = control_dim
        if control_type == 'velocity':
            self.state_dim = state_dim
        elif control_type == 'acceleration':
            # point.py:29-30 doubles the state here and then slices an empty tensor in dynamics(): the reference's acceleration
            # mode cannot run (include/mpb.h, mpb_mppi_step)
            raise IOError('control_type "acceleration" is not served (the reference\'s acceleration mode cannot run)')
        else:
            raise IOError('control_type "{}" not recognized'.format(control_type))
        self._c_weights = c_weights if c_weights is not None else {'pos': 10., 'vel': 10., 'ctrl': 0., 'pos_T': 10., 'vel_T': 0.}
        assert len(ctrl_min) == self.control_dim
        assert len(ctrl_max) == self.control_dim
        f = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(**self.tensor_args).contiguous()
        self.ctrl_min, self.ctrl_max = f(ctrl_min), f(ctrl_max)
        self.discount_seq = self._get_discount_seq(discount, rollout_steps)
        self.start_state = f(start_state) if start_state is not None else torch.zeros(self.state_dim, **self.tensor_args)
        self.state = self.start_state.clone()
        if goal_state is not None:
            if isinstance(goal_state, np.ndarray):
                self.goal_state = f(goal_state)
            elif isinstance(goal_state, torch.Tensor):
                self.goal_state = goal_state.to(**self.tensor_args)
            else:
                raise IOError
        else:
            self.goal_state = torch.zeros(self.state_dim, **self.tensor_args)
        self.rollout_steps = rollout_steps
        self.dt = dt
        self.dyn_std = f(dyn_std)
        self.control_type = control_type
        self.discount = discount
        self.verbose = verbose
        self.deterministic = deterministic

    @property
    def state(self):
        return self._state.clone().detach()

    @state.setter
    def state(self, value):
        self._state = value

    def reset(self):
        """point.py:83-89."""
        self.state = self.start_state.clone()
        cost = self.traj_cost(self.state.reshape(1, 1, 1, -1), torch.zeros(1, 1, 1, self.control_dim, **self.tensor_args))
        return self.state, cost

    def step(self, action):
        """point.py:91-100."""
        state = self.state.reshape(1, 1, -1)
        action = action.to(**self.tensor_args).reshape(1, 1, -1)
        self.state = self.dynamics(state, action)
        cost = self.traj_cost(state.reshape(1, 1, 1, -1), action.reshape(1, 1, 1, -1))
        return self.state.squeeze(), cost.squeeze()

    def dynamics(self, x, u_s, use_crn=False):
        """point.py:102-140: x (n_ctrl, n_state, state_dim), u_s (n_ctrl, n_state, ctrl_dim) -> x + clamp(u) * dt (+ noise in the
        control channel when not deterministic: dyn_std * randn, one draw per state sample with use_crn)."""
        x = x.to(**self.tensor_args)
        shape = torch.broadcast_shapes(x.shape, u_s.shape)
        xb = x.expand(shape).contiguous()
        ub = u_s.to(**self.tensor_args).expand(shape).contiguous()
        dim = shape[-1]
        if dim != self.control_dim or dim != self.state_dim:
            raise ValueError('dynamics: state and control rows must both have %d entries (velocity control)' % self.control_dim)
        noise = None
        if not self.deterministic:
            n_ctrl, n_state = x.size(0), x.size(1)
            noise = torch.randn(*((n_state, dim) if use_crn else (n_ctrl, n_state, dim)), **self.tensor_args)
            noise = noise.expand(shape).contiguous()
        return ops.point_dynamics(xb, ub, self.ctrl_min, self.ctrl_max, self.dt, dyn_std=self.dyn_std[:dim].contiguous(), noise=noise)

    def render(self, state=None, mode='human'):
        pass

    def _get_discount_seq(self, discount, rollout_steps):
        """point.py:145-152 (host-side set-up, the reference's own torch calls)."""
        seq = torch.cumprod(torch.ones(rollout_steps, dtype=torch.float32) * discount, dim=0)
        seq /= discount
        return seq.to(**self.tensor_args)

    def traj_cost(self, X_in, U_in, **observation):
        """point.py:154-226: X_in (steps, n_ctrl, n_state, state_dim), U_in (steps, n_ctrl, n_state, ctrl_dim) -> costs
        (n_ctrl, n_state).  observation: goal_state, cost (any cost object: its per-rollout costs collapse into ONE scalar
        added to every rollout, quirk Q6)."""
        T, n_ctrl, n_state, sd = X_in.shape
        B = n_ctrl * n_state
        X = X_in.to(**self.tensor_args).reshape(T, B, sd).contiguous()
        U = U_in.to(**self.tensor_args).reshape(T, B, self.control_dim).contiguous()
        goal = observation.get('goal_state', self.goal_state)
        goal = goal.to(**self.tensor_args).reshape(-1)[:self.state_dim].contiguous()
        cost = observation.get('cost', None)
        energy = 0.0
        if cost is not None:
            full_traj = torch.cat((X.transpose(0, 1), U.transpose(0, 1)), dim=-1)      # (B, T, sd + cd)
            energy = float(cost.eval(full_traj).sum(-1))
        disc = self.discount_seq if self.discount_seq.numel() == T else self.discount_seq[:T].contiguous()
        if disc.numel() != T:
            raise ValueError('traj_cost: %d steps but the discount sequence holds %d' % (T, disc.numel()))
        cw = self._c_weights
        out = ops.point_traj_cost(X, U, goal, disc, cw['pos'], cw['vel'], cw['ctrl'], cw['pos_T'], energy)
        if self.verbose:
            print('traj cost: mean {:5.4f}'.format(float(out.mean())))
        return out.view(n_ctrl, n_state)
