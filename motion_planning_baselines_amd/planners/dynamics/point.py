"""Point-particle system of the MPPI examples (mp_baselines/planners/dynamics/point.py:5-74): the parameters live here,
the dynamics and the trajectory cost run inside the MPPI kernel (csrc/mpb_mppi.hip)."""
import torch


class PointParticleDynamics:
    """Parameters of the reference's point-particle system (dynamics/point.py:5-74); the dynamics and
    trajectory cost themselves run inside the MPPI kernel."""

    def __init__(self, rollout_steps=None, control_dim=2, state_dim=2, dt=0.01, discount=1.0, goal_state=None,
                 ctrl_min=None, ctrl_max=None, control_type='velocity', c_weights=None, tensor_args=None, **kwargs):
        if control_type != 'velocity':
            raise IOError('only control_type "velocity" is served (the reference\'s acceleration mode cannot run)')
        self.control_dim = control_dim
        self.state_dim = state_dim
        self.dt = dt
        self.rollout_steps = rollout_steps
        self.tensor_args = tensor_args
        self._c_weights = c_weights or {'pos': 10., 'vel': 10., 'ctrl': 0., 'pos_T': 10., 'vel_T': 0.}
        assert len(ctrl_min) == control_dim and len(ctrl_max) == control_dim
        self.ctrl_min, self.ctrl_max = list(ctrl_min), list(ctrl_max)
        self.goal_state = goal_state
        seq = torch.cumprod(torch.ones(rollout_steps) * discount, dim=0) / discount   # point.py:145-152
        self.discount_seq = seq
