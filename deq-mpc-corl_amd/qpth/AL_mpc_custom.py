"""Drop-in for the reference's ``qpth.AL_mpc_custom`` module: ``Obstacle_MPC``.

The AL MPC of ``qpth.AL_mpc`` plus, at every stage, inequality rows that keep the position
``x_t[0:3]`` outside the ``n_obstacle_constraints`` (4) nearest of ``n_obstacles`` (40) spheres
(qpth/AL_mpc_custom.py:22-135; residual / Jacobian: qpth/al_utils.py:313-323, 351-388):

    c_k(x_t) = radius^2 - |x_t[0:3] - o_k|^2 <= 0

Which spheres are "nearest" is decided once per rollout from the reference trajectory handed to
``reinitialize`` (:104-109) and shifted by one stage in ``warm_start_initialize`` (:121-127) - host logic,
mirrored here. The arithmetic (the rank-<=4 Gauss-Newton update of the position corner of H_tt, the
state-dependent active set, the merit and dual-update terms) is in the HIP kernels behind
``alqp_newton_step_obs`` / ``alqp_merit_obs`` / ``alqp_dual_update_obs`` (include/mi_alqp.h). The reference
only reaches this class with PyTorch-coded dynamics (``FlyingCartpole_obstacles``, policies.py:1181), so the
solve runs in the nonlinear-caller mode: ``dx`` / ``dx_jac`` between launches.
"""
from __future__ import annotations

import torch

from deq_mpc_corl_amd.qpth.AL_mpc import MPC


class Obstacle_MPC(MPC):
    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_init=None, x_init=None, al_iter=2,
                 verbose=0, eps=1e-7, back_eps=1e-7, n_batch=None, linesearch_decay=0.2, max_linesearch_iter=10,
                 exit_unconverged=True, detach_unconverged=True, backprop=True, slew_rate_penalty=None,
                 solver_type="dense", add_goal_constraint=False, x_goal=None, diag_cost=True, ineqG=None,
                 ineqh=None, state_estimator=False, dtype=torch.float64, env=None, **kw):
        super().__init__(n_state, n_ctrl, T, u_lower, u_upper, u_init, x_init, al_iter, verbose, eps, back_eps,
                         n_batch, linesearch_decay, max_linesearch_iter, exit_unconverged, detach_unconverged,
                         backprop, slew_rate_penalty, solver_type, add_goal_constraint, x_goal, diag_cost, ineqG,
                         ineqh, state_estimator, dtype, **kw)
        if n_state < 3:
            raise ValueError("Obstacle_MPC: the obstacle rows act on the position x[0:3] (al_utils.py:316)")
        self.n_obstacles = 40                   # AL_mpc_custom.py:52
        self.n_obstacle_constraints = 4         # :53
        self.nineq += self.n_obstacle_constraints * T
        self.obstacle_radius = 0.2 if env is None else env.obstacle_radius
        self.obstacle_radius = torch.as_tensor(self.obstacle_radius).to(self.u_upper)
        self.obstacle_positions = None if env is None else env.obstacle_positions
        self.obstacles = None
        if n_batch is not None:
            self.lamda_prev = torch.zeros(n_batch, self.neq + self.nineq, dtype=dtype, device=self.u_upper.device)

    def _nearest(self, x):
        """The n_obstacle_constraints nearest centres to every x[..., :3]: [B, T', 4, 3] (:106-108)."""
        pos = self.obstacle_positions.to(x.device)
        dist = (pos[None, None] - x[:, :, None, :3]).norm(dim=-1)
        ids = torch.argsort(dist, dim=-1)[..., :self.n_obstacle_constraints]
        return pos[ids]

    def reinitialize(self, x, mask):
        super().reinitialize(x, mask)
        self.obstacles = (self._nearest(x), self.obstacle_radius)

    def warm_start_initialize(self, x, u, args):
        super().warm_start_initialize(x, u, args)
        # drop the first stage's spheres, pick new ones for the new last stage (:121-127)
        last = self._nearest(x[:, -1:])
        self.obstacles = (torch.cat([self.obstacles[0][:, 1:], last.to(self.obstacles[0])], dim=1), self.obstacle_radius)

    def _has_extra_rows(self):
        return True

    def _obs_kwargs(self, dtype, device):
        if self.obstacles is None:
            # (the reference dies with AttributeError here: its nearest-sphere table is built by reinitialize, :104-109.
            #  Passing "no obstacles" on would select the plain kernels, whose multiplier stride is M, on a [B, M + 4T] lamda)
            if self.obstacle_positions is None:
                raise RuntimeError("Obstacle_MPC: no obstacle positions (pass env=... with obstacle_positions)")
            raise RuntimeError("Obstacle_MPC: call reinitialize() before the first solve (it selects the nearest spheres)")
        pos, radius = self.obstacles
        return {"obs": (pos.detach().to(device=device, dtype=dtype).contiguous(), float(radius))}
