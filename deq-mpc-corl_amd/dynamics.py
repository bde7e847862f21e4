"""Dynamics providers that run as HIP kernels (no PyTorch round trip between Newton steps).

`Pendulum1lDynamics` mirrors the reference's `my_envs.dynamics.Dynamics` / `CartpoleDynamics`
family for the one-link pendulum package (deqmpc/my_envs/dynamics.py:15-75,
my_envs/pendulum1l/src/dynamics.cpp:13-47): state x = (theta, omega), action u = tau, one RK4
step of length dt; `jac` returns the next state and (A, B) like the `dx_jac` callables the MPC
takes (al_utils.py:237-248). fp64 or fp32 on a ROCm device only: no CPU fallback.
"""
import torch

from .backend import default_backend


class Pendulum1lDynamics:
    nx, nu = 2, 1
    fused_id = 1   # model id of the nonlinear fused solve (alqp_solve_nonlin)

    def __init__(self, dt=0.05, backend=None):
        self.dt = float(dt)
        self.backend = backend

    def _be(self):
        if self.backend is None:
            self.backend = default_backend()
        return self.backend

    def __call__(self, x, u):
        xn, _ = self._be().dyn_pendulum1l(x.contiguous(), u.contiguous(), self.dt, want_jac=False)
        return xn

    def jac(self, x, u):
        xn, F = self._be().dyn_pendulum1l(x.contiguous(), u.contiguous(), self.dt, want_jac=True)
        return xn, (F[..., :2], F[..., 2:])


class pendulum1l:
    """Module-shaped twin of the reference's compiled `pendulum1l` package
    (my_envs/pendulum1l/src/dynamics.cpp:49-53: `dynamics`, `derivatives`), for
    `Dynamics.package` (my_envs/dynamics.py:60-75): same argument order, shapes and return lists."""

    @staticmethod
    def dynamics(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        xn, _ = be.dyn_pendulum1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=False)
        return [xn[:, :1].contiguous(), xn[:, 1:].contiguous()]

    @staticmethod
    def derivatives(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        _, F = be.dyn_pendulum1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=True)
        # (dq'/dq, dq'/dqdot, dq'/dtau, dqdot'/dq, dqdot'/dqdot, dqdot'/dtau), each [bsz, 1, 1]
        return [F[:, i, j].reshape(-1, 1, 1).contiguous() for i in (0, 1) for j in (0, 1, 2)]


class Cartpole1lDynamics:
    """One-link cartpole (my_envs/cartpole.py:27-38 `CartpoleDynamics(nx=4)`): x = (cart position,
    pole angle, their rates), pole angle 0 = upright; the action drives the cart only
    (my_envs/dynamics.py:54-56), so tau = (u, 0)."""
    nx, nu = 4, 1
    fused_id = 2

    def __init__(self, dt=0.05, backend=None):
        self.dt = float(dt)
        self.backend = backend

    def _be(self):
        if self.backend is None:
            self.backend = default_backend()
        return self.backend

    @staticmethod
    def _tau(u):
        return torch.cat((u, torch.zeros_like(u)), 1).contiguous()

    def __call__(self, x, u):
        xn, _ = self._be().dyn_cartpole1l(x.contiguous(), self._tau(u), self.dt, want_jac=False)
        return xn

    def jac(self, x, u):
        xn, J = self._be().dyn_cartpole1l(x.contiguous(), self._tau(u), self.dt, want_jac=True)
        return xn, (J[..., :4], J[..., 4:5])


class cartpole1l:
    """Module-shaped twin of the reference's compiled `cartpole1l` package (same conventions as
    `pendulum1l` above; the six Jacobian blocks are [bsz, 2, 2])."""

    @staticmethod
    def dynamics(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        xn, _ = be.dyn_cartpole1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=False)
        return [xn[:, :2].contiguous(), xn[:, 2:].contiguous()]

    @staticmethod
    def derivatives(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        _, J = be.dyn_cartpole1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=True)
        return [J[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2].contiguous() for i in (0, 1) for j in (0, 1, 2)]


class Cartpole1lV2Dynamics(Cartpole1lDynamics):
    """The reference's second one-link cartpole package (`my_envs/cartpole1l_v2`: cart 0.5 kg, pole 0.2 kg at 0.5 m;
    shipped but not imported by `my_envs/cartpole.py:35`). Same state and action layout; also compiled into the
    nonlinear fused solve (model id 4)."""
    fused_id = 4

    def __call__(self, x, u):
        xn, _ = self._be().dyn_cartpole1l(x.contiguous(), self._tau(u), self.dt, want_jac=False, version=2)
        return xn

    def jac(self, x, u):
        xn, J = self._be().dyn_cartpole1l(x.contiguous(), self._tau(u), self.dt, want_jac=True, version=2)
        return xn, (J[..., :4], J[..., 4:5])


class cartpole1l_v2:
    """Module-shaped twin of the reference's compiled `cartpole1l_v2` package (conventions of `cartpole1l`)."""

    @staticmethod
    def dynamics(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        xn, _ = be.dyn_cartpole1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=False, version=2)
        return [xn[:, :2].contiguous(), xn[:, 2:].contiguous()]

    @staticmethod
    def derivatives(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        _, J = be.dyn_cartpole1l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=True, version=2)
        return [J[:, 2 * i:2 * i + 2, 2 * j:2 * j + 2].contiguous() for i in (0, 1) for j in (0, 1, 2)]


class Cartpole2lDynamics:
    """Two-link cartpole (`CartpoleDynamics(nx=6)`, my_envs/cartpole.py:30-32): x = (cart position,
    th1, th2 relative to link 1, their rates), both angles 0 = upright; tau = (u, 0, 0)."""
    nx, nu = 6, 1
    fused_id = 3
    # The model is compiled into the nonlinear fused solve as well; for this heavier model at B = 4096
    # it only ties with the launch-per-phase route (3.2 ms per call both; the line search costs five
    # full model evaluations per lane and stage), so the MPC only fuses on request
    fused_default = False

    def __init__(self, dt=0.05, backend=None):
        self.dt = float(dt)
        self.backend = backend

    def _be(self):
        if self.backend is None:
            self.backend = default_backend()
        return self.backend

    @staticmethod
    def _tau(u):
        return torch.cat((u, torch.zeros_like(u), torch.zeros_like(u)), 1).contiguous()

    def __call__(self, x, u):
        xn, _ = self._be().dyn_cartpole2l(x.contiguous(), self._tau(u), self.dt, want_jac=False)
        return xn

    def jac(self, x, u):
        xn, J = self._be().dyn_cartpole2l(x.contiguous(), self._tau(u), self.dt, want_jac=True)
        return xn, (J[..., :6], J[..., 6:7])


class cartpole2l:
    """Module-shaped twin of the reference's compiled `cartpole2l` package (six [bsz, 3, 3] blocks)."""

    @staticmethod
    def dynamics(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        xn, _ = be.dyn_cartpole2l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=False)
        return [xn[:, :3].contiguous(), xn[:, 3:].contiguous()]

    @staticmethod
    def derivatives(q_in, qdot_in, tau_in, h_in):
        be = default_backend()
        _, J = be.dyn_cartpole2l(torch.cat((q_in, qdot_in), 1).contiguous(), tau_in.contiguous(), h_in, want_jac=True)
        return [J[:, 3 * i:3 * i + 3, 3 * j:3 * j + 3].contiguous() for i in (0, 1) for j in (0, 1, 2)]


class _RigidDynamics:
    """Shared plumbing of the two torch-coded robots of the reference: `__call__(x, u)` is the
    reference's `dynamics` module (one RK4 step), `jac(x, u)` its `dynamics_derivatives` in the `dx_jac` convention the
    MPC takes - `(x_next, (A, B))` (al_utils.py:237-248) - from ONE kernel launch instead of nx replicated forward +
    backward passes of a TorchScript RK4 (rex_quadrotor.py:136-144).

    PARITY UNPINNED: the reference modules import `rexquad_utils`, which is not in the reference tree; mrp2quat, quatrot and
    w2pdotkinematics_mrp are restated from their standard definitions (csrc/alqp_dyn_rigid.hip). Constants the reference
    keeps in float32 tensors (J, inv(J), gravity, arm directions, Bf, u_hover) are rounded through float32 as it does."""
    nu = 4
    _model = None

    def _fill(self, mass, J, gravity, motor_dist, kf, bf, km, act_scale, u_hover, pend_L, bf_force):
        import numpy as np
        from . import _lib
        J32 = np.asarray(J, dtype=np.float32)
        if J32.ndim == 1:
            J32 = np.diag(J32)
        Jinv32 = np.linalg.inv(J32).astype(np.float32)
        ss = np.array([[1., 1, 0], [1., -1, 0], [-1., -1, 0], [-1., 1, 0]], dtype=np.float32)
        ss = (ss / np.linalg.norm(ss, axis=-1, keepdims=True).astype(np.float32)).astype(np.float64)
        p = _lib.AlqpRigidParams()
        p.mass = float(mass)
        for i in range(9):
            p.J[i] = float(J32.reshape(-1)[i])
            p.Jinv[i] = float(Jinv32.reshape(-1)[i])
        for i in range(3):
            p.g[i] = float(np.float32(gravity[i]))
        p.motor_dist, p.kf, p.bf, p.km = float(motor_dist), float(kf), float(bf), float(km)
        p.act_scale, p.u_hover, p.pend_L, p.bf_force = float(act_scale), float(u_hover), float(pend_L), float(bf_force)
        for i in range(12):
            p.ss[i] = float(ss.reshape(-1)[i])
        self.params = p

    def _be(self):
        if self.backend is None:
            self.backend = default_backend()
        return self.backend

    def __call__(self, x, u):
        xn, _ = self._be().dyn_rigid(self._model, self.params, x.contiguous(), u.contiguous(), self.dt, want_jac=False)
        return xn

    def jac(self, x, u):
        xn, F = self._be().dyn_rigid(self._model, self.params, x.contiguous(), u.contiguous(), self.dt, want_jac=True)
        return xn, (F[..., :self.nx], F[..., self.nx:])


class RexQuadrotorDynamics(_RigidDynamics):
    """`RexQuadrotor_dynamics` (deqmpc/rex_quadrotor.py:8-127; same constructor defaults): x = (position r, attitude as
    modified Rodrigues parameters, body-frame velocity v, body rate w), u = the four rotor commands (scaled by 100)."""
    nx = 12
    _model = "rex"

    def __init__(self, mass=2.0, J=((0.01566089, 0.00000318037, 0.0), (0.00000318037, 0.01562078, 0.0), (0.0, 0.0, 0.02226868)),
                 gravity=(0, 0, -9.81), motor_dist=0.28, kf=0.0244101, bf=-30.48576, km=0.00029958, dt=0.05, backend=None):
        import numpy as np
        self.dt, self.backend = float(dt), backend
        self._fill(mass, J, gravity, motor_dist, kf, bf, km, 100.0, 0.0, 1.0, float(np.float32(4 * bf)))
        self.u_hover = (-mass * gravity[2] - bf * 4) / 100.0 / kf / 4      # rex_quadrotor.py:44


class FlyingCartpoleDynamics(_RigidDynamics):
    """`FlyingCartpole_dynamics` as `FlyingCartpole` builds it (deqmpc/flying_cartpole2d.py:11-152): the quadrotor plus an
    inverted pendulum, x = (r, MRP, theta, v, w, theta'), u = rotor commands around hover (u <- 10 (u + u_hover))."""
    nx = 14
    _model = "flycart"

    def __init__(self, mass_q=2.0, mass_p=0.1, J=((0.0023, 0.0, 0.0), (0.0, 0.0023, 0.0), (0.0, 0.0, 0.004)), L=0.5,
                 gravity=(0, 0, -9.81), motor_dist=0.175, kf=1.0, bf=0.0, km=0.025, dt=0.05, backend=None):
        import numpy as np
        self.dt, self.backend = float(dt), backend
        m = mass_q + mass_p
        self.u_hover = float(np.float32((-m * gravity[2]) / 10.0 / kf / 4))   # flying_cartpole2d.py:42
        self._fill(m, J, gravity, motor_dist, kf, bf, km, 10.0, self.u_hover, float(np.float32(L)), 0.0)
