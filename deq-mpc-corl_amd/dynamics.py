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
