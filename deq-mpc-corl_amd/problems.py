"""Synthetic batched MPC/QP problems and dynamics callables.

The reference ships no problem data (its env modules import files that are not
in the tree), so the benchmark and the parity tests use the
survey-defined synthetic tracking problems: time-varying affine dynamics
``x_{t+1} = A_t x_t + B_t u_t + c_t`` with a diagonal tracking cost shaped like
``rex_quadrotor.py:164-165`` (Q = 10 on states, R = 1e-8 on controls) and box
bounds on the controls.

Everything here is plain torch on the CPU generator; tensors are moved to the
requested device afterwards, so the same seeds give the same problems here, in
``tools/gen_golden.py`` (which feeds them to the reference) and on the GPU box.
"""
from __future__ import annotations

from collections import namedtuple

import torch

Problem = namedtuple(
    "Problem", "B T nx nu Qd q F c x0 u_lo u_hi z0 xref"
)

_CHUNK = 4096


def _chunk(Bc, T, nx, nu, seed, active):
    n = nx + nu
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    f64 = torch.float64

    def randn(*shape):
        return torch.randn(*shape, generator=g, dtype=f64)

    A = torch.eye(nx, dtype=f64) + 0.05 * randn(Bc, T - 1, nx, nx)
    Bm = (0.5 if active else 0.1) * randn(Bc, T - 1, nx, nu)
    c = 0.01 * randn(Bc, T - 1, nx)
    x0 = randn(Bc, nx)
    xref = randn(Bc, T, n)
    xref[..., nx:] = 0.0
    F = torch.cat([A, Bm], dim=-1).contiguous()
    return F, c, x0, xref


def synthetic_problem(B, T, nx, nu, seed=0, dtype=torch.float64, active=False,
                      device="cpu"):
    """Seeded synthetic problem. Drawn in fp64 then cast.

    ``active=True`` is the branch-coverage variant (strong control authority,
    tight bounds) in which many controls sit on a bound.
    Batches above 4096 are generated per 4096-chunk with seed + chunk index.
    """
    n = nx + nu
    Fs, cs, x0s, xrefs = [], [], [], []
    done = 0
    k = 0
    while done < B:
        Bc = min(_CHUNK, B - done)
        F, c, x0, xref = _chunk(Bc, T, nx, nu, seed + k, active)
        Fs.append(F.to(dtype)); cs.append(c.to(dtype))
        x0s.append(x0.to(dtype)); xrefs.append(xref.to(dtype))
        done += Bc
        k += 1
    F = torch.cat(Fs).to(device)
    c = torch.cat(cs).to(device)
    x0 = torch.cat(x0s).to(device)
    xref = torch.cat(xrefs).to(device)
    qd = torch.tensor([10.0] * nx + [1e-8] * nu, dtype=dtype, device=device)
    Qd = qd.expand(B, T, n).contiguous()
    q = -(Qd * xref)
    b = 0.1 if active else 0.5
    u_hi = torch.full((nu,), b, dtype=dtype, device=device)
    u_lo = -u_hi
    return Problem(B, T, nx, nu, Qd, q, F, c, x0, u_lo, u_hi, xref.clone(), xref)


class AffineDynamics:
    """``dx``/``dx_jac`` callables for time-varying affine dynamics.

    Follows the calling convention the reference's AL path uses for its
    dynamics provider: ``dx(x[K,nx], u[K,nu]) -> x_next[K,nx]`` with
    ``K = m*B*(T-1)`` ordered (m, b, t) (``qpth/al_utils.py:215`` and the
    20-fold replication of the line search, ``:56-70, 629-633``) and
    ``dx_jac(x, u) -> (x_next, (A[K,nx,nx], B[K,nx,nu]))`` (``:237-248``).

    It also exposes ``F``/``f`` like the reference's ``LinDx`` tuple
    (``qpth/al_utils.py:9``) so the MI355X solver can take the fused path.
    """

    def __init__(self, F, c):
        self.F = F            # [B, T-1, nx, n]
        self.f = c            # [B, T-1, nx]
        self.B, self.Tm1, self.nx, self.n = F.shape

    def __call__(self, x, u):
        K = x.shape[0]
        m = K // (self.B * self.Tm1)
        xu = torch.cat([x, u], dim=-1).view(m, self.B, self.Tm1, self.n)
        F = self.F.to(xu.dtype)
        c = self.f.to(xu.dtype)
        out = (F[None] * xu[..., None, :]).sum(-1) + c[None]
        return out.reshape(K, self.nx)

    def jac(self, x, u):
        xn = self(x, u)
        K = x.shape[0]
        m = K // (self.B * self.Tm1)
        F = self.F.to(x.dtype)[None].expand(m, -1, -1, -1, -1).reshape(K, self.nx, self.n)
        return xn, (F[..., : self.nx], F[..., self.nx:])


class PendulumDynamics:
    """Small nonlinear test dynamics (explicit-Euler damped pendulum, nx=2, nu=1).

    Not taken from the reference (its ``envs.PendulumEnv`` is missing from the
    tree); it only supplies a smooth nonlinear ``dx``/``dx_jac``
    pair with an analytic Jacobian for the nonlinear-caller mode.
    """

    nx, nu = 2, 1

    def __init__(self, dt=0.05, damping=0.1):
        self.dt = dt
        self.damping = damping

    def __call__(self, x, u):
        th, om = x[:, 0], x[:, 1]
        dt = self.dt
        th1 = th + dt * om
        om1 = om + dt * (-torch.sin(th) - self.damping * om + u[:, 0])
        return torch.stack([th1, om1], dim=-1)

    def jac(self, x, u):
        xn = self(x, u)
        K = x.shape[0]
        dt = self.dt
        A = torch.zeros(K, 2, 2, dtype=x.dtype, device=x.device)
        A[:, 0, 0] = 1.0
        A[:, 0, 1] = dt
        A[:, 1, 0] = -dt * torch.cos(x[:, 0])
        A[:, 1, 1] = 1.0 - dt * self.damping
        Bm = torch.zeros(K, 2, 1, dtype=x.dtype, device=x.device)
        Bm[:, 1, 0] = dt
        return xn, (A, Bm)
