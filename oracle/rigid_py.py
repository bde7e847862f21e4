"""TEST INFRASTRUCTURE - torch (CPU, fp64) restatement of the reference's torch-coded robot dynamics:

    RexQuadrotor_dynamics    deqmpc/rex_quadrotor.py:8-127      (forces :53-68, moments :70-85, dynamics_ :113-127, RK4 :98-107)
    FlyingCartpole_dynamics  deqmpc/flying_cartpole2d.py:11-130 (forces :51-58, moments :60-75, dynamics_ :107-130, RK4 :79-89)

PARITY UNPINNED: both reference files import `rexquad_utils` (rk4, mrp2quat, quatrot, w2pdotkinematics_mrp, ...), which
is absent from the reference tree, so neither can be imported or run and the tree holds no outputs of them. The three
helpers the dynamics use are restated here from their standard definitions (modified Rodrigues parameters, scalar-first
unit quaternions; the conventions of the Julia RobotDynamics / Rotations packages the file names point to); everything
else follows the reference line by line, including the float32 rounding of the constants it keeps in float32 tensors
(J, inv(J), g, the arm directions, Bf, u_hover) before they meet float64 states.

Only tests/ import this (the checker of csrc/alqp_dyn_rigid.hip); autograd through `step` is the Jacobian oracle.
"""
import numpy as np
import torch


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rex_params(mass=2.0, J=((0.01566089, 0.00000318037, 0.0), (0.00000318037, 0.01562078, 0.0), (0.0, 0.0, 0.02226868)),
               gravity=(0, 0, -9.81), motor_dist=0.28, kf=0.0244101, bf=-30.48576, km=0.00029958, dt=0.05):
    """Constructor values of RexQuadrotor_dynamics (rex_quadrotor.py:9-46) as the kernel's parameter block."""
    J32 = np.asarray(J, dtype=np.float32)
    ss = np.array([[1., 1, 0], [1., -1, 0], [-1., -1, 0], [-1., 1, 0]], dtype=np.float32)
    ss = ss / np.linalg.norm(ss, axis=-1, keepdims=True).astype(np.float32)
    return dict(mass=float(mass), J=J32.astype(np.float64), Jinv=np.linalg.inv(J32).astype(np.float32).astype(np.float64),
                g=_f32(gravity), motor_dist=float(motor_dist), kf=float(kf), bf=float(bf), km=float(km), act_scale=100.0,
                u_hover=0.0, pend_L=1.0, ss=ss.astype(np.float64), bf_force=float(np.float32(4 * bf)), dt=float(dt), fly=False)


def flycart_params(mass_q=2.0, mass_p=0.1, J=((0.0023, 0.0, 0.0), (0.0, 0.0023, 0.0), (0.0, 0.0, 0.004)), L=0.5,
                   gravity=(0, 0, -9.81), motor_dist=0.175, kf=1.0, bf=0.0, km=0.025, dt=0.05):
    """Constructor values of FlyingCartpole_dynamics as FlyingCartpole builds it (flying_cartpole2d.py:13-47, 150-152)."""
    m = mass_q + mass_p
    J32 = np.asarray(J, dtype=np.float32)
    ss = np.array([[1., 1, 0], [1., -1, 0], [-1., -1, 0], [-1., 1, 0]], dtype=np.float32)
    ss = ss / np.linalg.norm(ss, axis=-1, keepdims=True).astype(np.float32)
    act_scale = 10.0
    u_hover = float(np.float32((-m * gravity[2]) / act_scale / kf / 4))
    return dict(mass=float(m), J=J32.astype(np.float64), Jinv=np.linalg.inv(J32).astype(np.float32).astype(np.float64),
                g=_f32(gravity), motor_dist=float(motor_dist), kf=float(kf), bf=float(bf), km=float(km), act_scale=act_scale,
                u_hover=u_hover, pend_L=float(np.float32(L)), ss=ss.astype(np.float64), bf_force=0.0, dt=float(dt), fly=True)


# ---- rexquad_utils helpers, restated (NOT in the reference tree) ------------------------------------
def mrp2quat(p):
    n2 = (p * p).sum(-1, keepdim=True)
    return torch.cat([(1 - n2) / (1 + n2), 2 * p / (1 + n2)], dim=-1)


def quatrot(q, v):
    q0, qv = q[..., :1], q[..., 1:]
    t = 2 * torch.cross(qv, v, dim=-1)
    return v + q0 * t + torch.cross(qv, t, dim=-1)


def w2pdotkinematics_mrp(p, w):
    n2 = (p * p).sum(-1, keepdim=True)
    return 0.25 * ((1 - n2) * w + 2 * torch.cross(p, w, dim=-1) + 2 * (p * w).sum(-1, keepdim=True) * p)


def _body(P, m, v, w, u):
    t = lambda a: torch.as_tensor(a, dtype=m.dtype)
    us = P["act_scale"] * (u + P["u_hover"])                               # rex :114 / flycart :112
    q = mrp2quat(m)
    g = t(P["g"])
    F = torch.zeros_like(m)
    F[..., 2] = (P["kf"] * us).sum(-1)                                      # forces: rex :57-59 / flycart :54-56
    F = F + quatrot(mrp2quat(-m), (P["mass"] * g).expand_as(m))
    F[..., 2] = F[..., 2] + P["bf_force"]                                   # + Bf (rex :66; none in the flying cartpole)
    M = P["km"] * us
    tau = torch.zeros_like(m)
    tau[..., 2] = M[..., 0] - M[..., 1] + M[..., 2] - M[..., 3]             # moments :76-78
    ss = t(P["ss"])
    fz = torch.zeros(*us.shape, 3, dtype=m.dtype)
    fz[..., 2] = P["kf"] * us + (0.0 if P["fly"] else P["bf"])              # rex :84 (kf u + bf) / flycart :74 (kf u)
    tau = tau + torch.cross((P["motor_dist"] * ss).expand_as(fz), fz, dim=-1).sum(-2)
    mdot = w2pdotkinematics_mrp(m, w)
    pdot = quatrot(q, v)
    vdot = F / P["mass"] - torch.cross(w, v, dim=-1)
    J, Jinv = t(P["J"]), t(P["Jinv"])
    wdot = (Jinv @ (tau - torch.cross(w, (J @ w[..., None])[..., 0], dim=-1))[..., None])[..., 0]
    return pdot, mdot, vdot, wdot, q


def deriv(P, x, u):
    if not P["fly"]:
        m, v, w = x[..., 3:6], x[..., 6:9], x[..., 9:12]
        pdot, mdot, vdot, wdot, _ = _body(P, m, v, w, u)
        return torch.cat([pdot, mdot, vdot, wdot], dim=-1)
    m, v, w = x[..., 3:6], x[..., 7:10], x[..., 10:13]
    th, thd = x[..., 6:7], x[..., 13:14]
    pdot, mdot, vdot, wdot, q = _body(P, m, v, w, u)
    x_ddot = quatrot(q, vdot)[..., 0:1]
    thdd = (float(P["g"][2]) * torch.sin(th) + x_ddot * torch.cos(th)) / P["pend_L"]    # flycart :126-127
    return torch.cat([pdot, mdot, thd, vdot, wdot, thdd], dim=-1)


def step(P, x, u, h=None):
    """One RK4 step (rex :98-107 / flycart :79-89). x [K,nx], u [K,4] float64 tensors."""
    h = P["dt"] if h is None else h
    k1 = deriv(P, x, u)
    k2 = deriv(P, x + 0.5 * h * k1, u)
    k3 = deriv(P, x + 0.5 * h * k2, u)
    k4 = deriv(P, x + h * k3, u)
    return x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


def jacobian(P, x, u, h=None):
    """[K, nx, nx+nu] by autograd, one point at a time (test sizes only)."""
    out = []
    for k in range(x.shape[0]):
        f = lambda z: step(P, z[None, :x.shape[1]], z[None, x.shape[1]:], h)[0]
        out.append(torch.autograd.functional.jacobian(f, torch.cat([x[k], u[k]])))
    return torch.stack(out)
