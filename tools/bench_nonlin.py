#!/usr/bin/env python3
"""Nonlinear-caller mode of the drop-in MPC with the HIP dynamics providers vs the same dynamics
written in PyTorch (autograd Jacobians), pendulum1l (B=4096, T=5) and cartpole1l (B=4096, T=10)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deq_mpc_corl_amd import MPC, QuadCost, Pendulum1lDynamics, Cartpole1lDynamics, Cartpole2lDynamics

dev, dt = "cuda:0", torch.float64
h = 0.05


def rk4(acc, q, qd, tau):
    k1q, k1v = qd, acc(q, qd, tau)
    k2q, k2v = qd + 0.5 * h * k1v, acc(q + 0.5 * h * k1q, qd + 0.5 * h * k1v, tau)
    k3q, k3v = qd + 0.5 * h * k2v, acc(q + 0.5 * h * k2q, qd + 0.5 * h * k2v, tau)
    k4q, k4v = qd + h * k3v, acc(q + h * k3q, qd + h * k3v, tau)
    return q + h / 6 * (k1q + 2 * k2q + 2 * k3q + k4q), qd + h / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)


def pend_torch(x, u):
    qn, vn = rk4(lambda q, v, t: 4.0 * t - 19.62 * torch.sin(q), x[:, :1], x[:, 1:], u)
    return torch.cat((qn, vn), 1)


def cart_torch(x, u):
    def acc(q, v, t):
        th, thd = q[:, 1], v[:, 1]
        sn, cs = torch.sin(th), torch.cos(th)
        r0, r1 = t[:, 0] - sn * thd * thd, 9.81 * sn
        idet = 1.0 / (22.0 - cs * cs)
        return torch.stack((idet * (2 * r0 + cs * r1), idet * (cs * r0 + 11 * r1)), 1)
    qn, vn = rk4(acc, x[:, :2], x[:, 2:], u)
    return torch.cat((qn, vn), 1)


def cart2_torch(x, u):
    def acc(q, v, t):
        t1, t2, w1, w2 = q[:, 1], q[:, 2], v[:, 1], v[:, 2]
        s1, c1, s2, c2 = torch.sin(t1), torch.cos(t1), torch.sin(t2), torch.cos(t2)
        s12, c12 = torch.sin(t1 + t2), torch.cos(t1 + t2)
        one = torch.ones_like(t1)
        M = torch.stack([torch.stack([12 * one, -(2 * c1 + c12), -c12], 1),
                         torch.stack([-(2 * c1 + c12), 5 + 2 * c2, 2 + c2], 1),
                         torch.stack([-c12, 2 + c2, 2 * one], 1)], 1)
        r = torch.stack([t[:, 0] - (2 * s1 * w1 ** 2 + s12 * (w1 + w2) ** 2),
                         s2 * w2 * (2 * w1 + w2) + 9.81 * (2 * s1 + s12),
                         -s2 * w1 ** 2 + 9.81 * s12], 1)
        return torch.linalg.solve(M, r.unsqueeze(-1)).squeeze(-1)
    qn, vn = rk4(acc, x[:, :3], x[:, 3:], u)
    return torch.cat((qn, vn), 1)


def autograd_jac(f, nx):
    def jac(x, u):
        with torch.enable_grad():
            xr, ur = x.detach().requires_grad_(True), u.detach().requires_grad_(True)
            xn = f(xr, ur)
            rows = [torch.autograd.grad(xn[:, i].sum(), (xr, ur), retain_graph=True) for i in range(nx)]
        return xn.detach(), (torch.stack([r[0] for r in rows], 1), torch.stack([r[1] for r in rows], 1))
    return jac


out = []
for name, nx, T, B, prov, ft in (("pendulum1l", 2, 5, int(os.environ.get("NL_B", "4096")), Pendulum1lDynamics(h), pend_torch),
                                 ("cartpole1l", 4, 10, int(os.environ.get("NL_B", "4096")), Cartpole1lDynamics(h), cart_torch),
                                 ("cartpole2l", 6, 10, int(os.environ.get("NL_B", "4096")), Cartpole2lDynamics(h), cart2_torch)):
    g = torch.Generator().manual_seed(1)
    x0 = (0.3 * torch.randn(B, nx, generator=g)).to(dt).to(dev)
    n = nx + 1
    Qd = torch.ones(B, T, n, dtype=dt, device=dev)
    Qd[..., -1] = 0.1
    q = torch.zeros(B, T, n, dtype=dt, device=dev)
    cost = QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dt, device=dev))
    class Plain:   # the provider kernels without the model id: launch-per-phase path
        def __call__(self, x, u):
            return prov(x, u)

        def jac(self, x, u):
            return prov.jac(x, u)

    plain = Plain()
    for label, dx, dxj, mode in (("model inlined: one launch per Newton step, exit test on the device (reference exit)", prov, prov.jac, "reference"),
                                 ("pytorch + autograd, reference exit", ft, autograd_jac(ft, nx), "reference"),
                                 ("provider kernels for dx/dx_jac, launch per phase, reference exit", plain, plain.jac, "reference"),
                                 ("provider kernels for dx/dx_jac, launch per phase, fixed 4 steps", plain, plain.jac, "fixed"),
                                 ("model inlined: ONE launch (alqp_solve_nonlin), fixed 4 steps", prov, prov.jac, "fixed")):
        ts = []
        for rep in range(4):
            mpc = MPC(nx, 1, T, u_lower=-5.0, u_upper=5.0, n_batch=B, dtype=dt, exit_mode=mode, prefer_fused=True)
            mpc.reinitialize(x0, None)
            mpc.al_iter = 2
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, u, _ = mpc(x0, cost, dx, dxj)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        r = {"env": name, "B": B, "T": T, "dynamics": label, "ms_per_call": 1e3 * min(ts), "newton_per_al": list(mpc.last_newton_per_al),
             "x_checksum": float(x.double().abs().sum())}
        print(json.dumps(r)); out.append(r)
json.dump(out, open("gpurun_out/nonlin_modes.json", "w"), indent=1)
