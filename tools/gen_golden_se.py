#!/usr/bin/env python3
"""Golden fixtures for the STATE-ESTIMATOR variant (second half), made by RUNNING the reference:
`qpth.AL_mpc.MPC(state_estimator=True)` (AL_mpc.py:179-199, 425-446, 507-519 dispatching to qpth/al_utils_se.py).

What that variant solves (read off al_utils_se.py:16-40, 44-75, 92-105, 141-200, 270-300): the controls are GIVEN,
only the states move:  min sum_t 1/2 x_t'Q_x x_t + q_x'x_t  s.t.  x_{t+1} = f(x_t, u_t)  (T-1 row blocks: no
initial-state rows, no bound rows; lamda is [B, nx (T-1)]); the dynamics Jacobian w.r.t. u is multiplied by 0.0
(:151), the cost gradient on u is zeroed (:300-310), the Hessian keeps diag(Q) on u (so du = 0 exactly).

Recorded: inputs, Newton steps per AL iteration, per-AL lamda / rho, x, u, backward gradients.
Usage: python tools/gen_golden_se.py     # writes tests/golden/se_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from qpth import AL_mpc, al_utils, al_utils_se  # noqa: E402

problems, np_, OUT = gg.problems, gg.np_, gg.OUT


class Callable:
    def __init__(self, F, c):
        self._d = problems.AffineDynamics(F, c)

    def __call__(self, x, u):
        return self._d(x, u)

    def jac(self, x, u):
        return self._d.jac(x, u)


def run_se(name, B, T, nx, nu, dtype, al_iter, kind, seed=0, backward=False):
    p = problems.synthetic_problem(B, T, nx, nu, seed=seed, dtype=dtype)
    dyn = problems.PendulumDynamics() if kind == "pendulum" else Callable(p.F, p.c)
    g = torch.Generator().manual_seed(40 + seed)
    z0 = p.z0.clone()
    z0[..., nx:] = 0.3 * torch.randn(B, T, nu, generator=g, dtype=dtype)     # the given controls
    Qd, q = p.Qd.clone(), p.q.clone()
    if backward:
        Qd.requires_grad_(True)
        q.requires_grad_(True)
    cost = al_utils.QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dtype))
    mpc = AL_mpc.MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dtype, state_estimator=True)
    mpc.reinitialize(p.x0, None)
    mpc.al_iter = al_iter
    counts = []
    o_apply, o_gh = al_utils.NewtonAL.apply, al_utils_se.merit_grad_hessian

    def apply(*a):
        counts.append(0)
        return o_apply(*a)

    def gh(*a, **k):
        counts[-1] += 1
        return o_gh(*a, **k)

    al_utils.NewtonAL.apply, al_utils_se.merit_grad_hessian = staticmethod(apply), gh
    try:
        x, u, status = mpc(p.x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    finally:
        al_utils.NewtonAL.apply, al_utils_se.merit_grad_hessian = o_apply, o_gh
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "al_iter": al_iter, "dtype": "f64" if dtype == torch.float64 else "f32",
           "kind": kind, "Qd": np_(p.Qd), "q": np_(p.q), "F": np_(p.F), "c": np_(p.c), "x0": np_(p.x0),
           "u_lo": np_(p.u_lo), "u_hi": np_(p.u_hi), "z0": np_(z0), "x": np_(x), "u": np_(u),
           "newton_per_al": np.array(counts, np.int32), "lam_final": np_(mpc.lamda_prev), "rho_final": np_(mpc.rho_prev),
           "lam_hist": np.stack([np_(l) for l in mpc.cost_lam_hist[1][1:]])}
    if backward:
        gw = torch.Generator().manual_seed(1234)
        wx = torch.randn(B, T, nx, generator=gw, dtype=torch.float32)
        (x * wx).sum().backward()
        out.update(bwd_wx=np_(wx), bwd_q_grad=np_(q.grad), bwd_Qd_grad=np_(Qd.grad))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: lam {tuple(mpc.lamda_prev.shape)} newton_per_al={counts} |u - u_given|max={float((u.double() - z0[..., nx:]).abs().max()):.2e} "
          f"dyn residual {float(mpc.dyn_res_prev.max()):.2e}")


def main():
    f64, f32 = torch.float64, torch.float32
    run_se("se_pend_f64_al3", 8, 5, 2, 1, f64, 3, "pendulum", backward=True)
    run_se("se_cart_f64_al2", 6, 10, 8, 2, f64, 2, "affine", seed=1, backward=True)
    run_se("se_quad13_f64_al2", 3, 20, 13, 4, f64, 2, "affine", seed=2)
    run_se("se_cart_f32_al2", 6, 10, 8, 2, f32, 2, "affine", seed=1)


if __name__ == "__main__":
    main()
