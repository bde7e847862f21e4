#!/usr/bin/env python3
"""Fixture for the FAILURE path: a batch in which some instances carry a penalty rho so
large that the reference's `torch.linalg.cholesky_ex` reports a non-positive pivot (al_utils.py:510), made by
RUNNING the reference. Recorded: per Newton step and instance cholesky_ex's `info`, whether the
`linalg.solve` fallback fired (:517-521), the line search's accept bits, the returned x, u, lamda, rho.

What the reference does there (observed; tools/gen_golden_fail.py prints it): cholesky_ex stops at the bad
pivot and returns the half-finished factor WITHOUT NaN, cholesky_solve uses it, the direction is finite
garbage, and the line search's strict-decrease test decides. The LU fallback only fires when the update
contains NaN/Inf. The healthy instances of the batch are untouched (apart from the batch-global exit test).

Usage: python tools/gen_golden_fail.py     # writes tests/golden/fail_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from qpth import AL_mpc, al_utils  # noqa: E402

problems, np_, OUT = gg.problems, gg.np_, gg.OUT


def run_fail(name, B, T, nx, nu, dtype, rho_init, al_iter=1, seed=0):
    p = problems.synthetic_problem(B, T, nx, nu, seed=seed, dtype=dtype, active=True)
    dyn = problems.AffineDynamics(p.F, p.c)
    cost = al_utils.QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dtype))
    mpc = AL_mpc.MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dtype)
    mpc.reinitialize(p.x0, None)
    mpc.rho_prev = torch.tensor(rho_init, dtype=dtype).reshape(B, 1)
    mpc.al_iter = al_iter
    infos, fallback = [], []
    orig_ex, orig_solve = torch.linalg.cholesky_ex, torch.linalg.solve

    def ex(H, *a, **k):
        U, info = orig_ex(H, *a, **k)
        infos.append(info.clone())
        fallback.append(0)
        return U, info

    def sv(H, g, *a, **k):
        fallback[-1] = 1
        return orig_solve(H, g, *a, **k)

    torch.linalg.cholesky_ex, torch.linalg.solve = ex, sv
    rec = gg.Recorder(T, nx + nu, record_H=False)
    try:
        with rec:
            x, u, status = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
    finally:
        torch.linalg.cholesky_ex, torch.linalg.solve = orig_ex, orig_solve
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "al_iter": al_iter, "dtype": "f64" if dtype == torch.float64 else "f32",
           "Qd": np_(p.Qd), "q": np_(p.q), "F": np_(p.F), "c": np_(p.c), "x0": np_(p.x0), "u_lo": np_(p.u_lo),
           "u_hi": np_(p.u_hi), "z0": np_(p.z0), "rho_init": np.array(rho_init, dtype=np.float64),
           "x": np_(x), "u": np_(u), "chol_info": np.stack([np_(i) for i in infos]),
           "lu_fallback": np.array(fallback, np.int32), "newton_per_al": np.array(rec.newton_per_al, np.int32),
           "step_accept": np.stack([np_(s["accept"]) for s in rec.steps]),
           "step_k": np.stack([np_(s["k"]) for s in rec.steps]),
           "lam_final": np_(mpc.lamda_prev), "rho_final": np_(mpc.rho_prev)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: newton_per_al={rec.newton_per_al} chol info>0 per step: {[(np_(i) > 0).astype(int).tolist() for i in infos]} "
          f"lu_fallback={fallback} accept={[np_(s['accept']).astype(int).tolist() for s in rec.steps]} "
          f"finite={bool(torch.isfinite(x).all())}")


def main():
    f64, f32 = torch.float64, torch.float32
    run_fail("fail_cart_f64", 8, 10, 8, 2, f64, [1, 1, 1e22, 1, 10, 1e24, 1, 1e22])
    run_fail("fail_quad13_f32", 6, 20, 13, 4, f32, [1, 1e11, 1, 1e12, 10, 1])


if __name__ == "__main__":
    main()
