#!/usr/bin/env python3
"""Golden fixtures for the OBSTACLE rows, made by RUNNING the reference's
`qpth.AL_mpc_custom.Obstacle_MPC` (qpth/AL_mpc_custom.py:22-135; al_utils.py:313-323, 351-388) the way
policies.Tracking_MPC constructs and drives it (policies.py:1181-1198): reinitialize(x_ref, mask) picks
the 4 nearest of 40 spheres per stage, then __call__; optionally warm_start_initialize + a streaming call.

Recorded: inputs (incl. the 40 sphere centres and the radius), the chosen centres [B,T,4,3], per Newton
step g / banded H / d / 20 merits / k / accept / z (tools/gen_golden.py's Recorder), per AL iteration lamda
(with the obstacle rows) and rho, final x, u, the backward gradients.

Usage:  python tools/gen_golden_obs.py      # writes tests/golden/obs_*.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from qpth import al_utils  # noqa: E402  (the reference)
from qpth.AL_mpc_custom import Obstacle_MPC  # noqa: E402

problems, np_, OUT = gg.problems, gg.np_, gg.OUT


class CallableAffine:
    """The synthetic affine dynamics as a plain callable pair (no F / f attributes): Obstacle_MPC is only
    reached with PyTorch-coded dynamics in the reference, i.e. the nonlinear-caller route."""

    def __init__(self, F, c):
        self._d = problems.AffineDynamics(F, c)

    def __call__(self, x, u):
        return self._d(x, u)

    def jac(self, x, u):
        return self._d.jac(x, u)


def run_obs(name, B, T, nx, nu, dtype, al_iter, seed=0, radius=0.6, backward=False, stream=False, active=False):
    if gg.ONLY and gg.ONLY not in name:
        return
    n = nx + nu
    p = problems.synthetic_problem(B, T, nx, nu, seed=seed, dtype=dtype, active=active)
    g = torch.Generator().manual_seed(100 + seed)
    # 40 spheres scattered where the reference trajectories live (positions ~ N(0,1)): several within `radius`
    centres = torch.randn(40, 3, generator=g, dtype=dtype)
    env = SimpleNamespace(obstacle_radius=radius, obstacle_positions=centres)
    dyn = CallableAffine(p.F, p.c)
    Qd, q = p.Qd.clone(), p.q.clone()
    if backward:
        Qd.requires_grad_(True)
        q.requires_grad_(True)
    cost = al_utils.QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dtype))
    mpc = Obstacle_MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dtype, env=env)
    x_ref = p.xref[..., :nx].clone()
    mpc.reinitialize(x_ref, None)
    mpc.al_iter = al_iter
    rec = gg.Recorder(T, n)
    with rec:
        x, u, status = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "al_iter": al_iter, "dtype": "f64" if dtype == torch.float64 else "f32",
           "Qd": np_(p.Qd), "q": np_(p.q), "F": np_(p.F), "c": np_(p.c), "x0": np_(p.x0), "u_lo": np_(p.u_lo),
           "u_hi": np_(p.u_hi), "z0": np_(p.z0), "x_ref": np_(x_ref), "centres": np_(centres), "radius": radius,
           "obs_pos": np_(mpc.obstacles[0]), "x": np_(x), "u": np_(u), "status": int(bool(status)),
           "newton_per_al": np.array(rec.newton_per_al, dtype=np.int32),
           "lam_final": np_(mpc.lamda_prev), "rho_final": np_(mpc.rho_prev),
           "lam_hist": np.stack([np_(l) for l in mpc.cost_lam_hist[1][1:]]),
           "rho_hist": np.stack([np_(r) for r in mpc.cost_lam_hist[2][1:]])}
    steps = rec.steps
    out["n_steps_recorded"] = len(steps)
    for key in ("g", "d", "phi", "phi_prev", "k", "accept", "z"):
        out["step_" + key] = np.stack([np_(s[key]) for s in steps])
    hidx, hd, hs = [], [], []
    for i, s in enumerate(steps):
        if "H" in s:
            d_, s_ = rec.band(s["H"])
            hidx.append(i); hd.append(np_(d_)); hs.append(np_(s_))
    out["H_step_index"] = np.array(hidx, dtype=np.int32)
    out["H_diag"], out["H_sub"] = np.stack(hd), np.stack(hs)
    M = out["lam_final"].shape[1]
    viol = (out["lam_final"][:, T * nx:].reshape(B, T, -1)[:, :, 2 * nu:] > 0).mean()
    if backward:
        gw = torch.Generator().manual_seed(1234)
        wx = torch.randn(B, T, nx, generator=gw, dtype=torch.float32)
        wu = torch.randn(B, T, nu, generator=gw, dtype=torch.float32)
        ((x * wx).sum() + (u * wu).sum()).backward()
        out.update(bwd_wx=np_(wx), bwd_wu=np_(wu), bwd_q_grad=np_(q.grad), bwd_Qd_grad=np_(Qd.grad))
    if stream:
        xw, uw = mpc.x_init, mpc.u_init
        mpc.warm_start_initialize(xw, uw, SimpleNamespace(rho_init_max=1e3))
        out["obs_pos_warm"] = np_(mpc.obstacles[0])
        out["x_warm"], out["u_warm"] = np_(xw), np_(uw)
        mpc.al_iter = 2
        x2, u2, st2 = mpc(p.x0, al_utils.QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dtype)), dyn, dyn.jac)
        out.update(x_stream=np_(x2), u_stream=np_(u2), status_stream=int(bool(st2)), lam_stream=np_(mpc.lamda_prev),
                   rho_stream=np_(mpc.rho_prev))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: M={M} newton_per_al={rec.newton_per_al} obstacle multipliers > 0: {viol:.2f} "
          f"accept={float(np.mean(out['step_accept'])):.2f} size={os.path.getsize(path) / 1024:.0f}KB")


def main():
    f64, f32 = torch.float64, torch.float32
    run_obs("obs_cart_f64_al2", 6, 10, 8, 2, f64, 2, backward=True)
    run_obs("obs_cart_f64_al4", 6, 10, 8, 2, f64, 4, seed=1, radius=0.8, stream=True)
    run_obs("obs_cart_f32_al2", 6, 10, 8, 2, f32, 2)
    run_obs("obs_fcp14_f64_al3", 3, 10, 14, 4, f64, 3, seed=2, radius=0.7, backward=True)
    run_obs("obs_quad13_f64_al2", 3, 20, 13, 4, f64, 2, seed=3, radius=0.7)


if __name__ == "__main__":
    main()
