#!/usr/bin/env python3
"""Golden fixtures for the INTERIOR-POINT path, made by RUNNING the reference:
`qpth.qp_wrapper.MPC` called directly (the `ip` dispatch inside policies.Tracking_MPC is stale in
this snapshot), with hooks on `qpth.solvers.pdipm.batch_LU` that record what the batched
primal-dual interior-point method computes.

Recorded per case
  * inputs, time-major like the reference: C [T,B,n,n] (diagonal-embedded, as Tracking_MPC builds it,
    policies.py:1172,1265), c [T,B,n], F [T-1,B,nx,n], f [T-1,B,nx], x0, bounds, u_init;
  * per QP solve (qp.DenseQPFunction -> pdipm_b_LU.forward, batch_LU.py:29-210): the initial point
    after the positivity shift (:69-81), per IPM iteration mu / residual sum / step sizes, the number of
    iterations executed, the returned (best) zhat, nus, lams, slacks and residuals;
  * how often `get_step`'s batch-global `a.max()` (batch_LU.py:207) decided a step length
    (`gs_coupled`): the one place where an instance's result depends on who else is in the batch besides
    the exit rule;
  * the rollout line search (qp_wrapper.py:402-421): alpha, and the returned x, u [T,B,.];
  * optionally the backward pass through DenseQPFunction (qp.py:238-270): grads w.r.t. C, c.

Usage:  python tools/gen_golden_ip.py      # writes tests/golden/ip_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sys.path for the reference + stubs)
from qpth import qp_wrapper  # noqa: E402  (the reference)
from qpth.solvers.pdipm import batch_LU  # noqa: E402

# torch.linalg.lu_solve on CPU hangs in this image with more than one intra-op thread on matrices of a
# few hundred rows ("Intel oneMKL ERROR: Parameter 6 was incorrect on entry to DLASWP", then a spin):
# an environment defect, not the reference's; the KKT solves (batch_LU.py:212-244) run single-threaded
torch.set_num_threads(1)

problems = gg.problems
np_ = gg.np_
OUT = gg.OUT


class IpmRecorder:
    """Hooks batch_LU.forward / get_step / solve_kkt."""

    def __init__(self):
        self.solves = []
        self._orig = None

    def __enter__(self):
        rec = self
        self._orig = (batch_LU.forward, batch_LU.get_step, batch_LU.solve_kkt)
        orig_fwd, orig_gs, orig_kkt = self._orig

        def get_step(v, dv):
            out = orig_gs(v, dv)
            # replicate to see whether the batch-global a.max() decided any instance's step
            a = -v / dv
            a[dv == 0] = 1.0
            big = max(1.0, float(a.max()))
            per = a.clone()
            per[dv > 0] = float("inf")
            alt = per.min(1)[0]
            coupled = ((alt != out) & (0.999 * out < 1.0)).sum().item()
            rec.solves[-1]["gs_coupled"] += int(coupled)
            rec.solves[-1]["gs_calls"] += 1
            return out

        def solve_kkt(K, Ktilde, rx, rs, rz, ry, niter=1):
            rec.solves[-1]["kkt_calls"] += 1
            out = orig_kkt(K, Ktilde, rx, rs, rz, ry, niter)
            if rec.solves[-1]["kkt_calls"] == 1:   # the initial point (before the shift)
                rec.solves[-1]["init"] = [o.detach().clone() for o in out]
            return out

        def forward(K, Didx, Q, p, G, GT, h, A, AT, b, dyn_res, cost_grad=None, eps=1e-12, verbose=0,
                    notImprovedLim=3, maxIter=20):
            rec.solves.append({"gs_coupled": 0, "gs_calls": 0, "kkt_calls": 0,
                               "p": p.detach().clone(), "h": h.detach().clone(), "b": b.detach().clone(),
                               "Qdiag": Q.diagonal(dim1=-2, dim2=-1).detach().clone()})
            out = orig_fwd(K, Didx, Q, p, G, GT, h, A, AT, b, dyn_res, cost_grad, eps, verbose, notImprovedLim, maxIter)
            s = rec.solves[-1]
            s["zhat"], s["nus"], s["lams"], s["slacks"] = [o.detach().clone() for o in out[:4]]
            # iterations executed: 1 + 2 solve_kkt calls per completed iteration
            s["iters"] = (s["kkt_calls"] - 1) // 2
            return out

        batch_LU.forward, batch_LU.get_step, batch_LU.solve_kkt = forward, get_step, solve_kkt
        return self

    def __exit__(self, *exc):
        batch_LU.forward, batch_LU.get_step, batch_LU.solve_kkt = self._orig


def run_ip(name, B, T, nx, nu, dtype, kind="lindx", qp_iter=1, seed=0, active=False, backward=False,
           ubound=None, u_init_scale=0.0):
    if gg.ONLY and gg.ONLY not in name:
        return
    n = nx + nu
    p = problems.synthetic_problem(B, T, nx, nu, seed=seed, dtype=dtype, active=active)
    if ubound is not None:
        p = p._replace(u_hi=torch.full((nu,), ubound, dtype=dtype), u_lo=torch.full((nu,), -ubound, dtype=dtype))
    Ft = p.F.transpose(0, 1).contiguous()      # [T-1,B,nx,n]
    ft = p.c.transpose(0, 1).contiguous()      # [T-1,B,nx]
    Cd = p.Qd.transpose(0, 1).contiguous()     # [T,B,n]
    ct = p.q.transpose(0, 1).contiguous()      # [T,B,n]
    g = torch.Generator().manual_seed(5 + seed)
    u_init = (u_init_scale * torch.randn(T, B, nu, generator=g, dtype=dtype)).contiguous()
    if backward:
        Cd.requires_grad_(True)
        ct.requires_grad_(True)
    C = torch.diag_embed(Cd)
    if kind == "lindx":
        dx = qp_wrapper.LinDx(Ft, ft)
        dx_jac = None
    elif kind == "pendulum":
        dyn = problems.PendulumDynamics()
        dx, dx_jac = dyn, dyn.jac
    else:
        raise ValueError(kind)
    mpc = qp_wrapper.MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, qp_iter=qp_iter, exit_unconverged=False,
                         eps=1e-5, n_batch=B, backprop=False, verbose=0, u_init=u_init,
                         grad_method=qp_wrapper.GradMethods.ANALYTIC, solver_type="dense",
                         single_qp_solve=(qp_iter == 1))
    alphas = []
    orig_ls = mpc.line_search

    def ls(*a):
        out = orig_ls(*a)
        alphas.append(out[2].detach().clone())
        return out

    mpc.line_search = ls
    with IpmRecorder() as rec:
        x, u = mpc(p.x0, qp_wrapper.QuadCost(C, ct), dx, dx_jac)
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "dtype": "f64" if dtype == torch.float64 else "f32",
           "kind": kind, "qp_iter": qp_iter, "Cd": np_(Cd), "c": np_(ct), "F": np_(Ft), "f": np_(ft),
           "x0": np_(p.x0), "u_lo": np_(p.u_lo), "u_hi": np_(p.u_hi), "u_init": np_(u_init),
           "x": np_(x), "u": np_(u), "alpha": np.stack([np_(a) for a in alphas]),
           "n_solves": len(rec.solves),
           "ipm_iters": np.array([s["iters"] for s in rec.solves], np.int32),
           "gs_coupled": np.array([s["gs_coupled"] for s in rec.solves], np.int32),
           "gs_calls": np.array([s["gs_calls"] for s in rec.solves], np.int32)}
    for key in ("p", "h", "b", "Qdiag", "zhat", "nus", "lams", "slacks"):
        out["qp_" + key] = np.stack([np_(s[key]) for s in rec.solves])
    for i, nm in enumerate(("x", "s", "z", "y")):
        out["qp_init_" + nm] = np.stack([np_(s["init"][i]) for s in rec.solves])
    if backward:
        gw = torch.Generator().manual_seed(1234)
        wx = torch.randn(T, B, nx, generator=gw, dtype=dtype)
        wu = torch.randn(T, B, nu, generator=gw, dtype=dtype)
        ((x * wx).sum() + (u * wu).sum()).backward()
        out.update(bwd_wx=np_(wx), bwd_wu=np_(wu), bwd_c_grad=np_(ct.grad), bwd_Cd_grad=np_(Cd.grad))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: solves={len(rec.solves)} ipm_iters={out['ipm_iters'].tolist()} gs_coupled={out['gs_coupled'].tolist()} "
          f"of {out['gs_calls'].tolist()} alpha_min={float(out['alpha'].min()):.3g} "
          f"at_bound={float((np.abs(np_(u)) > float(p.u_hi[0]) - 1e-6).mean()):.2f} size={os.path.getsize(path) / 1024:.0f}KB")


def main():
    f64, f32 = torch.float64, torch.float32
    run_ip("ip_pend_f64", 8, 5, 2, 1, f64, backward=True)
    run_ip("ip_pend_f32", 8, 5, 2, 1, f32)
    run_ip("ip_cart_f64", 8, 10, 8, 2, f64, backward=True)
    run_ip("ip_cart_active_f64", 8, 10, 8, 2, f64, active=True, backward=True)
    run_ip("ip_cart_f32", 8, 10, 8, 2, f32)
    run_ip("ip_quad13_f64", 4, 20, 13, 4, f64)
    run_ip("ip_quad13_active_f64", 4, 20, 13, 4, f64, active=True)
    run_ip("ip_quad13_f32", 4, 20, 13, 4, f32)
    run_ip("ip_quad12_f64", 4, 20, 12, 4, f64)
    run_ip("ip_fcp14_f64", 2, 10, 14, 4, f64)
    # the iterated (SQP-style) route: qp_iter > 1 -> solve_nonlin (qp_wrapper.py:345-387)
    run_ip("ip_cart_sqp3_f64", 8, 10, 8, 2, f64, qp_iter=3, u_init_scale=0.1)
    # nonlinear dynamics: re-linearised per QP, and the IPM's equality residual is the TRUE dynamics
    # residual of the iterate (dyn_res callback, qp_wrapper.py:306, batch_LU.py:95)
    run_ip("ip_pend_nonlin_f64", 8, 5, 2, 1, f64, kind="pendulum", backward=True)
    run_ip("ip_pend_nonlin_sqp3_f64", 8, 5, 2, 1, f64, kind="pendulum", qp_iter=3, u_init_scale=0.1)


if __name__ == "__main__":
    main()
