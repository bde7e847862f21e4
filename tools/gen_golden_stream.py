#!/usr/bin/env python3
"""Golden fixtures for the STREAMING / warm-start route of the reference, made by RUNNING it
(build container only; same import recipe as tools/gen_golden.py).

Pins SURVEY.md rows a4 (`MPC.al_solve_stream`, qpth/AL_mpc.py:342-423), a5
(`warm_start_initialize`, :581-592) and a13 (`linearize_once`, :370-391 with
qpth/al_utils_lin.py:140-189): the call sequence is the one `policies.DEQMPCPolicy` drives
(policies.py:147-171 then :205-262):

    reinitialize -> __call__ (al_solve) -> warm_start_initialize(x, u, args) -> __call__ x N (al_solve_stream)

Recorded per call: q, the warm start handed in, x/u/status returned, lamda/rho carried, the number
of AL iterations the stream loop executed, Newton steps per AL iteration, and per AL iteration the
(xu, lamda, rho) NewtonAL.apply received and the iterate it returned (hooks on both
al_utils.NewtonAL.apply and al_utils_lin.NewtonAL.apply) - enough to tell a mis-read break rule
(batch-mean residual, :406-408; rho.max() > 1e8, :412, :420) from a correct one.

Also `run_tracking_stream`: the same through the unchanged `policies.Tracking_MPC`
(reinitialize / warm_start_initialize of policies.py:1299-1310).

Usage:  python tools/gen_golden_stream.py     # writes tests/golden/*stream*.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up sys.path for the reference + stubs)
from qpth import AL_mpc, al_utils, al_utils_lin  # noqa: E402  (the reference)

problems = gg.problems
np_ = gg.np_
OUT = gg.OUT


class ApplyRecorder:
    """Logs every NewtonAL.apply (either module): inputs xu/lam/rho, output iterate, Newton steps."""

    def __init__(self):
        self.calls = []     # dicts: module, xu, lam, rho, z_out, newton
        self._o = {}

    def __enter__(self):
        rec = self
        for mod in (al_utils, al_utils_lin):
            self._o[mod] = (mod.NewtonAL.apply, mod.merit_grad_hessian)

            def gh(*a, _orig=mod.merit_grad_hessian, **k):
                rec.calls[-1]["newton"] += 1
                return _orig(*a, **k)

            def apply(*a, _orig=mod.NewtonAL.apply, _name=mod.__name__.split(".")[-1]):
                rec.calls.append({"module": _name, "xu": a[4].detach().clone(), "lam": a[6].detach().clone(),
                                  "rho": a[7].detach().clone(), "newton": 0})
                out = _orig(*a)
                rec.calls[-1]["z_out"] = out[0].detach().clone()
                return out

            mod.merit_grad_hessian = gh
            mod.NewtonAL.apply = staticmethod(apply)
        return self

    def __exit__(self, *exc):
        for mod, (ap, gh) in self._o.items():
            mod.NewtonAL.apply = ap
            mod.merit_grad_hessian = gh


def _dyn(kind, p):
    if kind == "affine":
        return problems.AffineDynamics(p.F, p.c)
    if kind == "pendulum":
        return problems.PendulumDynamics()
    if kind == "casadi_pendulum1l":
        return gg.CasadiPendulum1l(0.05)
    if kind == "casadi_cartpole1l":
        return gg.CasadiCartpole1l(0.05)
    raise ValueError(kind)


def run_stream(name, B, T, nx, nu, dtype, kind, linearize_once, stream_calls=2, al_iter_first=2,
               al_iter_stream=2, rho_init_max=1e4, seed=0, active=False, scale=None, backward=False):
    if gg.ONLY and gg.ONLY not in name:
        return
    p = problems.synthetic_problem(B, T, nx, nu, seed=seed, dtype=dtype, active=active)
    if scale is not None:
        p = p._replace(x0=scale * p.x0, z0=scale * p.z0, xref=scale * p.xref, q=-(p.Qd * scale * p.xref))
    dyn = _dyn(kind, p)
    C = torch.diag_embed(p.Qd)
    mpc = AL_mpc.MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dtype)
    mpc.reinitialize(p.x0, None)
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "dtype": "f64" if dtype == torch.float64 else "f32",
           "kind": kind, "linearize_once": int(linearize_once), "rho_init_max": rho_init_max,
           "al_iter_first": al_iter_first, "al_iter_stream": al_iter_stream, "stream_calls": stream_calls,
           "Qd": np_(p.Qd), "F": np_(p.F), "c": np_(p.c), "x0": np_(p.x0), "u_lo": np_(p.u_lo),
           "u_hi": np_(p.u_hi), "z0": np_(p.z0)}
    g = torch.Generator().manual_seed(77)
    # ---- call 0: the ordinary solve after reinitialize (al_solve) --------------------------------
    cost = al_utils.QuadCost(C, p.q, torch.zeros(B, T, dtype=dtype))
    mpc.al_iter = al_iter_first
    with ApplyRecorder() as rec:
        x, u, st = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
    out.update(q0=np_(p.q), x_first=np_(x), u_first=np_(u), lam_first=np_(mpc.lamda_prev),
               rho_first=np_(mpc.rho_prev), newton_first=np.array([c["newton"] for c in rec.calls], np.int32))
    # ---- warm start, the way Tracking_MPC.warm_start_initialize does it (policies.py:1305-1310) ----
    xref2 = p.xref + 0.05 * torch.randn(B, T, nx + nu, generator=g, dtype=dtype)
    xref2[..., nx:] = 0
    xw, uw = mpc.x_init, mpc.u_init          # fp32 tensors (the solver's returned, detached iterate)
    xw[:, -1:] = xref2[:, -1:, :nx].to(xw.dtype)
    uw[:, -1:] = xref2[:, -1:, nx:].to(uw.dtype)
    out["x_warm"], out["u_warm"] = np_(xw), np_(uw)
    mpc.warm_start_initialize(xw, uw, SimpleNamespace(rho_init_max=rho_init_max))
    assert float(mpc.lamda_prev.abs().max()) == 0.0
    out["rho_after_warm"] = np_(mpc.rho_prev)
    mpc.linearize_once = bool(linearize_once)
    qs, xs, us, sts, lams, rhos, nal, newt = [], [], [], [], [], [], [], []
    ap_xu, ap_lam, ap_rho, ap_z, ap_call = [], [], [], [], []
    bwd = {}
    for ci in range(stream_calls):
        xr = xref2 + 0.05 * ci * torch.randn(B, T, nx + nu, generator=g, dtype=dtype)
        xr[..., nx:] = 0
        q = -(p.Qd * xr)
        Qd = p.Qd.clone()
        last = backward and ci == stream_calls - 1
        if last:
            q.requires_grad_(True)
            Qd.requires_grad_(True)
        cost = al_utils.QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dtype))
        mpc.al_iter = al_iter_stream
        with ApplyRecorder() as rec:
            x, u, st = mpc(p.x0, cost, dyn, dyn.jac)
        if last:
            gw = torch.Generator().manual_seed(1234)
            wx = torch.randn(B, T, nx, generator=gw, dtype=torch.float32)
            wu = torch.randn(B, T, nu, generator=gw, dtype=torch.float32)
            ((x * wx).sum() + (u * wu).sum()).backward()
            bwd = {"bwd_wx": np_(wx), "bwd_wu": np_(wu), "bwd_q_grad": np_(q.grad), "bwd_Qd_grad": np_(Qd.grad)}
        qs.append(np_(q)); xs.append(np_(x)); us.append(np_(u)); sts.append(int(bool(st)))
        lams.append(np_(mpc.lamda_prev)); rhos.append(np_(mpc.rho_prev))
        nal.append(len(rec.calls))
        for c in rec.calls:
            newt.append(c["newton"]); ap_call.append(ci)
            ap_xu.append(np_(c["xu"])); ap_lam.append(np_(c["lam"])); ap_rho.append(np_(c["rho"]))
            ap_z.append(np_(c["z_out"]))
            assert c["module"] == ("al_utils_lin" if linearize_once else "al_utils")
    out.update(q=np.stack(qs), x=np.stack(xs), u=np.stack(us), status=np.array(sts, np.int32),
               lam=np.stack(lams), rho=np.stack(rhos), n_al=np.array(nal, np.int32),
               newton=np.array(newt, np.int32), ap_call=np.array(ap_call, np.int32),
               ap_xu=np.stack(ap_xu), ap_lam=np.stack(ap_lam), ap_rho=np.stack(ap_rho), ap_z=np.stack(ap_z), **bwd)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: first newton={out['newton_first'].tolist()} n_al={nal} newton={newt} status={sts} "
          f"rho_max={[float(r.max()) for r in rhos]} size={os.path.getsize(path) / 1024:.0f}KB")


def run_tracking_stream(name, B, T, nx, nu, linearize_once=False):
    """reinitialize -> 2 calls -> warm_start_initialize -> 2 calls through the reference's own
    policies.Tracking_MPC (policies.py:1236-1310), affine synthetic env."""
    if gg.ONLY and gg.ONLY not in name:
        return
    import policies  # noqa: F401  (reference, deqmpc/policies.py)
    dtype = torch.float64
    p = problems.synthetic_problem(B, T, nx, nu, seed=13, dtype=dtype)
    dyn = problems.AffineDynamics(p.F, p.c)
    env = SimpleNamespace(nu=nu, nx=nx, nq=nx // 2, dt=0.05, dynamics=dyn, dynamics_derivatives=dyn.jac,
                          action_space=SimpleNamespace(high=np.full(nu, 0.5), low=np.full(nu, -0.5)))
    args = SimpleNamespace(T=T, device="cpu", qp_iter=1, eps=1e-2, warm_start=True, bsz=B,
                           Q=torch.tensor([10.0] * nx), R=torch.tensor([1e-8] * nu), dtype="double",
                           solver_type="al", env="synthetic", rho_init_max=1e3)
    torch.manual_seed(5)
    tm = policies.Tracking_MPC(args, env)
    x_ref = p.xref[..., :nx].clone()
    u_ref = p.xref[..., nx:].clone()
    tm.reinitialize(x_ref, torch.ones(B, T, 1, dtype=dtype))
    g = torch.Generator().manual_seed(3)
    xs, us, xrefs, sts, rhos, phase = [], [], [], [], [], []
    for i in range(2):
        xr = x_ref + 0.05 * i * torch.randn(B, T, nx, generator=g, dtype=dtype)
        x, u, st = tm(p.x0, None, xr, u_ref, al_iters=2)
        xs.append(np_(x)); us.append(np_(u)); xrefs.append(np_(xr)); sts.append(int(bool(st)))
        rhos.append(np_(tm.ctrl.rho_prev)); phase.append(0)
    xr_w = x_ref + 0.05 * torch.randn(B, T, nx, generator=g, dtype=dtype)
    tm.warm_start_initialize(xr_w, u_ref)
    tm.ctrl.linearize_once = bool(linearize_once)
    for i in range(3):
        xr = xr_w + 0.03 * i * torch.randn(B, T, nx, generator=g, dtype=dtype)
        x, u, st = tm(p.x0, None, xr, u_ref, al_iters=2)
        xs.append(np_(x)); us.append(np_(u)); xrefs.append(np_(xr)); sts.append(int(bool(st)))
        rhos.append(np_(tm.ctrl.rho_prev)); phase.append(1)
    out = {"B": B, "T": T, "nx": nx, "nu": nu, "F": np_(p.F), "c": np_(p.c), "x0": np_(p.x0),
           "x_ref": np.stack(xrefs), "u_ref": np_(u_ref), "x_ref_warm": np_(xr_w), "x": np.stack(xs), "u": np.stack(us),
           "status": np.array(sts, np.int32), "rho": np.stack(rhos), "phase": np.array(phase, np.int32),
           "lam": np_(tm.ctrl.lamda_prev), "rho_init_max": 1e3, "linearize_once": int(linearize_once)}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: status={sts} rho_max={[float(r.max()) for r in rhos]}")


def main():
    f64, f32 = torch.float64, torch.float32
    # nonlinear dynamics, streaming with re-linearisation every Newton step (al_utils.NewtonAL)
    run_stream("pend_stream_f64", 8, 5, 2, 1, f64, "pendulum", False, backward=True)
    # frozen linearisation: al_utils_lin.NewtonAL, up to 100 AL iterations, batch-mean break
    run_stream("pend_stream_lin_f64", 8, 5, 2, 1, f64, "pendulum", True)   # (al_utils_lin.NewtonAL.backward returns 14 gradients for 15 inputs: the reference cannot differentiate this route)
    run_stream("pend_stream_lin_f32", 8, 5, 2, 1, f32, "pendulum", True)
    # affine data: rho overflow (> 1e8) inside the stream loop -> status True (AL_mpc.py:412, 420)
    run_stream("cart_stream_f64", 8, 10, 8, 2, f64, "affine", False, al_iter_stream=10, rho_init_max=1e4)
    run_stream("cart_stream_lin_f64", 8, 10, 8, 2, f64, "affine", True)
    run_stream("cart_stream_active_f64", 8, 10, 8, 2, f64, "affine", False, active=True, stream_calls=3, rho_init_max=10.0)
    run_stream("quad13_stream_lin_f64", 4, 20, 13, 4, f64, "affine", True)
    # the reference's own compiled dynamics packages under the frozen linearisation
    run_stream("pend1l_casadi_stream_lin_f64", 8, 6, 2, 1, f64, "casadi_pendulum1l", True)
    run_stream("cart1l_casadi_stream_f64", 6, 8, 4, 1, f64, "casadi_cartpole1l", False, scale=0.15, seed=1)
    run_stream("cart1l_casadi_stream_lin_f64", 6, 8, 4, 1, f64, "casadi_cartpole1l", True, scale=0.15, seed=1)
    run_tracking_stream("cart_tracking_stream_f64", 8, 10, 8, 2)
    run_tracking_stream("cart_tracking_stream_lin_f64", 8, 10, 8, 2, linearize_once=True)


if __name__ == "__main__":
    main()
