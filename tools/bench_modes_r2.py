#!/usr/bin/env python3
"""ms per MPC.__call__ of the drop-in class in its three modes (fused fixed / fused reference exit /
nonlinear-caller with PyTorch dynamics), and the kernel share of the nonlinear-caller call.
Usage (GPU box): python tools/bench_modes_r2.py [B] > profiles/r02/mpc_modes.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T, nx, nu = 20, 13, 4
dev = "cuda:0"
out = {"B": B, "T": T, "nx": nx, "nu": nu}


class Callable:
    """Affine dynamics as plain callables (no F / f): forces the nonlinear-caller route. Uses bmm, not the
    broadcast-multiply-sum of the test helper (which materialises a [20,B,T-1,nx,n] tensor)."""

    def __init__(self, F, c):
        self.F, self.c = F, c
        self.Bn, self.Tm1, self.nx, self.n = F.shape

    def __call__(self, x, u):
        K = x.shape[0]
        m = K // (self.Bn * self.Tm1)
        xu = torch.cat([x, u], -1).view(m, self.Bn * self.Tm1, self.n)
        Ff = self.F.reshape(self.Bn * self.Tm1, self.nx, self.n)
        return (torch.einsum("kij,mkj->mki", Ff, xu) + self.c.reshape(1, -1, self.nx)).reshape(K, self.nx)

    def jac(self, x, u):
        xn = self(x, u)
        Ff = self.F.reshape(-1, self.nx, self.n)
        return xn, (Ff[..., :self.nx], Ff[..., self.nx:])


for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev)
    cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dt, device=dev))
    rec = {}
    for mode, exit_mode, dyn in (("fused_fixed", "fixed", AffineDynamics(p.F, p.c)),
                                 ("fused_reference_exit", "reference", AffineDynamics(p.F, p.c)),
                                 ("nonlinear_caller", "reference", Callable(p.F, p.c))):
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode=exit_mode)

        def call():
            mpc.reinitialize(p.x0, None)
            mpc.al_iter = 2
            return mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())

        call(); call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 5
        for _ in range(k):
            call()
        torch.cuda.synchronize()
        rec[mode] = {"ms_per_call": 1e3 * (time.perf_counter() - t0) / k, "newton_per_al": list(mpc.last_newton_per_al)}
        if mode == "nonlinear_caller":
            # kernel share: time the library calls of one Newton step alone
            from deq_mpc_corl_amd.backend import default_backend
            be = default_backend()
            rec[mode]["step_kernel"] = getattr(be, "last_step_kernel", None)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                call()
                torch.cuda.synchronize()
            tot = sum(e.device_time_total for e in prof.key_averages())
            ours = sum(e.device_time_total for e in prof.key_averages() if "alqp" in e.key or "k_" in e.key[:3])
            rec[mode]["device_ms_total"] = tot / 1e3
            rec[mode]["device_ms_library_kernels"] = ours / 1e3
            rec[mode]["library_kernel_share"] = ours / max(tot, 1)
            rec[mode]["top_kernels"] = [(e.key[:60], round(e.device_time_total / 1e3, 3))
                                        for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]]
    out[name] = rec
print(json.dumps(out, indent=1))
