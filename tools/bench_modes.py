#!/usr/bin/env python3
"""End-to-end timing of the drop-in MPC class in its three modes (GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem

dev = "cuda:0"
class CallableOnly:
    """Hides the affine data: forces the nonlinear-caller mode (PyTorch dynamics between launches)."""
    def __init__(self, d): self.d = d
    def __call__(self, x, u): return self.d(x, u)
    def jac(self, x, u): return self.d.jac(x, u)

out = []
for dt, name in ((torch.float32, "f32"), (torch.float64, "f64")):
    B, T, nx, nu = int(os.environ.get("MODES_B", "16384")), 20, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev)
    dyn = AffineDynamics(p.F, p.c)
    cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, device=dev, dtype=dt))
    for mode, d, exit_mode in (("fused/fixed", dyn, "fixed"), ("fused/reference-exit", dyn, "reference"),
                               ("nonlinear-caller/reference-exit", CallableOnly(dyn), "reference")):
        ts = []
        for rep in range(4):
            mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode=exit_mode)
            mpc.reinitialize(p.x0, None)
            mpc.al_iter = 2
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x, u, _ = mpc(p.x0, cost, d, d.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        r = {"dtype": name, "mode": mode, "ms": 1e3 * min(ts), "solves_per_s": B / min(ts), "newton_per_al": list(mpc.last_newton_per_al)}
        print(json.dumps(r)); out.append(r)
json.dump(out, open("gpurun_out/modes.json", "w"), indent=1)
