#!/usr/bin/env python3
"""ms of the implicit-function backward pass (NewtonAL.backward, al_utils.py:578-615) behind MPC.__call__.
Usage: python tools/bench_backward.py [B] [f32|f64]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
T, nx, nu = 20, 13, 4
dev = "cuda:0"
p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev)
dyn = AffineDynamics(p.F, p.c)
w = torch.randn(B, T, nx, device=dev)
for mode in ("fixed", "reference"):
    mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode=mode)
    fw = bw = 0.0
    for it in range(7):
        Qd = p.Qd.clone().requires_grad_(True)
        q = p.q.clone().requires_grad_(True)
        cost = QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dt, device=dev))
        mpc.reinitialize(p.x0, None)
        mpc.al_iter = 2
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, u, _ = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (x * w).sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 2:
            fw += t1 - t0; bw += t2 - t1
    print(f"B={B} {dt} exit_mode={mode}: forward {fw / 5 * 1e3:.3f} ms, backward {bw / 5 * 1e3:.3f} ms (incl. autograd of diag_embed etc.)")
