#!/usr/bin/env python3
"""Reference exit rule, two routes: inside ONE cooperative launch (ALQP_EXIT_IN_KERNEL) or a launch per Newton step with
alqp_exit_test in between. ms per MPC call and the Newton-step counts. Usage: python tools/bench_exit_routes.py [B] [f32|f64]"""
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
cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dt, device=dev))
dyn = AffineDynamics(p.F, p.c)
for in_kernel in (True, False, True, False):
    mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode="reference", exit_in_kernel=in_kernel)

    def call():
        mpc.reinitialize(p.x0, None)
        mpc.al_iter = 2
        return mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())

    call(); call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    print(f"B={B} {dt} exit_in_kernel={in_kernel}: {(time.perf_counter() - t0) * 100:.3f} ms per call, Newton steps {mpc.last_newton_per_al}")
