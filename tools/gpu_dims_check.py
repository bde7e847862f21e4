#!/usr/bin/env python3
"""Fused kernel vs oracle on synthetic problems for every compiled (nx,nu)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle_py as orc
from deq_mpc_corl_amd import synthetic_problem
from deq_mpc_corl_amd.backend import default_backend
be = default_backend()
dev = "cuda:0"
for (nx, nu) in [(2,1),(4,1),(4,2),(6,2),(8,2),(10,3),(12,4),(13,4),(14,4)]:
    for dt, name in ((torch.float32,"f32"),(torch.float64,"f64")):
        B, T = 9, 7
        p = synthetic_problem(B, T, nx, nu, seed=5, dtype=dt, device=dev)
        M = T*nx + 2*T*nu
        z = p.z0.clone(); lam = torch.zeros(B, M, dtype=dt, device=dev); rho = torch.ones(B, dtype=dt, device=dev)
        phi = torch.zeros(B, dtype=dt, device=dev); rn2 = torch.zeros_like(phi)
        info = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.uint8, device=dev)
        be.solve_lin((B,T,nx,nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2, info, st,
                     al_iter=2, max_newton=4, n_ls=20, flags=3)
        torch.cuda.synchronize()
        c = lambda a: a.cpu().numpy()
        o = orc.solve_lin(name, c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0), al_iter=2, exit_mode="fixed")
        print(nx, nu, name, "max err z", float(np.abs(c(z)-o["z"]).max()), "lam", float(np.abs(c(lam)-o["lam"]).max()), "info", int(info.abs().sum()))
