import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from deq_mpc_corl_amd import synthetic_problem
from deq_mpc_corl_amd.backend import default_backend
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T, nx, nu = 20, 13, 4
be = default_backend()
for dt in (torch.float64, torch.float32):
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device="cuda:0")
    tm = lambda a: a.transpose(0, 1).contiguous()
    Cd, c, F, f = tm(p.Qd), tm(p.q), tm(p.F), tm(p.c)
    for mode in ("fixed", "reference"):
        run = lambda: be.ipm_solve((B, T, nx, nu), Cd, c, F, f, p.x0, p.u_hi, p.u_lo, exit_mode=mode)
        o = run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): o = run()
        torch.cuda.synchronize()
        print(B, dt, mode, f"{(time.perf_counter()-t0)*200:.3f} ms per solve, iters {o['iters']}")
