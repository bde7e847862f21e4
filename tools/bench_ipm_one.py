#!/usr/bin/env python3
"""One interior-point kernel variant, a few launches (for rocprofv3): python tools/bench_ipm_one.py [variant] [f64|f32] [B] [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deq_mpc_corl_amd import synthetic_problem
from deq_mpc_corl_amd.backend import default_backend

variant = sys.argv[1] if len(sys.argv) > 1 else "resident"
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
T, nx, nu = 20, 13, 4
be = default_backend()
p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device="cuda:0")
tm = lambda a: a.transpose(0, 1).contiguous()
Cd, c, F, f = tm(p.Qd), tm(p.q), tm(p.F), tm(p.c)
run = lambda: be.ipm_solve((B, T, nx, nu), Cd, c, F, f, p.x0, p.u_hi, p.u_lo, exit_mode="fixed", variant=variant)
o = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    o = run()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / reps
print(json.dumps({"variant": variant, "dtype": str(dt), "B": B, "ms": 1e3 * el, "qps": B / el,
                  "max_best_resid": float(o["resid"].max()), "info_nonzero": int((o["info"] != 0).sum())}))
