#!/usr/bin/env python3
"""QPs/s of the interior-point kernel (exit mode "fixed": 20 iterations in one launch).
Usage (GPU box): python tools/bench_ipm.py [B] [T] [nx] [nu]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deq_mpc_corl_amd import synthetic_problem
from deq_mpc_corl_amd.backend import default_backend

B, T, nx, nu = [int(a) for a in (sys.argv[1:5] + [8192, 20, 13, 4][len(sys.argv) - 1:])][:4]
be = default_backend()
out = {"B": B, "T": T, "nx": nx, "nu": nu}
for mode_name in ("resident", "generic_lds", "generic_ws"):   # AlqpIpmParams.variant ("auto" = resident at these sizes)
  for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
    name = f"{name}_{mode_name}"
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device="cuda:0")
    tm = lambda a: a.transpose(0, 1).contiguous()
    Cd, c, F, f = tm(p.Qd), tm(p.q), tm(p.F), tm(p.c)
    run = lambda: be.ipm_solve((B, T, nx, nu), Cd, c, F, f, p.x0, p.u_hi, p.u_lo, exit_mode="fixed", variant=mode_name)
    o = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 3
    for _ in range(k):
        o = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / k
    out[name] = {"ms": 1e3 * el, "qps": B / el, "max_best_resid": float(o["resid"].max()), "info_nonzero": int((o["info"] != 0).sum())}
print(json.dumps(out))
