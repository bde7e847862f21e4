#!/usr/bin/env python3
"""Calibration (build container only): the reference's qpth.AL_mpc.MPC on CPU vs the
oracle on the same inputs and cores. Writes profiles/cpu_calibration.json."""
import json, os, sys, time
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1"); sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_stubs")); sys.path.insert(1, "/root/reference"); sys.path.insert(2, ROOT)
import numpy as np, torch
torch.set_num_threads(8)
from qpth import AL_mpc, al_utils
from deq_mpc_corl_amd.problems import synthetic_problem, AffineDynamics
from oracle import oracle_py as orc

res = []
for (B, T, nx, nu) in [(128, 5, 2, 1), (64, 20, 13, 4), (128, 20, 13, 4)]:
    for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt)
        dyn = AffineDynamics(p.F, p.c)
        cost = al_utils.QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dt))
        best = 1e9
        for rep in range(3):
            mpc = AL_mpc.MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt)
            mpc.reinitialize(p.x0, None); mpc.al_iter = 2
            t0 = time.perf_counter()
            x, u, _ = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
            best = min(best, time.perf_counter() - t0)
        c = lambda a: a.numpy()
        arrs = [c(a) for a in (p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, p.z0)]
        orc.solve_lin(name, *arrs, al_iter=2, exit_mode="reference")
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 2.0:
            o = orc.solve_lin(name, *arrs, al_iter=2, exit_mode="reference"); reps += 1
        to = (time.perf_counter() - t0) / reps
        err = float(np.abs(o["z"][..., :nx].astype(np.float32) - x.detach().numpy()).max())
        r = {"B": B, "T": T, "nx": nx, "nu": nu, "dtype": name, "reference_s": best, "reference_solves_per_s": B / best,
             "oracle_s": to, "oracle_solves_per_s": B / to, "oracle_over_reference": best / to,
             "max_abs_diff_x": err, "threads": 8}
        print(r); res.append(r)
json.dump({"host": "build container, 8-core Xeon 2.1 GHz, torch 2.10 CPU", "rows": res},
          open(os.path.join(ROOT, "profiles", "cpu_calibration.json"), "w"), indent=1)
