#!/usr/bin/env python3
"""Per-phase cycle breakdown of the TEAM kernel k_solve_lin (debug build from tools/phase_timing.sh) at the reference's
batch size. Each stamp drains the memory queues, so the run is slower than the product kernel; the split is what matters.

    bash tools/phase_timing.sh && gpurun -- python tools/team_timing.py [B] [f32|f64]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["fwd: stage inputs, residual, gradient", "fwd: SYRK", "fwd: panel", "fwd: stage results", "backward sweep",
         "line-search merits (20 candidates)", "pick + apply", "other (init, residual pre-pass, dual update, outputs)"]


def main():
    import torch
    from deq_mpc_corl_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "deq-mpc-corl_amd", "csrc", "build", "libmi_alqp_timing.so")
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    lib = _lib.load()
    lib.alqp_debug_team_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
    T, nx, nu = 20, 13, 4
    dev = "cuda:0"
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev)
    M = T * nx + 2 * T * nu
    out = (C.c_ulonglong * 8)()
    res = {}
    for rep in range(3):
        z = p.z0.clone()
        lam = torch.zeros(B, M, dtype=dt, device=dev)
        rho = torch.ones(B, dtype=dt, device=dev)
        phi = torch.zeros(B, dtype=dt, device=dev)
        rn2 = torch.zeros(B, dtype=dt, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        lib.alqp_debug_team_cycles(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2,
                     info, status, al_iter=2, max_newton=4, n_ls=20, flags=3, variant="team")
        e1.record()
        torch.cuda.synchronize()
        lib.alqp_debug_team_cycles(out, 0)
        cyc = [int(v) for v in out]
        tot = sum(cyc)
        res = {"B": B, "dtype": str(dt), "kernel_ms": e0.elapsed_time(e1), "waves": B, "newton_steps": 8, "stages": T,
               "phases": {n: {"cycles_per_wave": c / B, "frac": c / tot} for n, c in zip(NAMES, cyc)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
