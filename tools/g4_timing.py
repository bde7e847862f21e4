#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_ipm_g4 (debug build from tools/g4_timing.sh), fp64 (20,13,4), exit mode "fixed".
    bash tools/g4_timing.sh && gpurun -- python tools/g4_timing.py [B]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["problem load", "residuals", "factor: Dt, Pu", "factor: blocks", "apply: right-hand side", "inward sweep",
         "outward sweep", "apply: outputs", "K product", "step lengths, updates", "workspace, outputs",
         "block: loads + F P F'", "block: Z", "block: Z D^-1 Z'", "block: L D L'", "block: inverse + store"]


def main():
    import torch
    from deq_mpc_corl_amd import _lib
    _lib.LIB_PATH = os.environ.get("G4_TIMING_LIB") or os.path.join(ROOT, "deq-mpc-corl_amd", "csrc", "build", "libmi_alqp_g4timing.so")
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    lib = _lib.load()
    lib.alqp_g4_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    T, nx, nu = 20, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=torch.float64, device="cuda:0")
    tm = lambda a: a.transpose(0, 1).contiguous()
    args = ((B, T, nx, nu), tm(p.Qd), tm(p.q), tm(p.F), tm(p.c), p.x0, p.u_hi, p.u_lo)
    be.ipm_solve(*args, exit_mode="fixed", variant="resident")
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib.alqp_g4_debug_phase_cycles(out, 1)
    be.ipm_solve(*args, exit_mode="fixed", variant="resident")
    torch.cuda.synchronize()
    lib.alqp_g4_debug_phase_cycles(out, 0)
    tot = sum(out[:16])
    res = {"B": B, "cycles_per_qp": tot / B, "phases": {n: {"cycles_per_qp_iteration": out[i] / B / 20, "share": out[i] / tot}
                                                       for i, n in enumerate(NAMES)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
