#!/usr/bin/env python3
"""GPU debugging aid: one Newton step of the fused kernel vs the oracle (g, factor, d)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle_py as orc
from tests import golden_util as gu
from tests.test_gpu_parity import run_fused

name = sys.argv[1] if len(sys.argv) > 1 else "quad12_f32_al2"
g = gu.load(name)
dt = g["dtype"]
B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]; n = nx + nu
h = run_fused(g, dt, 1, max_newton=1, flags=1, factor=True)
o = orc.solve_lin(dt, g["Qd"], g["q"], g["F"], g["c"], g["x0"], g["u_lo"], g["u_hi"], g["z0"],
                  al_iter=1, max_newton=1, exit_mode="fixed", trace_steps=1, save_factor=True)
print("g err", np.abs(h["tr"]["g"][0] - o["g"][0]).max())
X = h["factor"].cpu().numpy().reshape(B, T, -1)
L = o["L"]
for t in range(T):
    Xo = np.linalg.inv(L[0, t].astype(np.float64)).T
    Xh = np.zeros((n, n))
    off = 0
    for i in range(n):
        Xh[i, i:] = X[0, t, off:off + n - i]; off += n - i
    e = np.abs(Xh - np.triu(Xo)).max() / np.abs(Xo).max()
    if e > 1e-3 or t < 2:
        print("t", t, "X relerr", e)
        if e > 1e-3:
            bad = np.argwhere(np.abs(Xh - np.triu(Xo)) > 1e-3 * np.abs(Xo).max())
            print(" bad entries", bad[:10].tolist())
            print(" Xh row0", Xh[0, :6], " Xo row0", Xo[0, :6])
            break
d_h, d_o = h["tr"]["d"][0], o["d"][0]
print("d err per stage", [float(np.abs(d_h[0, t] - d_o[0, t]).max()) for t in range(T)])
