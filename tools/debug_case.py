#!/usr/bin/env python3
"""Debug helper: one linear fixture through the fused kernel (team / quad) against the oracle, decision by decision."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import golden_util as gu
from tests.test_gpu_parity import run_fused, orc
name = sys.argv[1]; variant = sys.argv[2]
g = gu.load(name); dt = g["dtype"]; al = min(g["al_iter"], 4 if dt == "f64" else 2); S = al * 4
o = orc.solve_lin(dt, g["Qd"], g["q"], g["F"], g["c"], g["x0"], g["u_lo"], g["u_hi"], g["z0"], al_iter=al, exit_mode="fixed", trace_steps=S)
h = run_fused(g, dt, al, variant=variant)
for s in range(S):
    phi_o = o["phi"][s]; srt = np.sort(phi_o, axis=0); gap = srt[1] - srt[0]
    margin = np.abs(phi_o.min(0) - o["phi_prev"][s]); scale = np.abs(o["phi_prev"][s]) + 1
    print(s, "k gpu", h["tr"]["k"][s], "orc", o["k"][s], "acc gpu", h["tr"]["accept"][s], "orc", o["accept"][s],
          "gap/scale", gap / scale, "margin/scale", margin / scale, "dmax", np.abs(o["d"][s]).reshape(g["B"], -1).max(1))
print("z err per instance", np.abs(h["z"] - o["z"]).reshape(g["B"], -1).max(1))
z = g["z0"].astype(np.float64).copy()
print("partial sums vs gpu z:")
for s in range(S + 1):
    print(s, np.abs(z - h["z"]).reshape(g["B"], -1).max(1))
    if s < S:
        al_s = np.where(h["tr"]["accept"][s] != 0, 1.0 / (1 << h["tr"]["k"][s]), 0.0)
        z = z + al_s[:, None, None] * h["tr"]["d"][s]
e = np.abs(h["z"] - o["z"])
print("err per stage (instance 0):", np.round(e[0].max(1), 3))
print("err per element (instance 0):", np.round(e[0].max(0), 3))
print("lam err", np.abs(h["lam"] - o["lam"]).max(), "rho", h["rho"], o["rho"], "phi", h["phi"], "rn2", h["rn2"])
print("gpu z[0,0]", h["z"][0, 0]); print("orc z[0,0]", o["z"][0, 0]); print("z0 [0,0]", g["z0"][0, 0])
print("gpu z[0,5]", h["z"][0, 5]); print("orc z[0,5]", o["z"][0, 5])
