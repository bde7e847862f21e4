#!/usr/bin/env python3
"""Golden vectors for the pendulum1l dynamics provider, produced by RUNNING the reference's
CasADi-generated code (compiled from its own sources into oracle/_ref by `make -C oracle ref`).
Writes tests/golden/dyn_pendulum1l.npz (inputs and expected outputs only - data, no source)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dyn_py  # noqa: E402

dyn_py.build()
assert dyn_py.have_ref(), "oracle/_ref/libpendulum1l_casadi.so missing (needs /root/reference)"
rng = np.random.default_rng(7)
K = 512
x = np.stack([rng.uniform(-np.pi, np.pi, K), rng.normal(0, 3.0, K)], 1)
u = rng.normal(0, 2.0, (K, 1))
# a few special points: hanging, upright, zero torque, large velocity
x[:4] = [[0, 0], [np.pi, 0], [0.3, 0], [1.0, 25.0]]
u[:4] = [[0], [0], [0], [-5.0]]
out = {}
for name, h in (("h05", 0.05), ("h01", 0.01)):
    xn, A, B = dyn_py.pendulum1l_ref(x, u, h)
    out.update({f"{name}_h": np.float64(h), f"{name}_xn": xn, f"{name}_A": A, f"{name}_B": B})
np.savez(os.path.join(ROOT, "tests", "golden", "dyn_pendulum1l.npz"), x=x, u=u, **out)
xn2, A2, B2 = dyn_py.pendulum1l(x, u, 0.05)
print("restatement vs reference (h=0.05): max |dxn| %.2e, |dA| %.2e, |dB| %.2e" % (
    np.abs(xn2 - out["h05_xn"]).max(), np.abs(A2 - out["h05_A"]).max(), np.abs(B2 - out["h05_B"]).max()))
