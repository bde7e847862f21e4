#!/usr/bin/env python3
"""Golden vectors for the pendulum1l, cartpole1l and cartpole2l dynamics providers, produced by RUNNING the reference's
CasADi-generated code (compiled from its own sources into oracle/_ref by `make -C oracle ref`).
Writes tests/golden/dyn_pendulum1l.npz and dyn_cartpole1l.npz (inputs and expected outputs only - data, no source)."""
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
# ---- cartpole1l
assert dyn_py.have_ref_cartpole(), "oracle/_ref/libcartpole1l_casadi.so missing"
Kc = 512
xc = np.stack([rng.normal(0, 1, Kc), rng.uniform(-np.pi, np.pi, Kc), rng.normal(0, 2, Kc), rng.normal(0, 4, Kc)], 1)
tc = np.stack([rng.normal(0, 8, Kc), np.zeros(Kc)], 1)      # the environments drive tau[0] only ...
tc[Kc // 2:, 1] = rng.normal(0, 2, Kc - Kc // 2)             # ... the generated code takes both
xc[:3] = [[0, 0, 0, 0], [0, np.pi, 0, 0], [0.5, 0.2, -1.0, 6.0]]
tc[:3] = [[0, 0], [0, 0], [20.0, 0]]
outc = {}
for name, h in (("h05", 0.05), ("h01", 0.01)):
    xn, J = dyn_py.cartpole1l_ref(xc, tc, h)
    outc.update({f"{name}_h": np.float64(h), f"{name}_xn": xn, f"{name}_J": J})
np.savez(os.path.join(ROOT, "tests", "golden", "dyn_cartpole1l.npz"), x=xc, tau=tc, **outc)
xn3, J3 = dyn_py.cartpole1l(xc, tc, 0.05)
print("cartpole1l restatement vs reference (h=0.05): max |dxn| %.2e, |dJ| %.2e" % (
    np.abs(xn3 - outc["h05_xn"]).max(), np.abs(J3 - outc["h05_J"]).max()))
# ---- cartpole1l_v2 (same model, lighter cart and pole; a package the reference ships but does not import)
assert dyn_py.have_ref_cartpole_v2(), "oracle/_ref/libcartpole1l_v2_casadi.so missing"
outv = {}
for name, h in (("h05", 0.05), ("h01", 0.01)):
    xn, J = dyn_py.cartpole1l_v2_ref(xc, tc, h)
    outv.update({f"{name}_h": np.float64(h), f"{name}_xn": xn, f"{name}_J": J})
np.savez(os.path.join(ROOT, "tests", "golden", "dyn_cartpole1l_v2.npz"), x=xc, tau=tc, **outv)
xn5, J5 = dyn_py.cartpole1l_v2(xc, tc, 0.05)
print("cartpole1l_v2 restatement vs reference (h=0.05): max |dxn| %.2e, |dJ| %.2e" % (
    np.abs(xn5 - outv["h05_xn"]).max(), np.abs(J5 - outv["h05_J"]).max()))
# ---- cartpole2l
assert dyn_py.have_ref_cartpole2(), "oracle/_ref/libcartpole2l_casadi.so missing"
K2 = 384
x2 = np.concatenate([rng.normal(0, 1, (K2, 1)), rng.uniform(-np.pi, np.pi, (K2, 2)), rng.normal(0, 2, (K2, 3))], 1)
t2 = np.concatenate([rng.normal(0, 8, (K2, 1)), np.zeros((K2, 2))], 1)   # environments: tau = (u, 0, 0) ...
t2[K2 // 2:, 1:] = rng.normal(0, 2, (K2 - K2 // 2, 2))                    # ... the generated code takes all three
x2[:3] = [[0, 0, 0, 0, 0, 0], [0, np.pi, 0, 0, 0, 0], [0.2, 0.3, -0.5, 1.0, -2.0, 3.0]]
t2[:3] = [[0, 0, 0], [0, 0, 0], [15.0, 0, 0]]
out2 = {}
for name, h in (("h05", 0.05), ("h01", 0.01)):
    xn, J = dyn_py.cartpole2l_ref(x2, t2, h)
    out2.update({f"{name}_h": np.float64(h), f"{name}_xn": xn, f"{name}_J": J})
np.savez(os.path.join(ROOT, "tests", "golden", "dyn_cartpole2l.npz"), x=x2, tau=t2, **out2)
xn4, J4 = dyn_py.cartpole2l(x2, t2, 0.05)
print("cartpole2l restatement vs reference (h=0.05): max |dxn| %.2e, |dJ| %.2e" % (
    np.abs(xn4 - out2["h05_xn"]).max(), np.abs(J4 - out2["h05_J"]).max()))
xn2, A2, B2 = dyn_py.pendulum1l(x, u, 0.05)
print("restatement vs reference (h=0.05): max |dxn| %.2e, |dA| %.2e, |dB| %.2e" % (
    np.abs(xn2 - out["h05_xn"]).max(), np.abs(A2 - out["h05_A"]).max(), np.abs(B2 - out["h05_B"]).max()))
