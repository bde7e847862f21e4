"""Helpers shared by the parity tests: load fixtures, rebuild per-step solver state."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    for k in ("B", "T", "nx", "nu", "al_iter", "nonlinear", "active", "seed", "status",
              "n_steps_recorded", "calls"):
        if k in d and np.ndim(d[k]) == 0:
            d[k] = int(d[k])
    if "dtype" in d:
        d["dtype"] = str(d["dtype"])
    return d


def names(pattern="*"):
    # dyn_*.npz and *casadi* belong to the dynamics-provider tests (tests/test_dynamics_provider.py)
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + ".npz"))
                  if not os.path.basename(p).startswith("dyn_") and "casadi" not in os.path.basename(p)
                  and "stream" not in os.path.basename(p) and not os.path.basename(p).startswith(("ip_", "obs_", "fail_", "se_")))


def step_context(g):
    """For every recorded Newton step s: (al iteration, step in it, z before, lam, rho)."""
    B = g["B"]
    npa = g["newton_per_al"]
    M = g["lam_hist"].shape[-1]
    dt = g["z0"].dtype
    out = []
    s = 0
    for it, cnt in enumerate(npa):
        lam = np.zeros((B, M), dt) if it == 0 else g["lam_hist"][it - 1]
        rho = np.ones((B,), dt) if it == 0 else g["rho_hist"][it - 1].reshape(B)
        for j in range(cnt):
            if s >= g["n_steps_recorded"]:
                return out
            zb = g["z0"] if s == 0 else g["step_z"][s - 1]
            out.append((it, j, zb, lam, rho))
            s += 1
    return out
