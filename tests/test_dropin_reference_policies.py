"""Drop-in proof, build container only (skipped where /root/reference is absent, e.g. on
the GPU box): the reference's UNCHANGED deqmpc/policies.py::Tracking_MPC drives our MPC
class after the module shadowing shown in INTEGRATION.md section 1, and reproduces the
trace the reference's own solver produced (tests/golden/cart_tracking_f64.npz).
The arithmetic runs on the TEST-ONLY oracle backend here (no GPU in this container)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests import golden_util as gu

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "qpth")), reason="reference tree not present")


def test_unchanged_tracking_mpc_runs_on_our_solver(monkeypatch):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.dont_write_bytecode = True
    for p in (os.path.join(root, "tools", "_stubs"), REF, os.path.join(REF, "deqmpc")):
        if p not in sys.path:
            monkeypatch.syspath_prepend(p)
    saved = {k: v for k, v in sys.modules.items() if k == "qpth" or k.startswith("qpth.") or k == "policies"}
    for k in saved:
        del sys.modules[k]
    try:
        import qpth  # the reference's package
        import deq_mpc_corl_amd.backend as backend_mod
        import deq_mpc_corl_amd.qpth.AL_mpc as mi_al_mpc
        import deq_mpc_corl_amd.qpth.al_utils as mi_al_utils
        from tests.oracle_backend import OracleBackend

        monkeypatch.setattr(backend_mod, "default_backend", lambda: OracleBackend())
        monkeypatch.setitem(sys.modules, "qpth.AL_mpc", mi_al_mpc)
        monkeypatch.setattr(qpth, "AL_mpc", mi_al_mpc)
        monkeypatch.setattr(qpth.al_utils, "QuadCost", mi_al_utils.QuadCost)
        import policies  # unchanged reference file
        assert policies.al_mpc is mi_al_mpc

        from deq_mpc_corl_amd.problems import AffineDynamics
        g = gu.load("cart_tracking_f64")
        dt = torch.float64
        B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
        dyn = AffineDynamics(t(g["F"]), t(g["c"]))
        env = SimpleNamespace(nu=nu, nx=nx, nq=nx // 2, dt=0.05, dynamics=dyn, dynamics_derivatives=dyn.jac,
                              action_space=SimpleNamespace(high=np.full(nu, 0.5), low=np.full(nu, -0.5)))
        args = SimpleNamespace(T=T, device="cpu", qp_iter=1, eps=1e-2, warm_start=False, bsz=B,
                               Q=torch.tensor([10.0] * nx), R=torch.tensor([1e-8] * nu), dtype="double",
                               solver_type="al", env="synthetic")
        tm = policies.Tracking_MPC(args, env)
        assert isinstance(tm.ctrl, mi_al_mpc.MPC)
        u_ref = t(g["u_ref"])
        tm.reinitialize(t(g["x_ref"][0]), torch.ones(B, T, 1, dtype=dt))
        for i in range(3):
            x, u, status = tm(t(g["x0"]), None, t(g["x_ref"][i]), u_ref, al_iters=2)
            assert np.abs(x.numpy() - g["x"][i]).max() < 1e-4
            assert np.abs(u.numpy() - g["u"][i]).max() < 1e-4
        assert np.allclose(tm.ctrl.rho_prev.numpy(), g["rho"])
    finally:
        for k in [k for k in sys.modules if k == "qpth" or k.startswith("qpth.") or k == "policies"]:
            del sys.modules[k]
        sys.modules.update(saved)
