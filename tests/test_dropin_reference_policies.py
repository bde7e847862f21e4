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


def test_policies_imports_resolve_against_the_shim_package_alone(monkeypatch):
    """`policies.py:5-8` (`qpth.qp_wrapper`, `qpth.AL_mpc`, `qpth.AL_mpc_custom.Obstacle_MPC`,
    `qpth.al_utils`) bound to the shim package only - the reference's own `qpth` is NOT imported - and the
    unchanged `Tracking_MPC` replays the streaming trace (reinitialize, 2 calls, warm_start_initialize,
    3 stream calls: tests/golden/cart_tracking_stream_f64.npz). Also: `--solver_type al` with an
    'obstacles' env constructs our Obstacle_MPC, `--solver_type ip` our qp_wrapper.MPC."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.dont_write_bytecode = True
    for p in (os.path.join(root, "tools", "_stubs"), os.path.join(REF, "deqmpc")):
        monkeypatch.syspath_prepend(p)
    saved = {k: v for k, v in sys.modules.items() if k == "qpth" or k.startswith("qpth.") or k == "policies"}
    for k in saved:
        del sys.modules[k]
    try:
        import deq_mpc_corl_amd.backend as backend_mod
        import deq_mpc_corl_amd.qpth as mi_qpth
        from tests.oracle_backend import OracleBackend
        monkeypatch.setattr(backend_mod, "default_backend", lambda: OracleBackend())
        monkeypatch.setitem(sys.modules, "qpth", mi_qpth)
        for m in ("AL_mpc", "al_utils", "AL_mpc_custom", "qp_wrapper"):
            monkeypatch.setitem(sys.modules, "qpth." + m, getattr(mi_qpth, m))
        import policies  # unchanged reference file
        assert policies.al_mpc is mi_qpth.AL_mpc and policies.ip_mpc is mi_qpth.qp_wrapper
        assert policies.Obstacle_MPC is mi_qpth.AL_mpc_custom.Obstacle_MPC

        from deq_mpc_corl_amd.problems import AffineDynamics
        g = gu.load("cart_tracking_stream_f64")
        dt = torch.float64
        B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
        dyn = AffineDynamics(t(g["F"]), t(g["c"]))
        env = SimpleNamespace(nu=nu, nx=nx, nq=nx // 2, dt=0.05, dynamics=dyn, dynamics_derivatives=dyn.jac,
                              action_space=SimpleNamespace(high=np.full(nu, 0.5), low=np.full(nu, -0.5)),
                              obstacle_radius=0.3, obstacle_positions=torch.randn(40, 3, dtype=dt))
        args = SimpleNamespace(T=T, device="cpu", qp_iter=1, eps=1e-2, warm_start=True, bsz=B,
                               Q=torch.tensor([10.0] * nx), R=torch.tensor([1e-8] * nu), dtype="double",
                               solver_type="al", env="synthetic", rho_init_max=float(g["rho_init_max"]))
        tm = policies.Tracking_MPC(args, env)
        assert type(tm.ctrl) is mi_qpth.AL_mpc.MPC
        u_ref = t(g["u_ref"])
        tm.reinitialize(t(g["x_ref"][0]), torch.ones(B, T, 1, dtype=dt))
        warmed = False
        for i, ph in enumerate(g["phase"].tolist()):
            if ph == 1 and not warmed:
                tm.warm_start_initialize(t(g["x_ref_warm"]), u_ref)
                warmed = True
            x, u, status = tm(t(g["x0"]), None, t(g["x_ref"][i]), u_ref, al_iters=2)
            assert status is bool(g["status"][i])
            assert np.abs(x.numpy() - g["x"][i]).max() < 1e-4 and np.abs(u.numpy() - g["u"][i]).max() < 1e-4
        assert np.array_equal(tm.ctrl.rho_prev.numpy(), g["rho"][-1])
        # the other two constructor branches of Tracking_MPC (policies.py:1179-1234)
        args.env = "flyingcartpole_obstacles"
        assert type(policies.Tracking_MPC(args, env).ctrl) is mi_qpth.AL_mpc_custom.Obstacle_MPC
        args.env, args.solver_type = "synthetic", "ip"
        assert type(policies.Tracking_MPC(args, env).ctrl) is mi_qpth.qp_wrapper.MPC
    finally:
        for k in [k for k in sys.modules if k == "qpth" or k.startswith("qpth.") or k == "policies"]:
            del sys.modules[k]
        sys.modules.update(saved)
