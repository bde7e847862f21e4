"""Host logic of the drop-in MPC class on CPU, with the TEST-ONLY oracle backend
injected: state carry, exit modes, nonlinear-caller loop, adapter trace, autograd."""
import sys
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests import golden_util as gu
from tests.oracle_backend import OracleBackend

TD = {"f64": torch.float64, "f32": torch.float32}


def t(a, dt):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dt)


def make(g, dt, exit_mode="reference", nonlinear=False):
    from deq_mpc_corl_amd import MPC, AffineDynamics, PendulumDynamics
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    dyn = PendulumDynamics() if nonlinear else AffineDynamics(t(g["F"], dt), t(g["c"], dt))
    mpc = MPC(nx, nu, T, u_lower=t(g["u_lo"], dt), u_upper=t(g["u_hi"], dt), n_batch=B, dtype=dt,
              exit_mode=exit_mode, backend=OracleBackend())
    x0 = t(g["x0"], dt)
    mpc.reinitialize(x0, None)
    return mpc, dyn, x0


@pytest.mark.parametrize("name", ["pend_f64_al2", "cart_f64_al2", "pend_active_f64_al6", "pend_f32_al2"])
def test_reference_exit_mode_reproduces_reference(name):
    from deq_mpc_corl_amd import QuadCost
    g = gu.load(name)
    dt = TD[g["dtype"]]
    B, T, nx = g["B"], g["T"], g["nx"]
    mpc, dyn, x0 = make(g, dt)
    mpc.al_iter = g["al_iter"]
    z0 = t(g["z0"], dt)
    cost = QuadCost(torch.diag_embed(t(g["Qd"], dt)), t(g["q"], dt), torch.zeros(B, T, dtype=dt))
    x, u, status = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    assert status is False and x.dtype == torch.float32
    assert list(mpc.last_newton_per_al) == list(g["newton_per_al"])
    tol = 2e-5 if g["dtype"] == "f64" else 5e-3
    assert np.abs(x.numpy() - g["x"]).max() < tol and np.abs(u.numpy() - g["u"]).max() < tol
    assert np.allclose(mpc.rho_prev.numpy(), g["rho_final"])


def test_fixed_mode_always_runs_four_steps():
    from deq_mpc_corl_amd import QuadCost
    g = gu.load("pend_f64_al2")
    dt = torch.float64
    mpc, dyn, x0 = make(g, dt, exit_mode="fixed")
    mpc.al_iter = 2
    z0 = t(g["z0"], dt)
    cost = QuadCost(torch.diag_embed(t(g["Qd"], dt)), t(g["q"], dt), torch.zeros(g["B"], g["T"], dtype=dt))
    mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :2].clone(), u_init=z0[..., 2:].clone())
    assert list(mpc.last_newton_per_al) == [4, 4]


@pytest.mark.parametrize("name", ["pend_nonlin_f64_al4"])
def test_nonlinear_caller_mode(name):
    from deq_mpc_corl_amd import QuadCost
    g = gu.load(name)
    dt = TD[g["dtype"]]
    B, T, nx = g["B"], g["T"], g["nx"]
    mpc, dyn, x0 = make(g, dt, nonlinear=True)
    mpc.al_iter = g["al_iter"]
    z0 = t(g["z0"], dt)
    Qd = t(g["Qd"], dt).requires_grad_(True)
    q = t(g["q"], dt).requires_grad_(True)
    cost = QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dt))
    x, u, _ = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    assert list(mpc.last_newton_per_al) == list(g["newton_per_al"])
    assert np.abs(x.detach().numpy() - g["x"]).max() < 2e-5
    assert np.abs(u.detach().numpy() - g["u"]).max() < 2e-5
    ((x * t(g["bwd_wx"], torch.float32)).sum() + (u * t(g["bwd_wu"], torch.float32)).sum()).backward()
    assert np.abs(q.grad.numpy() - g["bwd_q_grad"]).max() < 1e-5 * np.abs(g["bwd_q_grad"]).max()
    assert np.abs(Qd.grad.numpy() - g["bwd_Qd_grad"]).max() < 1e-5 * np.abs(g["bwd_Qd_grad"]).max()


def test_state_carry_and_rho_growth():
    from deq_mpc_corl_amd import QuadCost
    g = gu.load("cart_carry_f64")
    dt = torch.float64
    B, T, nx = g["B"], g["T"], g["nx"]
    mpc, dyn, x0 = make(g, dt)
    z0 = t(g["z0"], dt)
    mpc.x_init, mpc.u_init = z0[..., :nx].clone(), z0[..., nx:].clone()
    C = torch.diag_embed(t(g["Qd"], dt))
    for i in range(g["calls"]):
        mpc.al_iter = 2
        x, u, _ = mpc(x0, QuadCost(C, t(g["q"][i], dt), torch.zeros(B, T, dtype=dt)), dyn, dyn.jac)
        assert np.abs(x.numpy() - g["x"][i]).max() < 5e-5 * (i + 1)
        assert np.allclose(mpc.rho_prev.numpy(), g["rho"][i])
    assert float(mpc.rho_prev.max()) == 1e6


def test_tracking_mpc_adapter_trace():
    """The adapter row: the call sequence policies.Tracking_MPC makes -
    reinitialize, seed x_init/u_init from the reference trajectory on the first call,
    set al_iter, build p = -(Q o x_ref), call - against the reference's own trace."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
    g = gu.load("cart_tracking_f64")
    dt = torch.float64
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    dyn = AffineDynamics(t(g["F"], dt), t(g["c"], dt))
    ctrl = MPC(nx, nu, T, u_lower=torch.full((nu,), -0.5), u_upper=torch.full((nu,), 0.5),
               exit_unconverged=False, eps=1e-2, n_batch=B, backprop=False, verbose=0,
               u_init=torch.randn(B, T, nu, dtype=dt), solver_type="dense", dtype=dt,
               state_estimator=False, backend=OracleBackend())
    Qdiag = torch.cat([torch.full((nx,), 10.0), torch.full((nu,), 1e-8)]).to(dt)
    Q = torch.diag(Qdiag).repeat(B, T, 1, 1)
    x0 = t(g["x0"], dt)
    u_ref = t(g["u_ref"], dt)
    ctrl.reinitialize(t(g["x_ref"][0], dt), torch.ones(B, T, 1, dtype=dt))
    first = True
    for i in range(3):
        x_ref = t(g["x_ref"][i], dt)
        if first:
            ctrl.x_init, ctrl.u_init = x_ref.detach().clone(), u_ref.detach().clone()
            first = False
        xu_ref = torch.cat([x_ref, u_ref], -1)
        p = -(Q * xu_ref.unsqueeze(-2)).sum(-1)
        f = 0.5 * (xu_ref * (Q * xu_ref.unsqueeze(-2)).sum(-1)).sum(-1)
        ctrl.al_iter = 2
        x, u, status = ctrl(x0, QuadCost(Q, p, f), dyn, dyn.jac, None)
        assert np.abs(x.numpy() - g["x"][i]).max() < 1e-4
        assert np.abs(u.numpy() - g["u"][i]).max() < 1e-4
    assert np.allclose(ctrl.rho_prev.numpy(), g["rho"])
    xu = ctrl.get_xu()
    assert xu.shape == (B, T, nx + nu)


def test_warm_start_initialize_and_stream_mode():
    from deq_mpc_corl_amd import QuadCost
    g = gu.load("cart_f64_al2")
    dt = torch.float64
    B, T, nx = g["B"], g["T"], g["nx"]
    mpc, dyn, x0 = make(g, dt)
    z0 = t(g["z0"], dt)
    cost = QuadCost(torch.diag_embed(t(g["Qd"], dt)), t(g["q"], dt), torch.zeros(B, T, dtype=dt))
    mpc.al_iter = 2
    x, u, _ = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    lam_before = mpc.lamda_prev.clone()
    assert float(lam_before.abs().max()) > 0
    mpc.warm_start_initialize(x.double(), u.double(), SimpleNamespace(rho_init_max=50.0))
    assert mpc.warm_starting is True
    assert float(mpc.lamda_prev.abs().max()) == 0.0          # shifted then zeroed (AL_mpc.py:589)
    assert float(mpc.rho_prev.max()) == 50.0                  # clamped (AL_mpc.py:590)
    mpc.al_iter = 10
    x2, u2, status = mpc(x0, cost, dyn, dyn.jac)
    # rho: 50 -> ... stops once above rho_max = 1e8 and reports status True (AL_mpc.py:412-421)
    assert status is True
    assert float(mpc.rho_prev.max()) > 1e8
    assert torch.isfinite(x2).all()


def test_unsupported_configurations_raise():
    from deq_mpc_corl_amd import MPC
    lo, hi = torch.tensor([-1.0]), torch.tensor([1.0])
    with pytest.raises(NotImplementedError):
        MPC(2, 1, 5, u_lower=lo, u_upper=hi, add_goal_constraint=True)
    assert MPC(2, 1, 5, u_lower=lo, u_upper=hi, state_estimator=True).neq == 2 * 4   # tests/test_state_estimator_golden.py
    with pytest.raises(ValueError):
        MPC(2, 1, 5)
    with pytest.raises(ValueError):
        MPC(2, 1, 5, u_lower=lo, u_upper=hi, exit_mode="nope")
