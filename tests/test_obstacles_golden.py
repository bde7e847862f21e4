"""Obstacle rows against fixtures made by RUNNING the reference's
`qpth.AL_mpc_custom.Obstacle_MPC` (tools/gen_golden_obs.py).

  * oracle level (not gpu): gradient, Hessian band, Newton direction and the 20 merits of recorded
    Newton steps, with the obstacle rows switched on in oracle/alqp_oracle_impl.h;
  * drop-in level: `deq_mpc_corl_amd.qpth.AL_mpc_custom.Obstacle_MPC` - nearest-sphere selection in
    reinitialize / warm_start_initialize, Newton-step counts, per-AL-iteration lamda (incl. the obstacle
    rows) and rho, x, u, gradients - on the CPU with the TEST-ONLY oracle backend and on the MI355X through
    alqp_newton_step_obs / alqp_merit_obs / alqp_dual_update_obs (`-m gpu`).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests import golden_util as gu

TD = {"f64": torch.float64, "f32": torch.float32}
OBS = ["obs_cart_f64_al2", "obs_cart_f64_al4", "obs_fcp14_f64_al3", "obs_quad13_f64_al2", "obs_cart_f32_al2"]


def _ctx(g):
    """(al iteration, z before, lam, rho) for recorded step s; lam of an AL iteration > 0 from lam_hist."""
    B, npa = g["B"], g["newton_per_al"]
    out, s = [], 0
    for it, cnt in enumerate(npa):
        lam = np.zeros_like(g["lam_final"]) if it == 0 else g["lam_hist"][it - 1]
        rho = np.ones(B, g["z0"].dtype) if it == 0 else g["rho_hist"][it - 1].reshape(B)
        for _ in range(cnt):
            out.append((it, g["z0"] if s == 0 else g["step_z"][s - 1], lam, rho))
            s += 1
    return out


@pytest.mark.parametrize("name", [n for n in OBS if "f64" in n])
def test_oracle_obstacle_terms_vs_reference(name):
    from oracle import oracle_py as orc
    g = gu.load(name)
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    n = nx + nu
    ctx = _ctx(g)
    # tolerances: the reference builds the obstacle Jacobian in a float32 buffer (`jac_obs_pad =
    # torch.zeros(...)` without dtype, al_utils.py:381), so its fp64 gradient / Hessian carry a 6e-8
    # relative rounding in the position rows; the oracle and the kernels keep the working precision
    # (and its radius is `torch.tensor(radius)`, a float32, upcast: AL_mpc_custom.py:58-59 - reproduced)
    with orc.obstacles("f64", g["obs_pos"], float(np.float32(g["radius"]))):
        for s in (0, 3, len(ctx) - 1):
            it, zb, lam, rho = ctx[s]
            xn = np.einsum("btij,btj->bti", g["F"], zb[:, :-1]) + g["c"]
            gr, Hd, Hs = orc.grad_hess("f64", zb, xn, g["F"], g["x0"], lam, rho, g["Qd"], g["q"], g["u_lo"], g["u_hi"])
            assert np.abs(gr - g["step_g"][s].reshape(B, T, n)).max() < 2e-7 * max(1.0, np.abs(g["step_g"][s]).max())
            if s in g["H_step_index"].tolist():
                i = g["H_step_index"].tolist().index(s)
                assert np.abs(Hd - g["H_diag"][i]).max() < 2e-7 * max(1.0, np.abs(g["H_diag"][i]).max())
            d, info = orc.newton_dir("f64", gr, Hd, Hs, nx)
            assert np.abs(d - g["step_d"][s]).max() < 2e-6 * max(1.0, np.abs(g["step_d"][s]).max())
            for k in (0, 5, 19):
                zc = zb + 2.0 ** -k * g["step_d"][s]
                xc = np.einsum("btij,btj->bti", g["F"], zc[:, :-1]) + g["c"]
                phi, _ = orc.merit("f64", zc, xc, g["x0"], lam, rho, g["Qd"], g["q"], g["u_lo"], g["u_hi"])
                assert np.abs(phi - g["step_phi"][s][k]).max() < 1e-9 * max(1.0, np.abs(g["step_phi"][s][k]).max())


class _Dyn:
    """Callable-only affine dynamics (no F / f attributes), like the torch-coded environments."""

    def __init__(self, F, c):
        from deq_mpc_corl_amd import AffineDynamics
        self._d = AffineDynamics(F, c)

    def __call__(self, x, u):
        return self._d(x, u)

    def jac(self, x, u):
        return self._d.jac(x, u)


def _replay(name, backend, dev):
    from deq_mpc_corl_amd import QuadCost
    from deq_mpc_corl_amd.qpth.AL_mpc_custom import Obstacle_MPC
    g = gu.load(name)
    dt = TD[g["dtype"]]
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    tt = lambda a, d=dt: torch.as_tensor(np.ascontiguousarray(a)).to(d).to(dev)
    env = SimpleNamespace(obstacle_radius=float(g["radius"]), obstacle_positions=tt(g["centres"]))
    dyn = _Dyn(tt(g["F"]), tt(g["c"]))
    mpc = Obstacle_MPC(nx, nu, T, u_lower=tt(g["u_lo"]), u_upper=tt(g["u_hi"]), n_batch=B, dtype=dt, env=env,
                       backend=backend)
    mpc.reinitialize(tt(g["x_ref"]), None)
    assert np.array_equal(mpc.obstacles[0].cpu().numpy(), g["obs_pos"])      # the same 4 nearest spheres
    mpc.al_iter = g["al_iter"]
    Qd, q = tt(g["Qd"]), tt(g["q"])
    with_grad = "bwd_q_grad" in g
    if with_grad:
        Qd.requires_grad_(True)
        q.requires_grad_(True)
    z0 = tt(g["z0"])
    x, u, st = mpc(tt(g["x0"]), QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dt, device=dev)), dyn, dyn.jac,
                   x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    f64 = g["dtype"] == "f64"
    assert st is False
    assert list(mpc.last_newton_per_al) == g["newton_per_al"].tolist()
    tol = 2e-5 if f64 else 5e-3
    c = lambda a: a.detach().cpu().numpy()
    assert np.abs(c(x) - g["x"]).max() < tol and np.abs(c(u) - g["u"]).max() < tol
    assert np.array_equal(c(mpc.rho_prev), g["rho_final"])
    assert mpc.lamda_prev.shape[1] == T * nx + T * (2 * nu + 4)
    if f64:
        assert np.abs(c(mpc.lamda_prev) - g["lam_final"]).max() < 1e-6 * max(1.0, np.abs(g["lam_final"]).max())
    if with_grad:
        ((x * tt(g["bwd_wx"], torch.float32)).sum() + (u * tt(g["bwd_wu"], torch.float32)).sum()).backward()
        assert np.abs(c(q.grad) - g["bwd_q_grad"]).max() < 1e-5 * np.abs(g["bwd_q_grad"]).max()
        assert np.abs(c(Qd.grad) - g["bwd_Qd_grad"]).max() < 1e-5 * np.abs(g["bwd_Qd_grad"]).max()
    if "x_stream" in g:
        mpc.warm_start_initialize(tt(g["x_warm"], torch.float32), tt(g["u_warm"], torch.float32),
                                  SimpleNamespace(rho_init_max=1e3))
        assert np.array_equal(mpc.obstacles[0].cpu().numpy(), g["obs_pos_warm"])
        mpc.al_iter = 2
        x2, u2, st2 = mpc(tt(g["x0"]), QuadCost(torch.diag_embed(tt(g["Qd"])), tt(g["q"]),
                                                torch.zeros(B, T, dtype=dt, device=dev)), dyn, dyn.jac)
        assert st2 is bool(int(g["status_stream"]))
        assert np.abs(c(x2) - g["x_stream"]).max() < tol and np.abs(c(u2) - g["u_stream"]).max() < tol
        assert np.array_equal(c(mpc.rho_prev), g["rho_stream"])
        assert np.abs(c(mpc.lamda_prev) - g["lam_stream"]).max() < 1e-5 * max(1.0, np.abs(g["lam_stream"]).max())


@pytest.mark.parametrize("name", OBS)
def test_obstacle_mpc_host_logic_cpu(name):
    from tests.oracle_backend import OracleBackend
    _replay(name, OracleBackend(), "cpu")


def _hip_backend(route):
    """route "quad": every Newton direction through the quad step kernel (alqp_newton_step_ws_obs: what the class takes by
    itself from B = 4096 on), the factor in its workspace records for backward_ws; "team": the small-batch default."""
    from deq_mpc_corl_amd.backend import HipBackend
    be = HipBackend()
    if route == "quad":
        be.QUAD_MIN_BATCH = 1
    return be


@pytest.mark.gpu
@pytest.mark.parametrize("name", OBS)
@pytest.mark.parametrize("route", ["team", "quad"])
def test_obstacle_mpc_hip(name, route):
    """The reference-generated obstacle fixtures (Newton counts, lamda incl. the obstacle rows, rho, x, u, gradients, the
    streaming call) on both step kernels."""
    be = _hip_backend(route)
    _replay(name, be, "cuda:0")
    assert be.last_step_kernel == ("k_newton_step_quad" if route == "quad" else "k_newton_step")


@pytest.mark.gpu
def test_obstacle_kernels_reject_bad_arguments():
    """nx < 3 (no position to constrain) and malformed centre tensors fail loudly."""
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    dt, dev = torch.float64, "cuda:0"
    B, T, nx, nu = 4, 5, 2, 1
    z = torch.zeros(B, T, 3, dtype=dt, device=dev)
    lam = torch.zeros(B, T * nx + T * (2 * nu + 4), dtype=dt, device=dev)
    one = torch.ones(B, dtype=dt, device=dev)
    lo, hi = -torch.ones(nu, dtype=dt, device=dev), torch.ones(nu, dtype=dt, device=dev)
    pos = torch.zeros(B, T, 4, 3, dtype=dt, device=dev)
    with pytest.raises(RuntimeError):
        be.merit((B, T, nx, nu), 1, z, torch.zeros(B, T - 1, nx, dtype=dt, device=dev), torch.zeros(B, nx, dtype=dt, device=dev),
                 lam, one, z.clone(), z.clone(), lo, hi, 0, 0, one.clone(), obs=(pos, 0.2))
    with pytest.raises(ValueError):
        be.merit((B, T, 8, 2), 1, z, z, z, lam, one, z, z, lo, hi, 0, 0, one.clone(), obs=(pos[:, :, :, :2].contiguous(), 0.2))


def test_obstacle_mpc_refuses_to_solve_before_reinitialize():
    """The nearest-sphere table is built by reinitialize() (AL_mpc_custom.py:104-109; the reference dies with an
    AttributeError without it). The drop-in must not fall through to the plain kernels (multiplier stride M on a
    [B, M + 4T] lamda): it raises."""
    from tests.oracle_backend import OracleBackend
    from deq_mpc_corl_amd import QuadCost
    from deq_mpc_corl_amd.qpth.AL_mpc_custom import Obstacle_MPC
    dt = torch.float64
    B, T, nx, nu = 2, 4, 3, 1
    env = SimpleNamespace(obstacle_radius=0.2, obstacle_positions=torch.randn(40, 3, dtype=dt))
    mpc = Obstacle_MPC(nx, nu, T, u_lower=torch.tensor([-1.0], dtype=dt), u_upper=torch.tensor([1.0], dtype=dt), n_batch=B,
                       dtype=dt, env=env, backend=OracleBackend())
    assert mpc._has_extra_rows()                       # no tensor is moved to find that out
    with pytest.raises(RuntimeError, match="reinitialize"):
        mpc._obs_kwargs(dt, "cpu")
