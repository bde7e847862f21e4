"""Nonlinear-caller mode at scale (what the reference's torch-coded environments hit, al_utils.py:233-265,
618-642): the quad-variant Newton direction (alqp_newton_step_ws) and the one-launch line search
(alqp_merit_pick) against the kernels they replace at large batches, against the oracle, and end to end
through the drop-in MPC."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(B, T, nx, nu, dt, seed=0, active=False):
    from deq_mpc_corl_amd import synthetic_problem
    p = synthetic_problem(B, T, nx, nu, seed=seed, dtype=dt, device=DEV, active=active)
    g = torch.Generator(device="cpu").manual_seed(seed + 7)
    M = T * nx + 2 * T * nu
    lam = (0.3 * torch.randn(B, M, generator=g, dtype=torch.float64)).to(dt).to(DEV)
    lam[:, T * nx:].clamp_(min=0)
    rho = (1.0 + 9.0 * torch.rand(B, generator=g, dtype=torch.float64)).to(dt).to(DEV)
    z = (p.z0 + 0.2 * torch.randn(B, T, nx + nu, generator=g, dtype=torch.float64).to(dt).to(DEV)).contiguous()
    xn = (torch.einsum("btij,btj->bti", p.F, z[:, :-1]) + p.c
          + 0.05 * torch.randn(B, T - 1, nx, generator=g, dtype=torch.float64).to(dt).to(DEV)).contiguous()
    return p, z, xn, lam, rho


@pytest.mark.parametrize("dims", [(40, 20, 13, 4), (33, 10, 8, 2), (50, 5, 2, 1), (20, 10, 14, 4)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
def test_newton_step_quad_equals_team_and_oracle(dims, dtype, tol):
    from deq_mpc_corl_amd.backend import default_backend
    from oracle import oracle_py as orc
    be = default_backend()
    B, T, nx, nu = dims
    p, z, xn, lam, rho = _problem(B, T, nx, nu, dtype, active=True)
    out = {}
    for variant in ("team", "quad"):
        d = torch.empty_like(z)
        g = torch.empty_like(z)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        ws = be.new_workspace(dims, z) if variant == "quad" else None
        be.newton_step(dims, z, xn, p.F, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi, 0, 0, d, g_out=g, info=info, workspace=ws)
        torch.cuda.synchronize()
        assert int(info.abs().max()) == 0
        out[variant] = (d.cpu().numpy(), g.cpu().numpy())
    c = lambda a: a.cpu().numpy()
    s = "f64" if dtype == torch.float64 else "f32"
    go, Hd, Hs = orc.grad_hess(s, c(z), c(xn), c(p.F), c(p.x0), c(lam), c(rho), c(p.Qd), c(p.q), c(p.u_lo), c(p.u_hi))
    do, _ = orc.newton_dir(s, go, Hd, Hs, nx)
    scale = max(1.0, np.abs(do).max())
    for variant in ("team", "quad"):
        assert np.abs(out[variant][1] - go).max() < tol * max(1.0, np.abs(go).max()), variant
        assert np.abs(out[variant][0] - do).max() < tol * scale, variant


@pytest.mark.parametrize("dims", [(37, 20, 13, 4), (64, 5, 2, 1)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("with_obs", [False, True])
def test_merit_pick_equals_merit_then_pick(dims, dtype, with_obs):
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = dims
    if with_obs and nx < 3:
        pytest.skip("obstacle rows need a position x[0:3]")
    p, z, xn, lam, rho = _problem(B, T, nx, nu, dtype, seed=3, active=True)
    n = nx + nu
    g = torch.Generator(device="cpu").manual_seed(11)
    d = (0.5 * torch.randn(B, T, n, generator=g, dtype=torch.float64)).to(dtype).to(DEV)
    obs = None
    if with_obs:
        pos = torch.randn(B, T, 4, 3, generator=g, dtype=torch.float64).to(dtype).to(DEV).contiguous()
        obs = (pos, 0.7)
        lam = torch.cat([lam[:, :T * nx], torch.cat([lam[:, T * nx:].reshape(B, T, 2 * nu),
                                                   0.2 * torch.rand(B, T, 4, generator=g, dtype=torch.float64).to(dtype).to(DEV)], -1)
                         .reshape(B, -1)], 1).contiguous()
    alphas = (2.0 ** -torch.arange(20, device=DEV, dtype=dtype)).view(20, 1, 1, 1)
    zc = (z.unsqueeze(0) + alphas * d.unsqueeze(0)).contiguous()
    xnc = (torch.einsum("btij,kbtj->kbti", p.F, zc[:, :, :-1]) + p.c).contiguous()
    kw = {"obs": obs} if obs is not None else {}
    # reference route: 20 x B merit waves, then pick
    phis = torch.empty(20, B, dtype=dtype, device=DEV)
    rn2s = torch.empty(20, B, dtype=dtype, device=DEV)
    be.merit((B, T, nx, nu), 20, zc, xnc, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi, 0, 0, phis, rn2s, **kw)
    # phi_prev between the candidates' merits: some instances accept, some reject
    phi_prev = (phis.min(0).values + torch.where(torch.arange(B, device=DEV) % 3 == 0, -1.0, 1.0).to(dtype)).contiguous()
    z1, pp1 = z.clone(), phi_prev.clone()
    k1 = torch.zeros(B, dtype=torch.int32, device=DEV)
    a1 = torch.zeros(B, dtype=torch.int32, device=DEV)
    be.linesearch_pick((B, T, nx, nu), 20, phis, pp1, d, z1, k1, a1)
    # one launch
    z2, pp2 = z.clone(), phi_prev.clone()
    k2 = torch.zeros(B, dtype=torch.int32, device=DEV)
    a2 = torch.zeros(B, dtype=torch.int32, device=DEV)
    rn2 = torch.full((B,), -1.0, dtype=dtype, device=DEV)
    phi_all = torch.empty(20, B, dtype=dtype, device=DEV)
    be.merit_pick((B, T, nx, nu), 20, d, xnc, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi, 0, 0, z2, pp2, rnorm2=rn2,
                  phi_all=phi_all, k_out=k2, accept_out=a2, **kw)
    torch.cuda.synchronize()
    rel = 1e-12 if dtype == torch.float64 else 2e-5
    assert torch.allclose(phi_all, phis, rtol=rel, atol=rel * float(phis.abs().max()))
    assert 0 < int(a1.sum()) < B
    if dtype == torch.float64:
        assert torch.equal(k1, k2) and torch.equal(a1, a2)
        assert torch.allclose(z1, z2, rtol=0, atol=1e-14) and torch.allclose(pp1, pp2, rtol=1e-12)
        want = torch.where(a1.bool(), rn2s.gather(0, k1.long().unsqueeze(0)).squeeze(0), torch.full_like(rn2, -1.0))
        assert torch.allclose(rn2, want, rtol=1e-12)
    else:
        same = (k1 == k2) & (a1 == a2)
        assert float(same.float().mean()) > 0.9          # fp32 near-ties between neighbouring candidates
        assert torch.allclose(z1[same], z2[same], atol=1e-6)


@pytest.mark.parametrize("with_grad", [False, True])
def test_nonlinear_caller_mode_quad_route_equals_team_route(with_grad):
    """The drop-in MPC with PyTorch dynamics at a batch that takes the quad Newton step (B >= 4096) against
    the same call forced onto the team kernels: same Newton-step counts, x, u, lamda, gradients."""
    from deq_mpc_corl_amd import MPC, QuadCost, PendulumDynamics, synthetic_problem
    from deq_mpc_corl_amd.backend import HipBackend
    dt = torch.float64
    B, T, nx, nu = 4096, 5, 2, 1
    res = {}
    for route in ("quad", "team"):
        be = HipBackend()
        if route == "team":
            be.QUAD_MIN_BATCH = 1 << 40
        p = synthetic_problem(B, T, nx, nu, seed=5, dtype=dt, device=DEV)
        Qd, q = p.Qd.clone(), p.q.clone()
        if with_grad:
            Qd.requires_grad_(True)
            q.requires_grad_(True)
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, backend=be)
        mpc.reinitialize(p.x0, None)
        mpc.al_iter = 3
        dyn = PendulumDynamics()
        x, u, _ = mpc(p.x0, QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, dtype=dt, device=DEV)), dyn, dyn.jac,
                      x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
        assert be.last_step_kernel == ("k_newton_step_quad" if route == "quad" else "k_newton_step")
        grads = None
        if with_grad:
            (x.sum() + (u * u).sum()).backward()
            grads = (q.grad.clone(), Qd.grad.clone())
        res[route] = (x.detach(), u.detach(), mpc.lamda_prev.clone(), list(mpc.last_newton_per_al), grads)
    a, b = res["quad"], res["team"]
    assert a[3] == b[3]
    assert torch.allclose(a[0], b[0], atol=1e-6) and torch.allclose(a[1], b[1], atol=1e-6)
    # (lamda after three AL iterations, rho = 100: the rounding difference between the root-free LDL' sweep
    #  and the team kernel's Cholesky, amplified by rho)
    assert float((a[2] - b[2]).abs().max()) < 1e-6 * float(b[2].abs().max())
    if with_grad:
        for ga, gb in zip(a[4], b[4]):
            assert float((ga - gb).abs().max()) < 1e-6 * float(gb.abs().max())


@pytest.mark.parametrize("dims", [(40, 20, 13, 4), (33, 10, 8, 2), (4096, 20, 13, 4)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
@pytest.mark.parametrize("rows", ["obstacles", "state_estimator"])
def test_newton_step_quad_with_extra_rows_equals_team(dims, dtype, tol, rows):
    """Obstacle rows (Obstacle_MPC, al_utils.py:313-323, 351-388) and the state-estimator row set on the quad step
    kernel (alqp_newton_step_ws_obs) against the team step kernel that the reference fixtures pin
    (tests/test_obstacles_golden.py, test_state_estimator_golden.py): ragged batches and the B = 4096 the class switches at."""
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = dims
    p, z, xn, lam, rho = _problem(B, T, nx, nu, dtype, active=True)
    if rows == "obstacles":
        nobs = 4
        g = torch.Generator(device="cpu").manual_seed(11)
        # spheres around the trajectory's positions: about a third of the rows active (c_k >= 0)
        pos = (z[:, :, None, :3].cpu().double() + 0.25 * torch.randn(B, T, nobs, 3, generator=g, dtype=torch.float64)).to(dtype).to(DEV).contiguous()
        obs = (pos, 0.3)
        lam_o = (0.2 * torch.rand(B, T, nobs, generator=g, dtype=torch.float64)).to(dtype).to(DEV)
        lam = torch.cat([lam[:, :T * nx], torch.cat([lam[:, T * nx:].reshape(B, T, 2 * nu), lam_o], 2).reshape(B, -1)], 1).contiguous()
    else:
        obs = "state_estimator"
        F = p.F.clone()
        F[..., nx:] = 0        # as the host class prepares it: the control Jacobian is multiplied by 0 (al_utils_se.py:151)
        p = p._replace(F=F.contiguous())
    out = {}
    for variant in ("team", "quad"):
        d = torch.empty_like(z)
        g_ = torch.empty_like(z)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        ws = be.new_workspace(dims, z) if variant == "quad" else None
        be.newton_step(dims, z, xn, p.F, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi, 0, 0, d, g_out=g_, info=info, obs=obs,
                       workspace=ws)
        torch.cuda.synchronize()
        assert int(info.abs().max()) == 0
        out[variant] = (d.cpu().numpy(), g_.cpu().numpy())
    scale = max(1.0, float(np.abs(out["team"][0]).max()))
    assert np.abs(out["quad"][1] - out["team"][1]).max() < tol * max(1.0, float(np.abs(out["team"][1]).max()))
    assert np.abs(out["quad"][0] - out["team"][0]).max() < tol * scale
    if rows == "state_estimator":
        assert np.abs(out["quad"][0][..., nx:]).max() < tol * scale      # du = 0: the controls are given
