"""GPU edge cases of the hot path, both kernel variants: per-(b,t) bounds through the
stride arguments, minimum horizon / single instance, non-finite input and indefinite
Hessian reported per instance without disturbing the neighbours, fewer line-search
candidates, and equivalence of one fused launch with the same work split over launches."""
import numpy as np
import pytest
import torch

from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VARIANTS = ["team", "quad"]


def solve(p, dt, variant, al_iter=2, max_newton=4, n_ls=20, flags=3, ulo=None, uhi=None, sb=0, st=0,
          z=None, lam=None, rho=None, phi=None):
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = p.B, p.T, p.nx, p.nu
    M = T * nx + 2 * T * nu
    z = p.z0.clone() if z is None else z
    lam = torch.zeros(B, M, dtype=dt, device=DEV) if lam is None else lam
    rho = torch.ones(B, dtype=dt, device=DEV) if rho is None else rho
    phi = torch.zeros(B, dtype=dt, device=DEV) if phi is None else phi
    rn2 = torch.zeros(B, dtype=dt, device=DEV)
    info = torch.zeros(B, dtype=torch.int32, device=DEV)
    status = torch.zeros(B, dtype=torch.uint8, device=DEV)
    be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo if ulo is None else ulo,
                 p.u_hi if uhi is None else uhi, sb, st, z, lam, rho, phi, rn2, info, status,
                 al_iter=al_iter, max_newton=max_newton, n_ls=n_ls, flags=flags, variant=variant)
    torch.cuda.synchronize()
    return z, lam, rho, phi, rn2, info, status


def c(a):
    return a.cpu().numpy()


@pytest.mark.parametrize("variant", VARIANTS)
def test_per_instance_per_stage_bounds(variant):
    """u_lower/u_upper of shape [B,T,nu] (the reference broadcasts whatever it is given,
    al_utils.py:293) reach the kernels through the (sb_u, st_u) element strides."""
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float64
    B, T, nx, nu = 21, 6, 8, 2
    p = synthetic_problem(B, T, nx, nu, seed=2, dtype=dt, device=DEV, active=True)
    g = torch.Generator(device="cpu").manual_seed(4)
    hi = (0.05 + 0.3 * torch.rand(B, T, nu, generator=g, dtype=dt)).to(DEV)
    lo = -(0.05 + 0.3 * torch.rand(B, T, nu, generator=g, dtype=dt)).to(DEV)
    z, lam, *_ = solve(p, dt, variant, ulo=lo.contiguous(), uhi=hi.contiguous(), sb=T * nu, st=nu)
    o = orc.solve_lin("f64", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(lo), c(hi), c(p.z0), al_iter=2,
                      exit_mode="fixed")
    assert np.abs(c(z) - o["z"]).max() < 1e-9
    assert np.abs(c(lam) - o["lam"]).max() < 1e-8
    assert float(lam[:, T * nx:].max()) > 0  # some bound multipliers became active


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("B,T", [(1, 2), (1, 20), (17, 2)])
def test_minimum_horizon_and_single_instance(variant, B, T):
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float64
    p = synthetic_problem(B, T, 13, 4, seed=3, dtype=dt, device=DEV)
    z, lam, rho, *_ = solve(p, dt, variant)
    o = orc.solve_lin("f64", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0), al_iter=2,
                      exit_mode="fixed")
    assert np.abs(c(z) - o["z"]).max() < 1e-9
    assert np.allclose(c(rho), 100.0)


@pytest.mark.parametrize("variant", VARIANTS)
def test_nonfinite_input_is_flagged_per_instance(variant):
    """A NaN in one instance's x0 makes that instance's status 0 (al_utils.py:545-549) and
    leaves every other instance bit-for-bit unchanged."""
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float32
    B = 40
    p = synthetic_problem(B, 10, 8, 2, seed=6, dtype=dt, device=DEV)
    z_ref, lam_ref, *_ , st_ref = solve(p, dt, variant)
    assert int(st_ref.sum()) == B
    x0 = p.x0.clone()
    x0[7, 3] = float("nan")
    p2 = p._replace(x0=x0)
    z, lam, _, _, _, info, status = solve(p2, dt, variant)
    assert int(status[7]) == 0 and int(status.sum()) == B - 1
    keep = torch.ones(B, dtype=torch.bool, device=DEV)
    keep[7] = False
    assert torch.equal(z[keep], z_ref[keep]) and torch.equal(lam[keep], lam_ref[keep])


@pytest.mark.parametrize("variant", VARIANTS)
def test_indefinite_hessian_is_reported_in_info(variant):
    """Negative cost weights make H indefinite for one instance: its info holds the first
    non-positive pivot (stage*n + column + 1, like cholesky_ex's info); the rest are clean."""
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float64
    B, T, nx, nu = 24, 8, 8, 2
    p = synthetic_problem(B, T, nx, nu, seed=8, dtype=dt, device=DEV)
    Qd = p.Qd.clone()
    Qd[5, 2, :] = -50.0
    p2 = p._replace(Qd=Qd, q=-(Qd * p.xref))
    z, lam, _, _, _, info, status = solve(p2, dt, variant, al_iter=1, max_newton=1)
    info = c(info)
    assert info[5] != 0 and (np.delete(info, 5) == 0).all()
    stage, col = divmod(int(info[5]) - 1, nx + nu)
    assert stage <= 2


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_ls", [1, 5])
def test_fewer_line_search_candidates(variant, n_ls):
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float64
    p = synthetic_problem(12, 7, 13, 4, seed=10, dtype=dt, device=DEV)
    z, lam, *_ = solve(p, dt, variant, n_ls=n_ls)
    o = orc.solve_lin("f64", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0), al_iter=2,
                      n_ls=n_ls, exit_mode="fixed")
    assert np.abs(c(z) - o["z"]).max() < 1e-9


@pytest.mark.parametrize("variant", VARIANTS)
def test_one_launch_equals_step_by_step_launches(variant):
    """The host-driven sequence the 'reference' exit mode uses (merit init, one Newton step per
    launch, dual update; state carried in z/lam/rho/phi between launches) is the same
    arithmetic as the single fused launch: results agree to rounding (the factor is rebuilt
    from identical inputs, so in fact bit for bit)."""
    from deq_mpc_corl_amd import synthetic_problem
    dt = torch.float32
    p = synthetic_problem(50, 10, 8, 2, seed=12, dtype=dt, device=DEV)
    zf, lamf, rhof, *_ = solve(p, dt, variant, al_iter=2, max_newton=4, flags=3)
    z, lam, rho, phi = p.z0.clone(), None, None, None
    z, lam, rho, phi, *_ = solve(p, dt, variant, al_iter=1, max_newton=0, flags=1, z=z)
    for it in range(2):
        if it > 0:
            z, lam, rho, phi, *_ = solve(p, dt, variant, al_iter=1, max_newton=0, flags=1, z=z, lam=lam, rho=rho, phi=phi)
        for _ in range(4):
            z, lam, rho, phi, *_ = solve(p, dt, variant, al_iter=1, max_newton=1, flags=0, z=z, lam=lam, rho=rho, phi=phi)
        z, lam, rho, phi, *_ = solve(p, dt, variant, al_iter=1, max_newton=0, flags=2, z=z, lam=lam, rho=rho, phi=phi)
    assert torch.allclose(z, zf, atol=1e-6) and torch.allclose(lam, lamf, atol=1e-5)
    assert torch.equal(rho, rhof)


def test_primed_workspace_skips_copy_in():
    """ALQP_WS_PRIMED (quad): a Newton-step launch that follows another launch of the same solve on
    the same workspace may skip the copy-in pass; the results are bit-identical to the unprimed
    sequence, and the arrays (z, lam) stay in sync with the records after every launch."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd import _lib
    dt = torch.float32
    p = synthetic_problem(40, 10, 13, 4, seed=14, dtype=dt, device=DEV, active=True)
    outs = []
    for primed in (0, _lib.ALQP_WS_PRIMED):
        z, lam, rho, phi, *_ = solve(p, dt, "quad", al_iter=1, max_newton=0, flags=1)
        for _ in range(3):
            z, lam, rho, phi, *_ = solve(p, dt, "quad", al_iter=1, max_newton=1, flags=primed, z=z, lam=lam, rho=rho, phi=phi)
        z, lam, rho, phi, *_ = solve(p, dt, "quad", al_iter=1, max_newton=0, flags=2, z=z, lam=lam, rho=rho, phi=phi)
        outs.append((z.clone(), lam.clone(), rho.clone(), phi.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float((outs[0][0] - p.z0).abs().max()) > 1e-3  # the steps did move z


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("flags,max_newton", [(3, 4), (1, 0), (2, 0), (0, 1)])
def test_quad_result_does_not_depend_on_workspace_contents(dt, flags, max_newton):
    """Nothing in the workspace is read before this launch has written it (unless the caller says
    ALQP_WS_PRIMED): a NaN-poisoned and a zeroed workspace give bit-identical results, for the fused
    solve and for the single-purpose launches of the host-driven mode."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = 37, 7, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=31, dtype=dt, device=DEV, active=True)
    M = T * nx + 2 * T * nu
    outs = []
    for fill in (float("nan"), 0.0, 1e30):
        ws = be.new_workspace((B, T, nx, nu), p.z0)
        ws.fill_(fill)
        z = p.z0.clone()
        lam = 0.1 * torch.ones(B, M, dtype=dt, device=DEV)
        rho = torch.full((B,), 2.0, dtype=dt, device=DEV)
        phi = torch.zeros(B, dtype=dt, device=DEV)
        rn2 = torch.zeros(B, dtype=dt, device=DEV)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        status = torch.zeros(B, dtype=torch.uint8, device=DEV)
        be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2, info,
                     status, al_iter=(2 if max_newton == 4 else 1), max_newton=max_newton, n_ls=20, flags=flags,
                     variant="quad", workspace=ws)
        torch.cuda.synchronize()
        outs.append((z, lam, rho, phi, rn2, status))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert int(outs[0][5].sum()) == B


def test_streaming_mode_matches_cpu_host_logic():
    """warm_start_initialize + al_solve_stream semantics (AL_mpc.py:342-423, 581-592) on the GPU
    backend against the same host logic driven by the test-only oracle backend: lamda zeroed,
    rho clamped, iteration stops once rho exceeds 1e8 and reports status True."""
    from types import SimpleNamespace
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem
    from tests.oracle_backend import OracleBackend
    dt = torch.float64
    B, T, nx, nu = 12, 8, 8, 2
    out = {}
    for name, device, backend in (("hip", DEV, None), ("cpu", "cpu", OracleBackend())):
        p = synthetic_problem(B, T, nx, nu, seed=21, dtype=dt, device=device)
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, backend=backend)
        mpc.reinitialize(p.x0, None)
        dyn = AffineDynamics(p.F, p.c)
        cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dt, device=device))
        mpc.al_iter = 2
        x, u, s0 = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
        mpc.warm_start_initialize(x.to(dt), u.to(dt), SimpleNamespace(rho_init_max=50.0))
        mpc.al_iter = 10
        x2, u2, s1 = mpc(p.x0, cost, dyn, dyn.jac)
        out[name] = (x.cpu(), u.cpu(), s0, x2.cpu(), u2.cpu(), s1, mpc.rho_prev.cpu(), list(mpc.last_newton_per_al))
    h, cc = out["hip"], out["cpu"]
    assert h[2] is False and h[5] is True and cc[5] is True
    assert h[7] == cc[7]
    assert torch.allclose(h[0], cc[0], atol=1e-6) and torch.allclose(h[1], cc[1], atol=1e-6)
    assert torch.allclose(h[3], cc[3], atol=1e-5) and torch.allclose(h[4], cc[4], atol=1e-5)
    assert torch.equal(h[6], cc[6])


def test_bad_arguments_fail_loudly():
    """Error behaviour of the boundary: unsupported sizes, missing workspace, wrong line-search length
    for the nonlinear solve, CPU tensors and dtype mismatches raise; nothing falls back silently."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    dt = torch.float64
    p = synthetic_problem(8, 5, 8, 2, seed=1, dtype=dt, device=DEV)
    with pytest.raises(RuntimeError):                      # (5, 3) is not an instantiated size
        q = synthetic_problem(8, 5, 5, 3, seed=1, dtype=dt, device=DEV)
        solve(q, dt, "quad")
    assert not be.supported(8, 5, 5, 3, dt) and be.supported(8, 5, 8, 2, dt)
    with pytest.raises(ValueError):                        # explicit workspace too small
        be.solve_lin((8, 5, 8, 2), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, p.z0.clone(),
                     torch.zeros(8, 5 * 8 + 2 * 5 * 2, dtype=dt, device=DEV), torch.ones(8, dtype=dt, device=DEV),
                     torch.zeros(8, dtype=dt, device=DEV), variant="quad", workspace=torch.zeros(16, dtype=dt, device=DEV))
    with pytest.raises(RuntimeError):                      # CPU tensor
        be.solve_lin((8, 5, 8, 2), p.Qd.cpu(), p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, p.z0.clone(),
                     torch.zeros(8, 60, dtype=dt, device=DEV), torch.ones(8, dtype=dt, device=DEV),
                     torch.zeros(8, dtype=dt, device=DEV))
    with pytest.raises(TypeError):                         # dtype mismatch
        be.solve_lin((8, 5, 8, 2), p.Qd.float(), p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, p.z0.clone(),
                     torch.zeros(8, 60, dtype=dt, device=DEV), torch.ones(8, dtype=dt, device=DEV),
                     torch.zeros(8, dtype=dt, device=DEV))
    with pytest.raises(RuntimeError):                      # model id / size mismatch in the nonlinear solve
        be.solve_nonlin((8, 5, 8, 2), 1, 0.05, p.Qd, p.q, p.x0, p.u_lo, p.u_hi, 0, 0, p.z0.clone(),
                        torch.zeros(8, 60, dtype=dt, device=DEV), torch.ones(8, dtype=dt, device=DEV),
                        torch.zeros(8, dtype=dt, device=DEV))


def test_exit_test_kernel_matches_the_rule():
    """alqp_exit_test: mode 0 initialises {done, steps, old}; mode 1 counts a step and applies
    new < tol or |old - new| / new < tol (al_utils.py:560-564); once done it is inert."""
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    ctl = torch.full((3,), 7.0, dtype=torch.float64, device=DEV)
    s = lambda v: torch.tensor([v], dtype=torch.float64, device=DEV)
    be.exit_test(s(4.0), ctl, 0)
    assert ctl.tolist() == [0.0, 0.0, 2.0]
    be.exit_test(s(1.0), ctl, 1)                           # 2 -> 1: keeps going
    assert ctl.tolist() == [0.0, 1.0, 1.0]
    be.exit_test(s(1.0 * (1 + 1e-4) ** 2), ctl, 1)         # relative change 1e-4 < 1e-3: done
    assert ctl[0].item() == 1.0 and ctl[1].item() == 2.0
    be.exit_test(s(100.0), ctl, 1)                         # inert afterwards
    assert ctl[1].item() == 2.0
    be.exit_test(s(1e-8), ctl, 0)
    be.exit_test(s(1e-8), ctl, 1)                          # new = 1e-4 < 1e-3: done after one step
    assert ctl[0].item() == 1.0 and ctl[1].item() == 1.0
    be.exit_test(s(float("nan")), ctl, 0)
    be.exit_test(s(float("nan")), ctl, 1)                  # NaN compares false: keeps going (as torch does)
    assert ctl[0].item() == 0.0 and ctl[1].item() == 1.0


def test_long_horizon_beyond_team_lds_runs_on_quad_at_small_batch():
    """T = 400 at (13,4): the team variant's LDS image does not fit; 'auto' must take the quad kernels even
    at a small batch instead of refusing (alqp_supported is per variant now)."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    from oracle import oracle_py as orc
    be = default_backend()
    dt = torch.float64
    B, T, nx, nu = 5, 400, 13, 4
    assert be.supported(B, T, nx, nu, dt)
    p = synthetic_problem(B, T, nx, nu, seed=2, dtype=dt, device=DEV)
    M = T * nx + 2 * T * nu
    z = p.z0.clone()
    lam = torch.zeros(B, M, dtype=dt, device=DEV)
    rho = torch.ones(B, dtype=dt, device=DEV)
    phi = torch.zeros(B, dtype=dt, device=DEV)
    info = torch.zeros(B, dtype=torch.int32, device=DEV)
    st = torch.zeros(B, dtype=torch.uint8, device=DEV)
    be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, None, info, st,
                 al_iter=1, max_newton=2, n_ls=20, flags=3)
    torch.cuda.synchronize()
    assert be.last_variant == "quad" and int(info.abs().sum()) == 0 and int(st.sum()) == B
    c = lambda a: a.cpu().numpy()
    o = orc.solve_lin("f64", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0), al_iter=1,
                      max_newton=2, exit_mode="fixed")
    assert np.abs(c(z) - o["z"]).max() < 1e-8


@pytest.mark.gpu
def test_quad_stagger_changes_timing_only():
    """alqp_set_quad_stagger: the start offset between a CU's wavefronts must not change a single bit of the
    results (B = 16384 fills the SIMDs, so the automatic rule is active)."""
    from deq_mpc_corl_amd.backend import default_backend
    from deq_mpc_corl_amd.problems import synthetic_problem
    be = default_backend()
    dev = "cuda:0"
    B, T, nx, nu = 16384, 20, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=5, dtype=torch.float32)
    mv = lambda t: t.to(dev).contiguous()
    outs = []
    prev = be.set_quad_stagger(-1)
    try:
        for mode in (0, -1, 37):
            be.set_quad_stagger(mode)
            z, lam = mv(p.z0).clone(), torch.zeros(B, T * nx + 2 * T * nu, device=dev)
            rho, phi = torch.ones(B, device=dev), torch.zeros(B, device=dev)
            be.solve_lin((B, T, nx, nu), mv(p.Qd), mv(p.q), mv(p.F), mv(p.c), mv(p.x0), mv(p.u_lo), mv(p.u_hi), 0, 0,
                         z, lam, rho, phi, al_iter=2, max_newton=4, variant="quad")
            torch.cuda.synchronize()
            outs.append((z.cpu(), lam.cpu(), rho.cpu(), phi.cpu()))
    finally:
        be.set_quad_stagger(prev)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B,dtype", [(200, torch.float64), (4099, torch.float32), (16384, torch.float32)])
def test_reference_exit_inside_one_cooperative_launch_equals_launch_per_step(B, dtype):
    """ALQP_EXIT_IN_KERNEL: the batch-global exit test of the Newton loop taken behind a grid-wide barrier inside one
    launch must reproduce the launch-per-step route (alqp_exit_test between one-step launches): same Newton-step counts,
    same iterate, multipliers and penalty (team kernels at B = 200, quad kernels above; 4099 is not a multiple of 16)."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
    from deq_mpc_corl_amd.problems import synthetic_problem
    dev = "cuda:0"
    T, nx, nu = 20, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=11, dtype=dtype, device=dev, active=True)
    cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dtype, device=dev))
    dyn = AffineDynamics(p.F, p.c)
    res = {}
    for in_kernel in (True, False):
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dtype, exit_mode="reference",
                  exit_in_kernel=in_kernel)
        mpc.reinitialize(p.x0, None)
        mpc.al_iter = 3
        x, u, _ = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
        torch.cuda.synchronize()
        res[in_kernel] = (list(mpc.last_newton_per_al), x.cpu(), u.cpu(), mpc.lamda_prev.cpu(), mpc.rho_prev.cpu())
    assert res[True][0] == res[False][0] and len(res[True][0]) == 3
    assert 1 <= min(res[True][0]) and max(res[True][0]) <= 4
    # not bit-identical in the team kernels: a launch starts from residuals evaluated afresh at z, the single launch
    # carries r + alpha s from step to step (as the fixed-mode launch always has)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    for a, b in zip(res[True][1:], res[False][1:]):
        assert float((a.double() - b.double()).abs().max()) <= tol * max(1.0, float(b.double().abs().max()))
