"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs, against the reference's golden outputs, and - at the benchmark's
full size - through size-independent properties. Tolerances are stated per test:
fp64 kernels agree with the fp64 oracle to ~1e-9 relative; fp32 kernels to ~2e-4 at
rho <= 1e3 (the regime in which the reference's own fp32 run agrees with its fp64
run)."""
import numpy as np
import pytest
import torch

from oracle import oracle_py as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TD = {"f64": torch.float64, "f32": torch.float32}
LIN = [n for n in gu.names() if "nonlin" not in n and "carry" not in n and "tracking" not in n]


def dev(a, dt):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device=DEV, dtype=dt).contiguous()


def run_fused(g, dtype, al_iter, trace=True, max_newton=4, flags=3, lam0=None, rho0=None, factor=False,
              variant="team"):
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    dt = TD[dtype]
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    n = nx + nu
    M = T * nx + 2 * T * nu
    S = al_iter * max_newton
    z = dev(g["z0"], dt)
    lam = torch.zeros(B, M, dtype=dt, device=DEV) if lam0 is None else dev(lam0, dt)
    rho = torch.ones(B, dtype=dt, device=DEV) if rho0 is None else dev(rho0, dt)
    phi = torch.zeros(B, dtype=dt, device=DEV)
    rn2 = torch.zeros(B, dtype=dt, device=DEV)
    info = torch.zeros(B, dtype=torch.int32, device=DEV)
    status = torch.zeros(B, dtype=torch.uint8, device=DEV)
    tr = None
    if trace:
        tr = {"g": torch.zeros(S, B, T, n, dtype=dt, device=DEV),
              "d": torch.zeros(S, B, T, n, dtype=dt, device=DEV),
              "phi": torch.zeros(S, 20, B, dtype=dt, device=DEV),
              "phi_prev": torch.zeros(S, B, dtype=dt, device=DEV),
              "k": torch.zeros(S, B, dtype=torch.int32, device=DEV),
              "accept": torch.zeros(S, B, dtype=torch.int32, device=DEV)}
    fac = torch.zeros(B, T, n * (n + 1) // 2, dtype=dt, device=DEV) if factor else None
    ulo, uhi = dev(g["u_lo"], dt), dev(g["u_hi"], dt)
    be.solve_lin((B, T, nx, nu), dev(g["Qd"], dt), dev(g["q"], dt), dev(g["F"], dt), dev(g["c"], dt),
                 dev(g["x0"], dt), ulo, uhi, 0, 0, z, lam, rho, phi, rn2, info, status,
                 factor=fac, al_iter=al_iter, max_newton=max_newton, n_ls=20,
                 flags=flags | (4 if factor else 0), rho_scale=10.0, trace=tr, variant=variant)
    torch.cuda.synchronize()
    out = {"z": z.cpu().numpy(), "lam": lam.cpu().numpy(), "rho": rho.cpu().numpy(),
           "phi": phi.cpu().numpy(), "rn2": rn2.cpu().numpy(), "info": info.cpu().numpy(),
           "status": status.cpu().numpy()}
    if trace:
        out["tr"] = {k: v.cpu().numpy() for k, v in tr.items()}
    if factor:
        out["factor"] = fac
    return out


def near_tie_instances(o, dtype):
    """Instances whose line search has, at some step, two candidates (or the best candidate
    and the previous merit) closer than the arithmetic can resolve: there the kernel and the
    oracle may legitimately choose different steps (fp32: 1 ulp of a merit ~450 is 3e-5)."""
    eps = 1e-11 if dtype == "f64" else 2e-6
    B = o["phi"].shape[-1]
    tie = np.zeros(B, bool)
    for s in range(o["phi"].shape[0]):
        ph = o["phi"][s]
        if not np.isfinite(ph).any():
            continue
        srt = np.sort(ph, axis=0)
        scale = np.abs(o["phi_prev"][s]) + 1
        big = np.abs(o["d"][s]).reshape(B, -1).max(1) > 1e-4
        tie |= (((srt[1] - srt[0]) < eps * scale) | (np.abs(srt[0] - o["phi_prev"][s]) < eps * scale)) & big
    return tie


# Near-tie instances are a property of the ORACLE's trace on the seeded problem (near_tie_instances), so their number is
# known without a GPU: counted on the CPU for every budgeted test (round 3); each budget below is that count + 1.
NEAR_TIES_DIMS_F32 = {(2, 1): 0, (4, 1): 1, (4, 2): 0, (6, 1): 5, (6, 2): 5, (8, 2): 1, (10, 3): 2, (12, 4): 5, (13, 4): 2, (14, 4): 3}
NEAR_TIES_T50 = {"f32": 1, "f64": 0}
NEAR_TIES_FULL = {"pend": 3, "(8,2": 1, "(13,": 4}


def check_excluded(dtype, excluded, zg, lamg, rho, o, prob, max_count=1, label=""):
    """fp32 accounting (an instance whose 20-point line search met a near-tie may take another step than
    the oracle and is left out of the element-wise comparison - but not out of every check):
      * the excluded number is printed and bounded by max_count = the number counted on the CPU for this test + 1;
      * every excluded instance's GPU iterate is finite;
      * its merit, re-evaluated BY THE ORACLE at the GPU's (z, lam, rho), is no worse than the oracle's own
        final merit + 1e-3 (|phi| + 1): a different tie-break must not be a worse point.
    prob = dict(Qd, q, F, c, x0, u_lo, u_hi) numpy; o = oracle result dict with z, lam, rho."""
    B = len(excluded)
    n_ex = int(excluded.sum())
    print(f"[fp32 accounting] {label}: {n_ex}/{B} instances excluded as line-search near-ties ({100.0 * n_ex / B:.1f} %)")
    assert n_ex <= max_count, (label, n_ex, max_count, B)
    if n_ex == 0:
        return
    assert np.isfinite(zg[excluded]).all() and np.isfinite(lamg[excluded]).all(), label
    xn = lambda z: np.einsum("btij,btj->bti", prob["F"], z[:, :-1]) + prob["c"]
    pg, _ = orc.merit(dtype, zg, xn(zg).astype(zg.dtype), prob["x0"], lamg, rho, prob["Qd"], prob["q"], prob["u_lo"], prob["u_hi"])
    po, _ = orc.merit(dtype, o["z"], xn(o["z"]).astype(zg.dtype), prob["x0"], o["lam"], o["rho"], prob["Qd"], prob["q"],
                      prob["u_lo"], prob["u_hi"])
    worse = pg[excluded] - po[excluded]
    assert (worse <= 1e-3 * (np.abs(po[excluded]) + 1)).all(), (label, worse.max(), po[excluded])


def scale_err(a, b, floor):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


@pytest.mark.parametrize("variant", ["team", "quad"])
@pytest.mark.parametrize("name", LIN)
def test_fused_solve_vs_oracle(name, variant):
    """Fixed-4-step fused kernel (both variants) vs the oracle run in the same mode, step by step."""
    g = gu.load(name)
    dt = g["dtype"]
    al = min(g["al_iter"], 4 if dt == "f64" else 2)
    S = al * 4
    o = orc.solve_lin(dt, g["Qd"], g["q"], g["F"], g["c"], g["x0"], g["u_lo"], g["u_hi"], g["z0"],
                      al_iter=al, exit_mode="fixed", trace_steps=S)
    h = run_fused(g, dt, al, variant=variant)
    assert (h["info"] == 0).all() and (h["status"] == 1).all()
    rt = 1e-8 if dt == "f64" else 3e-4
    zs = np.abs(g["z0"]).max()
    g0 = np.abs(o["g"][0]).max()
    diverged = np.zeros(g["B"], bool)  # instances whose line search legitimately took another branch
    for s in range(S):
        ok = ~diverged
        assert scale_err(h["tr"]["g"][s][ok], o["g"][s][ok], 1e-5 * g0) < rt * 10, ("g", s)
        assert scale_err(h["tr"]["d"][s][ok], o["d"][s][ok], 1e-5 * zs) < rt * 100, ("d", s)
        # decisions: only where they are not rounding noise
        phi_o = o["phi"][s]
        srt = np.sort(phi_o, axis=0)
        gap = srt[1] - srt[0]
        margin = np.abs(phi_o.min(0) - o["phi_prev"][s])
        scale = np.abs(o["phi_prev"][s]) + 1
        eps = (1e-10 if dt == "f64" else 3e-5) * scale
        sure = (gap > eps) & (margin > eps) & ok
        assert np.array_equal(h["tr"]["k"][s][sure], o["k"][s][sure]), ("k", s)
        assert np.array_equal(h["tr"]["accept"][s][sure], o["accept"][s][sure]), ("accept", s)
        differs = (h["tr"]["k"][s] != o["k"][s]) | (h["tr"]["accept"][s] != o["accept"][s])
        if dt == "f32":
            # an fp32 near-tie may legitimately resolve differently; it only matters
            # when the step itself is not negligible (converged instances have d ~ 0)
            big = np.abs(o["d"][s]).reshape(g["B"], -1).max(1) > 1e-3 * zs
            diverged |= differs & ~sure & big
    ok = ~diverged
    check_excluded(dt, diverged, h["z"], h["lam"], h["rho"], o, g, label=f"{name}/{variant}")
    assert scale_err(h["z"][ok], o["z"][ok], 1e-3 * zs) < rt * 100
    assert scale_err(h["lam"][ok], o["lam"][ok], 1e-3) < rt * 1000
    assert np.allclose(h["rho"], o["rho"])


@pytest.mark.parametrize("name", [n for n in LIN if n.endswith("_al2")])
def test_mpc_reference_mode_vs_reference_outputs(name):
    """Drop-in class, exit_mode='reference', against what the reference itself returned."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
    g = gu.load(name)
    dt = TD[g["dtype"]]
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    dyn = AffineDynamics(dev(g["F"], dt), dev(g["c"], dt))
    mpc = MPC(nx, nu, T, u_lower=dev(g["u_lo"], dt), u_upper=dev(g["u_hi"], dt), n_batch=B, dtype=dt,
              exit_mode="reference")
    x0 = dev(g["x0"], dt)
    mpc.reinitialize(x0, None)
    mpc.al_iter = g["al_iter"]
    cost = QuadCost(torch.diag_embed(dev(g["Qd"], dt)), dev(g["q"], dt), torch.zeros(B, T, device=DEV, dtype=dt))
    z0 = dev(g["z0"], dt)
    x, u, status = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    assert status is False
    assert x.dtype == torch.float32 and u.dtype == torch.float32
    assert list(mpc.last_newton_per_al) == list(g["newton_per_al"])
    tol = 2e-5 if g["dtype"] == "f64" else 5e-3
    assert np.abs(x.cpu().numpy() - g["x"]).max() < tol
    assert np.abs(u.cpu().numpy() - g["u"]).max() < tol
    assert np.allclose(mpc.rho_prev.cpu().numpy(), g["rho_final"])
    rel = 1e-6 if g["dtype"] == "f64" else 5e-2
    assert scale_err(mpc.lamda_prev.cpu().numpy(), g["lam_final"], 1e-3) < rel


@pytest.mark.parametrize("name", gu.names("*nonlin*"))
def test_mpc_nonlinear_mode_vs_reference_outputs(name):
    """Arbitrary dx/dx_jac callables between kernel launches (nonlinear-caller mode)."""
    from deq_mpc_corl_amd import MPC, PendulumDynamics, QuadCost
    g = gu.load(name)
    dt = TD[g["dtype"]]
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    dyn = PendulumDynamics()
    mpc = MPC(nx, nu, T, u_lower=dev(g["u_lo"], dt), u_upper=dev(g["u_hi"], dt), n_batch=B, dtype=dt,
              exit_mode="reference")
    x0 = dev(g["x0"], dt)
    mpc.reinitialize(x0, None)
    mpc.al_iter = g["al_iter"]
    cost = QuadCost(torch.diag_embed(dev(g["Qd"], dt)), dev(g["q"], dt), torch.zeros(B, T, device=DEV, dtype=dt))
    z0 = dev(g["z0"], dt)
    x, u, status = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    assert list(mpc.last_newton_per_al) == list(g["newton_per_al"])
    tol = 2e-5 if g["dtype"] == "f64" else 5e-3
    assert np.abs(x.cpu().numpy() - g["x"]).max() < tol
    assert np.abs(u.cpu().numpy() - g["u"]).max() < tol


@pytest.mark.parametrize("name", ["pend_f64_al2", "cart_f64_al2", "pend_active_f64_al6", "pend_nonlin_f64_al4"])
def test_backward_vs_reference_grads(name):
    from deq_mpc_corl_amd import MPC, AffineDynamics, PendulumDynamics, QuadCost
    g = gu.load(name)
    dt = torch.float64
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    dyn = PendulumDynamics() if g["nonlinear"] else AffineDynamics(dev(g["F"], dt), dev(g["c"], dt))
    mpc = MPC(nx, nu, T, u_lower=dev(g["u_lo"], dt), u_upper=dev(g["u_hi"], dt), n_batch=B, dtype=dt,
              exit_mode="reference")
    x0 = dev(g["x0"], dt)
    mpc.reinitialize(x0, None)
    mpc.al_iter = g["al_iter"]
    Qd = dev(g["Qd"], dt).requires_grad_(True)
    q = dev(g["q"], dt).requires_grad_(True)
    cost = QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, device=DEV, dtype=dt))
    z0 = dev(g["z0"], dt)
    x, u, _ = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    loss = (x * dev(g["bwd_wx"], torch.float32)).sum() + (u * dev(g["bwd_wu"], torch.float32)).sum()
    loss.backward()
    assert scale_err(q.grad.cpu().numpy(), g["bwd_q_grad"], 1e-12) < 1e-5
    assert scale_err(Qd.grad.cpu().numpy(), g["bwd_Qd_grad"], 1e-12) < 1e-5


def test_state_carry_vs_reference():
    """reinitialize + 3 calls: lamda/rho/x_init/u_init persist, rho 1->1e2->1e4->1e6."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
    g = gu.load("cart_carry_f64")
    dt = torch.float64
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    dyn = AffineDynamics(dev(g["F"], dt), dev(g["c"], dt))
    mpc = MPC(nx, nu, T, u_lower=dev(g["u_lo"], dt), u_upper=dev(g["u_hi"], dt), n_batch=B, dtype=dt)
    x0 = dev(g["x0"], dt)
    mpc.reinitialize(x0, None)
    z0 = dev(g["z0"], dt)
    mpc.x_init, mpc.u_init = z0[..., :nx].clone(), z0[..., nx:].clone()
    C = torch.diag_embed(dev(g["Qd"], dt))
    for i in range(g["calls"]):
        mpc.al_iter = 2
        cost = QuadCost(C, dev(g["q"][i], dt), torch.zeros(B, T, device=DEV, dtype=dt))
        x, u, _ = mpc(x0, cost, dyn, dyn.jac)
        assert list(mpc.last_newton_per_al) == list(g["newton_per_al"][i])
        assert np.abs(x.cpu().numpy() - g["x"][i]).max() < 5e-5 * (i + 1)
        assert np.abs(u.cpu().numpy() - g["u"][i]).max() < 5e-5 * (i + 1)
        assert np.allclose(mpc.rho_prev.cpu().numpy(), g["rho"][i])


def test_headline_size_properties():
    """(B=16384, T=20, nx=13, nu=4) fp32, the benchmark workload: a sample of instances
    against the oracle, plus properties that need no oracle: the result of an instance
    does not depend on its position in the batch (bit-exact under permutation) and the
    AL solve reduces the constraint violation."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = 16384, 20, 13, 4
    dt = torch.float32
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=DEV)
    M = T * nx + 2 * T * nu

    def solve(perm=None):
        sel = (lambda a: a) if perm is None else (lambda a: a[perm].contiguous())
        z = sel(p.z0).clone()
        lam = torch.zeros(B, M, dtype=dt, device=DEV)
        rho = torch.ones(B, dtype=dt, device=DEV)
        phi = torch.zeros(B, dtype=dt, device=DEV)
        rn2 = torch.zeros(B, dtype=dt, device=DEV)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        st = torch.zeros(B, dtype=torch.uint8, device=DEV)
        be.solve_lin((B, T, nx, nu), sel(p.Qd), sel(p.q), sel(p.F), sel(p.c), sel(p.x0), p.u_lo, p.u_hi,
                     0, 0, z, lam, rho, phi, rn2, info, st, al_iter=2, max_newton=4, n_ls=20, flags=3)
        torch.cuda.synchronize()
        return z, lam, rn2, info, st

    z, lam, rn2, info, st = solve()
    assert int(info.abs().sum()) == 0 and int(st.sum()) == B
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    z2, lam2, _, _, _ = solve(perm)
    assert torch.equal(z2, z[perm]) and torch.equal(lam2, lam[perm])
    # start residual (z0 = xref violates dynamics) vs end residual
    xn0 = torch.einsum("btij,btj->bti", p.F, p.z0[:, :-1]) + p.c
    r0 = ((p.z0[:, 1:, :nx] - xn0) ** 2).sum((1, 2)) + ((p.z0[:, 0, :nx] - p.x0) ** 2).sum(1)
    assert float(rn2.mean()) < 0.2 * float(r0.mean())
    # sample vs oracle
    idx = np.arange(0, B, B // 32)
    c = lambda a: a[idx].cpu().numpy()
    o = orc.solve_lin("f32", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), p.u_lo.cpu().numpy(),
                      p.u_hi.cpu().numpy(), c(p.z0), al_iter=2, exit_mode="fixed")
    assert np.abs(c(z) - o["z"]).max() < 5e-3


@pytest.mark.parametrize("variant", ["team", "quad"])
@pytest.mark.parametrize("nx,nu", [(2, 1), (4, 1), (4, 2), (6, 1), (6, 2), (8, 2), (10, 3), (12, 4), (13, 4), (14, 4)])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_every_compiled_dims_vs_oracle(nx, nu, dtype, variant):
    """Every (nx,nu) instance in the library, ragged batch (B not a multiple of the
    instances per wavefront), odd horizon. Caught a row-stride overflow for n % 4 == 0."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    dt = TD[dtype]
    B, T = 19, 7
    p = synthetic_problem(B, T, nx, nu, seed=5, dtype=dt, device=DEV)
    M = T * nx + 2 * T * nu
    z = p.z0.clone()
    lam = torch.zeros(B, M, dtype=dt, device=DEV)
    rho = torch.ones(B, dtype=dt, device=DEV)
    phi = torch.zeros(B, dtype=dt, device=DEV)
    rn2 = torch.zeros(B, dtype=dt, device=DEV)
    info = torch.zeros(B, dtype=torch.int32, device=DEV)
    st = torch.zeros(B, dtype=torch.uint8, device=DEV)
    be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2,
                 info, st, al_iter=2, max_newton=4, n_ls=20, flags=3, variant=variant)
    torch.cuda.synchronize()
    c = lambda a: a.cpu().numpy()
    o = orc.solve_lin(dtype, c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0),
                      al_iter=2, exit_mode="fixed", trace_steps=8)
    tol = 1e-10 if dtype == "f64" else 2e-3
    assert int(info.abs().sum()) == 0
    ok = ~near_tie_instances(o, dtype)
    prob = dict(Qd=c(p.Qd), q=c(p.q), F=c(p.F), c=c(p.c), x0=c(p.x0), u_lo=c(p.u_lo), u_hi=c(p.u_hi))
    budget = (NEAR_TIES_DIMS_F32[(nx, nu)] if dtype == "f32" else 0) + 1
    check_excluded(dtype, ~ok, c(z), c(lam), c(rho), o, prob, max_count=budget, label=f"({nx},{nu})/{dtype}/{variant}")
    assert np.abs(c(z) - o["z"])[ok].max() < tol
    assert np.abs(c(lam) - o["lam"])[ok].max() < tol * 20


@pytest.mark.parametrize("variant", ["team", "quad"])
def test_long_horizon_T50(variant):
    """BASELINE config 5 shape (nx=13, nu=4, T=50): a small batch against the oracle, fp32 and fp64."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    for dtype in ("f64", "f32"):
        dt = TD[dtype]
        B, T, nx, nu = 33, 50, 13, 4
        p = synthetic_problem(B, T, nx, nu, seed=9, dtype=dt, device=DEV)
        M = T * nx + 2 * T * nu
        z = p.z0.clone()
        lam = torch.zeros(B, M, dtype=dt, device=DEV)
        rho = torch.ones(B, dtype=dt, device=DEV)
        phi = torch.zeros(B, dtype=dt, device=DEV)
        rn2 = torch.zeros(B, dtype=dt, device=DEV)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        st = torch.zeros(B, dtype=torch.uint8, device=DEV)
        be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2,
                     info, st, al_iter=2, max_newton=4, n_ls=20, flags=3, variant=variant)
        torch.cuda.synchronize()
        c = lambda a: a.cpu().numpy()
        o = orc.solve_lin(dtype, c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_lo), c(p.u_hi), c(p.z0),
                          al_iter=2, exit_mode="fixed", trace_steps=8)
        ok = ~near_tie_instances(o, dtype)
        assert int(info.abs().sum()) == 0
        prob = dict(Qd=c(p.Qd), q=c(p.q), F=c(p.F), c=c(p.c), x0=c(p.x0), u_lo=c(p.u_lo), u_hi=c(p.u_hi))
        check_excluded(dtype, ~ok, c(z), c(lam), c(rho), o, prob, max_count=NEAR_TIES_T50[dtype] + 1, label=f"T50/{dtype}/{variant}")
        assert np.abs(c(z) - o["z"])[ok].max() < (1e-9 if dtype == "f64" else 3e-3)


def test_variants_agree_at_headline_size():
    """team and quad kernels on the full benchmark batch: same answers up to fp32 rounding
    on all but the few instances where a line-search near-tie resolves differently."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = 16384, 20, 13, 4
    dt = torch.float32
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=DEV)
    M = T * nx + 2 * T * nu
    res = {}
    for variant in ("team", "quad"):
        z = p.z0.clone()
        lam = torch.zeros(B, M, dtype=dt, device=DEV)
        rho = torch.ones(B, dtype=dt, device=DEV)
        phi = torch.zeros(B, dtype=dt, device=DEV)
        rn2 = torch.zeros(B, dtype=dt, device=DEV)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        st = torch.zeros(B, dtype=torch.uint8, device=DEV)
        be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, z, lam, rho, phi, rn2,
                     info, st, al_iter=2, max_newton=4, n_ls=20, flags=3, variant=variant)
        torch.cuda.synchronize()
        assert int(info.abs().sum()) == 0 and int(st.sum()) == B
        res[variant] = z
    err = (res["team"] - res["quad"]).abs().amax(dim=(1, 2))
    assert float((err < 1e-3).float().mean()) > 0.97
    assert float(err.median()) < 2e-5


@pytest.mark.parametrize("exit_mode", ["fixed", "reference"])
def test_backward_quad_workspace_equals_team_factor(exit_mode):
    """Gradients w.r.t. q and diag(Q) through the two backward routes - the quad solve's
    workspace as factor (alqp_backward_ws) and the team kernel's packed factor (alqp_backward) -
    agree on a (13,4) batch; both implement NewtonAL.backward (al_utils.py:578-615)."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem
    from deq_mpc_corl_amd.backend import HipBackend

    dt = torch.float64
    B, T, nx, nu = 37, 9, 13, 4
    p = synthetic_problem(B, T, nx, nu, seed=31, dtype=dt, device=DEV, active=True)
    g = torch.Generator(device="cpu").manual_seed(2)
    wx = torch.randn(B, T, nx, generator=g).to(DEV)
    wu = torch.randn(B, T, nu, generator=g).to(DEV)
    grads = {}
    for name, be in (("quad", HipBackend()), ("team", None)):
        if be is None:
            be = HipBackend()
            be.default_variant = "team"
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode=exit_mode, backend=be)
        if name == "team":
            # hide the quad route from the host logic (hasattr check)
            class _B:  # thin proxy without backward_ws
                def __init__(self, inner): self._i = inner
                def __getattr__(self, k):
                    if k == "backward_ws": raise AttributeError(k)
                    return getattr(self._i, k)
            mpc._backend = _B(be)
        mpc.reinitialize(p.x0, None)
        mpc.al_iter = 2
        Qd = p.Qd.clone().requires_grad_(True)
        q = p.q.clone().requires_grad_(True)
        dyn = AffineDynamics(p.F, p.c)
        cost = QuadCost(torch.diag_embed(Qd), q, torch.zeros(B, T, device=DEV, dtype=dt))
        x, u, _ = mpc(p.x0, cost, dyn, dyn.jac, x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())
        ((x * wx).sum() + (u * wu).sum()).backward()
        grads[name] = (q.grad.clone(), Qd.grad.clone(), x.detach().clone())
    assert torch.allclose(grads["quad"][2], grads["team"][2], atol=1e-5)
    sq = float(grads["team"][0].abs().max())
    assert float((grads["quad"][0] - grads["team"][0]).abs().max()) < 1e-6 * sq
    assert float((grads["quad"][1] - grads["team"][1]).abs().max()) < 1e-6 * float(grads["team"][1].abs().max())


@pytest.mark.parametrize("cfg", ["pendulum T=5 B=4096 f32 (BASELINE config 2)", "(8,2) T=10 B=8192 f32 (config 3)",
                                 "(13,4) T=50 B=65536 f32 (config 5, whole on one GPU)"])
def test_baseline_configs_full_size_properties(cfg):
    """BASELINE.json's other configurations at FULL size through the default ('auto' -> quad) path, checked by
    size-independent properties (no oracle run of that size): every instance ok, bit-exact invariance
    under a permutation of the batch, the AL solve reduces the constraint violation, idempotent re-run
    (same inputs -> same bits), plus a 32-instance sample against the oracle."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import default_backend
    be = default_backend()
    B, T, nx, nu = {"pend": (4096, 5, 2, 1), "(8,2": (8192, 10, 8, 2), "(13,": (65536, 50, 13, 4)}[cfg[:4]]
    dt = torch.float32
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=DEV)
    M = T * nx + 2 * T * nu

    def solve(perm=None):
        sel = (lambda a: a) if perm is None else (lambda a: a[perm].contiguous())
        z = sel(p.z0).clone()
        lam = torch.zeros(B, M, dtype=dt, device=DEV)
        rho = torch.ones(B, dtype=dt, device=DEV)
        phi = torch.zeros(B, dtype=dt, device=DEV)
        rn2 = torch.zeros(B, dtype=dt, device=DEV)
        info = torch.zeros(B, dtype=torch.int32, device=DEV)
        st = torch.zeros(B, dtype=torch.uint8, device=DEV)
        be.solve_lin((B, T, nx, nu), sel(p.Qd), sel(p.q), sel(p.F), sel(p.c), sel(p.x0), p.u_lo, p.u_hi,
                     0, 0, z, lam, rho, phi, rn2, info, st, al_iter=2, max_newton=4, n_ls=20, flags=3)
        torch.cuda.synchronize()
        return z, lam, rn2, info, st

    z, lam, rn2, info, st = solve()
    assert be.last_variant == "quad"
    assert int(info.abs().sum()) == 0 and int(st.sum()) == B
    z1, lam1, _, _, _ = solve()
    assert torch.equal(z1, z) and torch.equal(lam1, lam)                      # idempotent
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    z2, lam2, _, _, _ = solve(perm)
    assert torch.equal(z2, z[perm]) and torch.equal(lam2, lam[perm])          # position in the batch does not matter
    xn0 = torch.einsum("btij,btj->bti", p.F, p.z0[:, :-1]) + p.c
    r0 = ((p.z0[:, 1:, :nx] - xn0) ** 2).sum((1, 2)) + ((p.z0[:, 0, :nx] - p.x0) ** 2).sum(1)
    assert float(rn2.mean()) < 0.35 * float(r0.mean())
    idx = np.arange(0, B, B // 32)
    c = lambda a: a[idx].cpu().numpy()
    o = orc.solve_lin("f32", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), p.u_lo.cpu().numpy(),
                      p.u_hi.cpu().numpy(), c(p.z0), al_iter=2, exit_mode="fixed", trace_steps=8)
    ok = ~near_tie_instances(o, "f32")
    prob = dict(Qd=c(p.Qd), q=c(p.q), F=c(p.F), c=c(p.c), x0=c(p.x0), u_lo=p.u_lo.cpu().numpy(), u_hi=p.u_hi.cpu().numpy())
    check_excluded("f32", ~ok, c(z), c(lam), np.full(len(idx), 100.0, np.float32), o, prob, max_count=NEAR_TIES_FULL[cfg[:4]] + 1, label=cfg)
    assert np.abs(c(z) - o["z"])[ok].max() < 5e-3
