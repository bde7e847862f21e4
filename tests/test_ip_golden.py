"""Interior-point path against fixtures made by RUNNING the reference's
`qpth.qp_wrapper.MPC` (tools/gen_golden_ip.py).

  * oracle level (not gpu): oracle/ipm_oracle.c - both its literal dense-LU KKT solver and the structured
    elimination the HIP kernel implements - against the recorded initial point, iteration count and
    returned (zhat, nus, lams, slacks) of `pdipm_b_LU.forward`, and the backward KKT solve;
  * drop-in level: `deq_mpc_corl_amd.qpth.qp_wrapper.MPC` replayed on the CPU with the TEST-ONLY oracle
    backend (host logic: linearisation, rollout line search, SQP loop, autograd) and on the MI355X through
    the C ABI (`-m gpu`): x, u, line-search alpha, IPM iteration count, gradients w.r.t. C and c.
"""
import numpy as np
import pytest
import torch

from tests import golden_util as gu

TD = {"f64": torch.float64, "f32": torch.float32}
IP_LIN = ["ip_pend_f64", "ip_cart_f64", "ip_cart_active_f64", "ip_quad13_f64", "ip_quad13_active_f64",
          "ip_quad12_f64", "ip_fcp14_f64", "ip_pend_f32", "ip_cart_f32", "ip_quad13_f32"]
IP_ALL = IP_LIN + ["ip_cart_sqp3_f64", "ip_pend_nonlin_f64", "ip_pend_nonlin_sqp3_f64"]


def _load(name):
    g = gu.load(name)
    for k in ("qp_iter", "n_solves"):
        g[k] = int(g[k])
    g["kind"] = str(g["kind"])
    return g


def _bm(a):
    return np.ascontiguousarray(np.swapaxes(a, 0, 1))


@pytest.mark.parametrize("name", IP_LIN)
@pytest.mark.parametrize("solver", ["dense LU (literal)", "structured elimination (the kernel's)"])
def test_ipm_oracle_vs_reference(name, solver):
    from oracle import ipm_py
    g = _load(name)
    dense = solver.startswith("dense")
    if dense and g["T"] * (g["nx"] + g["nu"]) > 120:
        pytest.skip("dense KKT of order > 400: minutes of scalar LU; the structured solver covers this size")
    o = ipm_py.forward(g["dtype"], _bm(g["Cd"]), _bm(g["c"]), _bm(g["F"]), _bm(g["f"]), g["x0"], g["u_hi"], g["u_lo"],
                       solver=1 if dense else 0)
    f64 = g["dtype"] == "f64"
    e = lambda k, r: float(np.abs(o[k] - r).max() / max(1.0, np.abs(r).max()))
    tol_i, tol = (1e-11, 1e-8) if f64 else (1e-4, 2e-3)   # fp32: both sides carry the rounding of ~20 KKT solves at cond ~1e7
    for k in ("x", "s", "z", "y"):
        assert e("init_" + k, g["qp_init_" + k][0]) < tol_i, k
    if f64:
        assert abs(o["iters"] - int(g["ipm_iters"][0])) <= (1 if dense else 0)
    for k in ("zhat", "nus", "lams", "slacks"):
        assert e(k, g["qp_" + k][0]) < tol, (k, e(k, g["qp_" + k][0]))
    if f64:   # fp32: the Schur complement loses definiteness at the noise floor of the last iterations (|p| policy, flagged)
        assert int(o["info"].max()) == 0


@pytest.mark.parametrize("name", ["ip_pend_f64", "ip_cart_f64"])
def test_ipm_oracle_structured_equals_dense_backward(name):
    """The backward KKT solve (qp.py:243-252) by both solvers at the reference's returned iterate."""
    from oracle import ipm_py
    g = _load(name)
    rng = np.random.default_rng(0)
    gbar = rng.standard_normal(g["qp_zhat"][0].shape)
    a = ipm_py.backward("f64", _bm(g["Cd"]), _bm(g["F"]), g["qp_lams"][0], g["qp_slacks"][0], gbar, solver=1)
    b = ipm_py.backward("f64", _bm(g["Cd"]), _bm(g["F"]), g["qp_lams"][0], g["qp_slacks"][0], gbar, solver=0)
    for u, v in zip(a, b):
        assert np.abs(u - v).max() < 1e-7 * max(1.0, np.abs(u).max())


def _replay_ip(name, backend, dev, exit_mode="reference", variant=None):
    from deq_mpc_corl_amd import PendulumDynamics
    from deq_mpc_corl_amd.qpth import qp_wrapper as ip
    g = _load(name)
    dt = TD[g["dtype"]]
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    tt = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev)
    Cd, c = tt(g["Cd"]), tt(g["c"])
    with_grad = "bwd_c_grad" in g
    if with_grad:
        Cd.requires_grad_(True)
        c.requires_grad_(True)
    if g["kind"] == "lindx":
        dx, dx_jac = ip.LinDx(tt(g["F"]), tt(g["f"])), None
    else:
        dyn = PendulumDynamics()
        dx, dx_jac = dyn, dyn.jac
    if variant is not None:   # which interior-point kernel the HIP backend launches (AlqpIpmParams.variant)
        from deq_mpc_corl_amd.backend import HipBackend
        backend = HipBackend()
        backend.ipm_variant = variant
    mpc = ip.MPC(nx, nu, T, u_lower=tt(g["u_lo"]), u_upper=tt(g["u_hi"]), qp_iter=g["qp_iter"], exit_unconverged=False,
                 eps=1e-5, n_batch=B, backprop=False, verbose=0, u_init=tt(g["u_init"]),
                 grad_method=ip.GradMethods.ANALYTIC, solver_type="dense", single_qp_solve=(g["qp_iter"] == 1),
                 exit_mode=exit_mode, backend=backend)
    x, u = mpc(tt(g["x0"]), ip.QuadCost(torch.diag_embed(Cd), c), dx, dx_jac)
    assert x.shape == (T, B, nx) and u.shape == (T, B, nu)
    f64 = g["dtype"] == "f64"
    # fp32: ~20 KKT solves at a conditioning of ~1e7 (D~ = z/s spans 1e-7..1e7 near convergence) leave the returned
    # controls at a noise floor of a few 1e-3 on both sides (the reference's fp32 run and ours): measured on the MI355X
    # 4.5e-3 on ip_quad13_f32 with the register-resident kernel (hardware reciprocals), 1e-6 on the two small fixtures;
    # the same kernel source with exact divisions (CPU wave emulator, tests/test_ipm_g4_emu.py) is within 2e-5
    tol = 1e-7 if f64 else 1e-2
    if f64 and exit_mode == "reference":
        assert mpc.last_ipm["iters"] == int(g["ipm_iters"][-1])
    ex = np.abs(x.detach().cpu().numpy() - g["x"]).max()
    eu = np.abs(u.detach().cpu().numpy() - g["u"]).max()
    assert ex < tol * max(1.0, np.abs(g["x"]).max()) and eu < tol, (ex, eu)
    if g["qp_iter"] == 1:
        # (in the SQP fixtures the last line search runs at a converged point, where "did the rollout cost
        #  strictly decrease" is decided by the last bits of the QP solution - one instance of ip_pend_nonlin_sqp3
        #  takes 8e-3 instead of 1 on a step of ~1e-9 with the register-resident kernel, whose sums round in another
        #  order than the reference's LU; x and u, compared above, are what that decision can move)
        assert np.allclose(mpc.last_alpha.cpu().numpy(), g["alpha"][-1])
    if with_grad:
        loss = (x * tt(g["bwd_wx"])).sum() + (u * tt(g["bwd_wu"])).sum()
        loss.backward()
        for got, want in ((c.grad, g["bwd_c_grad"]), (Cd.grad, g["bwd_Cd_grad"])):
            err = np.abs(got.cpu().numpy() - want).max()
            assert err < 1e-5 * max(1.0, np.abs(want).max()), err
    return mpc


@pytest.mark.parametrize("name", IP_ALL)
def test_ip_dropin_host_logic_cpu(name):
    from tests.oracle_backend import OracleBackend
    _replay_ip(name, OracleBackend(), "cpu")


@pytest.mark.parametrize("name", ["ip_cart_f64", "ip_quad13_active_f64"])
def test_ip_fixed_mode_matches_reference_result_cpu(name):
    """exit_mode='fixed' (all 20 iterations, best iterate per instance): the reference stops earlier on its
    batch-global rule, but the extra iterations only ever replace an instance's best iterate by a better
    one, so the returned point agrees to the IPM's own accuracy."""
    from tests.oracle_backend import OracleBackend
    _replay_ip(name, OracleBackend(), "cpu", exit_mode="fixed")


@pytest.mark.gpu
@pytest.mark.parametrize("name", IP_ALL)
@pytest.mark.parametrize("exit_mode", ["reference", "fixed"])
@pytest.mark.parametrize("variant", ["resident", "generic_lds", "generic_ws"])
def test_ip_dropin_hip(name, exit_mode, variant):
    """Every reference-generated fixture through every kernel the library can launch for it: the
    register/LDS-resident kernel ("auto" picks it for these horizons) and the size-generic kernel with the
    Schur factor in LDS and in the workspace (the placement the generic kernel takes at B >= 2048 in fp64)."""
    _replay_ip(name, None, "cuda:0", exit_mode=exit_mode, variant=variant)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("variant", ["resident", "generic_ws", "generic_lds"])
def test_ip_at_scale_properties(dtype, variant):
    """The configuration bench.py times (B = 8192, (T, nx, nu) = (20, 13, 4), exit mode "fixed"), on every kernel:
    a 32-instance sample against the CPU oracle (the reference-pinned restatement), idempotence (same launch
    twice, bit-identical), permutation invariance (instances are independent: reversing the batch reverses the
    result bit for bit), and the kernel's own best residual at the interior-point method's floor."""
    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd.backend import HipBackend
    from oracle import ipm_py
    B, T, nx, nu = 8192, 20, 13, 4
    dt = TD[dtype]
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device="cuda:0")
    be = HipBackend()
    tm = lambda a: a.transpose(0, 1).contiguous()
    run = lambda q: be.ipm_solve((B, T, nx, nu), *q, p.u_hi, p.u_lo, exit_mode="fixed", variant=variant)
    args = (tm(p.Qd), tm(p.q), tm(p.F), tm(p.c), p.x0)
    o1 = run(args)
    keys = ("zhat", "nus", "lams", "slacks")
    r1 = {k: o1[k].clone() for k in keys + ("resid", "info")}
    o2 = run(args)
    for k in keys:
        assert torch.equal(r1[k], o2[k]), k                      # idempotent (workspace contents do not matter)
    flip = lambda a, dim: a.flip(dim).contiguous()
    o3 = run((flip(args[0], 1), flip(args[1], 1), flip(args[2], 1), flip(args[3], 1), flip(args[4], 0)))
    for k in keys:
        assert torch.equal(r1[k], o3[k].flip(0)), k              # instances do not interact
    f64 = dtype == "f64"
    if f64:
        assert int((r1["info"] != 0).sum()) == 0
    assert float(r1["resid"].max()) < (1e-10 if f64 else 5e-2), float(r1["resid"].max())
    sel = torch.arange(0, B, B // 32, device="cuda:0")[:32]
    c = lambda a: a.index_select(0, sel).cpu().numpy()
    o = ipm_py.forward(dtype, c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), p.u_hi.cpu().numpy(), p.u_lo.cpu().numpy(),
                       solver=0, exit_mode=1)
    # fp64: both sides converge to the same point far below the tolerance; fp32: ~20 KKT solves at cond ~1e7 leave the
    # controls at a noise floor of a few 1e-3 on both sides (see _replay_ip)
    tol = 1e-8 if f64 else 1e-2
    for k in ("zhat", "lams", "slacks"):
        got = r1[k].index_select(0, sel).cpu().numpy()
        err = np.abs(got - o[k]).max() / max(1.0, np.abs(o[k]).max())
        assert err < tol, (k, err)


def test_policies_import_surface_resolves():
    """deqmpc/policies.py:5-8 imports qpth.qp_wrapper, qpth.AL_mpc, qpth.AL_mpc_custom.Obstacle_MPC,
    qpth.al_utils - all of them exist in the shim package."""
    import importlib
    for mod in ("qp_wrapper", "AL_mpc", "al_utils"):
        importlib.import_module("deq_mpc_corl_amd.qpth." + mod)
    from deq_mpc_corl_amd.qpth import qp_wrapper as ip
    assert {"MPC", "QuadCost", "GradMethods", "LinDx"} <= set(dir(ip))
    assert ip.GradMethods.ANALYTIC.value == 3
