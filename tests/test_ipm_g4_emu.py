"""The register/LDS-resident interior-point kernel (deq-mpc-corl_amd/csrc/alqp_ipm_g4.hpp) executed in the CPU
wave emulator (tests/emu: the SAME kernel source on a 64-lane execution policy, cross-lane primitives with their
gfx950 semantics) against the fixtures made by RUNNING the reference's qp_wrapper.MPC (tools/gen_golden_ip.py) and
against the C oracle. This is how the kernel's lane-level logic - stage-owned register layout, row-broadcast
mat-vecs, block LDL' sweeps, the workspace hand-over between the launches of the reference exit mode - is checked
where no GPU is present; tests/test_ip_golden.py runs the same fixtures through the real kernel (-m gpu)."""
import numpy as np
import pytest

from tests import golden_util as gu

IP_LIN = ["ip_pend_f64", "ip_cart_f64", "ip_cart_active_f64", "ip_quad13_f64", "ip_quad13_active_f64",
          "ip_quad12_f64", "ip_fcp14_f64", "ip_pend_f32", "ip_cart_f32", "ip_quad13_f32"]


def _bm(a):
    return np.ascontiguousarray(np.swapaxes(a, 0, 1))


@pytest.mark.parametrize("name", IP_LIN)
@pytest.mark.parametrize("exit_mode", ["reference", "fixed"])
def test_resident_kernel_emulated_vs_reference_fixture(name, exit_mode):
    from tests.emu import ipm_g4_emu_py as emu
    g = gu.load(name)
    dt = str(g["dtype"])
    f64 = dt == "f64"
    o = emu.forward(dt, _bm(g["Cd"]), _bm(g["c"]), _bm(g["F"]), _bm(g["f"]), g["x0"], g["u_hi"], g["u_lo"],
                    exit_mode=exit_mode)
    if f64 and exit_mode == "reference":
        assert o["iters"] == int(g["ipm_iters"][0])          # the reference's batch-global exit fires in the same iteration
    tol = 1e-8 if f64 else 2e-3
    for k in ("zhat", "nus", "lams", "slacks"):
        ref = g["qp_" + k][0]
        err = float(np.abs(o[k] - ref).max() / max(1.0, np.abs(ref).max()))
        assert err < tol, (k, err)
    if f64:
        assert int(o["info"].max()) == 0


@pytest.mark.parametrize("name", ["ip_cart_f64", "ip_quad13_active_f64"])
def test_resident_kernel_emulated_equals_oracle_every_launch_mode(name):
    """One launch for the whole solve == INIT, (RESID, STEP) x n, FINAL through the workspace, and both agree with
    the C oracle's structured solver to rounding."""
    from oracle import ipm_py
    from tests.emu import ipm_g4_emu_py as emu
    g = gu.load(name)
    args = (_bm(g["Cd"]), _bm(g["c"]), _bm(g["F"]), _bm(g["f"]), g["x0"], g["u_hi"], g["u_lo"])
    one = emu.forward("f64", *args, exit_mode="fixed")
    s = emu.Solve("f64", *args)
    s.launch(emu.INIT)
    for it in range(20):
        s.launch(emu.RESID, 0, it)
        s.launch(emu.STEP, 0, it)
    many = s.launch(emu.FINAL)
    orc = ipm_py.forward("f64", *args, solver=0, exit_mode=1)
    for k in ("zhat", "nus", "lams", "slacks"):
        assert np.array_equal(one[k], many[k]), k
        assert np.abs(one[k] - orc[k]).max() < 1e-12 * max(1.0, np.abs(orc[k]).max()), k


@pytest.mark.parametrize("name", ["ip_pend_f64", "ip_quad13_active_f64", "ip_fcp14_f64"])
def test_resident_backward_kernel_emulated_vs_oracle(name):
    from oracle import ipm_py
    from tests.emu import ipm_g4_emu_py as emu
    g = gu.load(name)
    gbar = np.random.default_rng(0).standard_normal(g["qp_zhat"][0].shape)
    a = ipm_py.backward("f64", _bm(g["Cd"]), _bm(g["F"]), g["qp_lams"][0], g["qp_slacks"][0], gbar, solver=0)
    b = emu.backward("f64", _bm(g["Cd"]), _bm(g["F"]), g["qp_lams"][0], g["qp_slacks"][0], gbar)
    for u, v in zip(a, b):
        assert np.abs(u - v).max() < 1e-8 * max(1.0, np.abs(u).max())


@pytest.mark.parametrize("T", [2, 3, 4, 5, 6, 9, 12, 19, 20])
@pytest.mark.parametrize("dims", [(13, 4), (4, 2), (14, 4), (2, 1)])
def test_resident_kernel_emulated_every_horizon(T, dims):
    """The two elimination chains of the twisted factorisation have equal length for odd T and differ by one block for
    even T (the bottom chain then starts one step late), and the smallest horizons have an empty bottom chain (T = 2)
    or single-block chains (T = 3): every case against the C oracle's one-directional structured solver, forward
    (two instances, one with active bounds) and backward."""
    import torch
    from deq_mpc_corl_amd import synthetic_problem
    from oracle import ipm_py
    from tests.emu import ipm_g4_emu_py as emu
    nx, nu = dims
    if T in (5, 6, 9, 12, 19) and dims != (13, 4):
        pytest.skip("the intermediate horizons run on the flagship size only (emulator time)")
    p = synthetic_problem(2, T, nx, nu, seed=T, dtype=torch.float64, device="cpu")
    c = lambda a: a.numpy()
    args = (c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_hi) * 0.2, c(p.u_lo) * 0.2)
    got = emu.forward("f64", *args, exit_mode="fixed")
    orc = ipm_py.forward("f64", *args, solver=0, exit_mode=1)
    assert int(got["info"].max()) == 0
    for k in ("zhat", "nus", "lams", "slacks"):
        assert np.abs(got[k] - orc[k]).max() < 1e-9 * max(1.0, np.abs(orc[k]).max()), k
    gbar = np.random.default_rng(T).standard_normal(orc["zhat"].shape)
    a = ipm_py.backward("f64", args[0], args[2], orc["lams"], orc["slacks"], gbar, solver=0)
    b = emu.backward("f64", args[0], args[2], orc["lams"], orc["slacks"], gbar)
    for u, v in zip(a, b):
        assert np.abs(u - v).max() < 1e-8 * max(1.0, np.abs(u).max())


def test_resident_kernel_emulated_reports_a_non_positive_pivot():
    """A stage with negative curvature makes a pivot of the Schur complement non-positive: the kernel replaces it by
    its magnitude and reports the first one as block * nx + column + 1, like the oracle. The top chain of the twisted
    factorisation meets exactly the pivots of the oracle's top-down elimination, so a bad stage in the first half gives
    the same index; a bad stage in the second half is met from below, through different (equally valid) pivots - the
    flag is raised, its index is the twisted order's own."""
    import torch
    from deq_mpc_corl_amd import synthetic_problem
    from oracle import ipm_py
    from tests.emu import ipm_g4_emu_py as emu
    T, nx, nu = 12, 4, 2
    p = synthetic_problem(3, T, nx, nu, seed=3, dtype=torch.float64, device="cpu")
    Qd = p.Qd.numpy().copy()
    Qd[1, 3, :nx] = -50.0        # instance 1: stage 3 (top chain)
    Qd[2, 9, :nx] = -50.0        # instance 2: stage 9 (bottom chain)
    c = lambda a: a.numpy()
    args = (Qd, c(p.q), c(p.F), c(p.c), c(p.x0), c(p.u_hi), c(p.u_lo))
    got = emu.forward("f64", *args, exit_mode="fixed", max_iter=2)
    orc = ipm_py.forward("f64", *args, solver=0, exit_mode=1, max_iter=2)
    assert int(got["info"][0]) == 0 and int(orc["info"][0]) == 0
    assert int(got["info"][1]) == int(orc["info"][1]) != 0
    assert int(got["info"][2]) != 0 and int(orc["info"][2]) != 0
