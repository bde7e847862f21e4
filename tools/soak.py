#!/usr/bin/env python3
"""Randomised soak of the fused LinDx solve on the GPU against the CPU oracle (fp64, both kernel
variants): random compiled sizes, horizons, batch sizes (ragged last wavefront), active bounds,
AL depth, per-instance bounds. Prints the worst deviations; exits non-zero on a failure.

    gpurun -- python tools/soak.py 120
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deq_mpc_corl_amd import synthetic_problem
from deq_mpc_corl_amd.backend import default_backend
from oracle import oracle_py as orc

DIMS = [(2, 1), (4, 1), (4, 2), (6, 1), (6, 2), (8, 2), (10, 3), (12, 4), (13, 4), (14, 4)]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
be = default_backend()
dev, dt = "cuda:0", torch.float64
c = lambda a: a.cpu().numpy()
worst = {"team": 0.0, "quad": 0.0}
fails = 0
for case in range(n_cases):
    nx, nu = DIMS[rng.integers(len(DIMS))]
    T = int(rng.integers(2, 13))
    B = int(rng.integers(1, 71))
    active = bool(rng.integers(2))
    al_iter = int(rng.integers(1, 4))
    seed = int(rng.integers(1 << 30))
    p = synthetic_problem(B, T, nx, nu, seed=seed, dtype=dt, device=dev, active=active)
    per_inst = bool(rng.integers(2))
    if per_inst:
        g = torch.Generator().manual_seed(seed)
        hi = (0.05 + 0.5 * torch.rand(B, T, nu, generator=g, dtype=dt)).to(dev).contiguous()
        lo = (-(0.05 + 0.5 * torch.rand(B, T, nu, generator=g, dtype=dt))).to(dev).contiguous()
        sb, st = T * nu, nu
    else:
        hi, lo, sb, st = p.u_hi, p.u_lo, 0, 0
    o = orc.solve_lin("f64", c(p.Qd), c(p.q), c(p.F), c(p.c), c(p.x0), c(lo), c(hi), c(p.z0), al_iter=al_iter, exit_mode="fixed")
    M = T * nx + 2 * T * nu
    for variant in ("team", "quad"):
        z = p.z0.clone()
        lam = torch.zeros(B, M, dtype=dt, device=dev)
        rho = torch.ones(B, dtype=dt, device=dev)
        phi = torch.zeros(B, dtype=dt, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.uint8, device=dev)
        be.solve_lin((B, T, nx, nu), p.Qd, p.q, p.F, p.c, p.x0, lo, hi, sb, st, z, lam, rho, phi, None, info, status,
                     al_iter=al_iter, max_newton=4, n_ls=20, flags=3, variant=variant)
        torch.cuda.synchronize()
        scale = 1.0 + np.abs(o["z"]).max()
        ez = np.abs(c(z) - o["z"]).max() / scale
        el = np.abs(c(lam) - o["lam"]).max() / (1.0 + np.abs(o["lam"]).max())
        worst[variant] = max(worst[variant], ez, el)
        ok = ez < 1e-7 and el < 1e-6 and int(status.sum()) == B and int((info != 0).sum()) == 0
        if not ok:
            fails += 1
            print(f"FAIL case {case}: dims ({nx},{nu}) T={T} B={B} active={active} al={al_iter} per_inst={per_inst} seed={seed} "
                  f"{variant}: ez={ez:.2e} el={el:.2e} status_ok={int(status.sum())}/{B} info_bad={int((info != 0).sum())}")
# ---- the drop-in class in the default (reference exit) mode: device-side exit test, primed
#      workspaces, host state carry, against the same host logic on the test-only CPU backend
from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
from tests.oracle_backend import OracleBackend
n_mpc = max(4, n_cases // 6)
worst_mpc = 0.0
for case in range(n_mpc):
    nx, nu = DIMS[rng.integers(len(DIMS))]
    T, B = int(rng.integers(3, 10)), int(rng.integers(2, 40))
    active = bool(rng.integers(2))
    seed = int(rng.integers(1 << 30))
    res = {}
    for name, device, backend, variant in (("gpu-quad", dev, None, "quad"), ("gpu-auto", dev, None, "auto"),
                                           ("cpu", "cpu", OracleBackend(), None)):
        if variant:
            be.default_variant = variant
        p = synthetic_problem(B, T, nx, nu, seed=seed, dtype=dt, device=device, active=active)
        mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, backend=backend)
        mpc.reinitialize(p.x0, None)
        dyn = AffineDynamics(p.F, p.c)
        cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, dtype=dt, device=device))
        outs = []
        for call in range(2):                 # second call carries lam / rho / warm start over
            mpc.al_iter = 2
            x, u, _ = mpc(p.x0, cost, dyn, dyn.jac, **({} if call else dict(x_init=p.z0[..., :nx].clone(), u_init=p.z0[..., nx:].clone())))
            outs.append((x.double().cpu().numpy(), u.double().cpu().numpy(), list(mpc.last_newton_per_al)))
        res[name] = (outs, mpc.lamda_prev.cpu().numpy(), mpc.rho_prev.cpu().numpy())
    be.default_variant = "auto"
    for name in ("gpu-quad", "gpu-auto"):
        ok = all(a[2] == b[2] for a, b in zip(res[name][0], res["cpu"][0]))
        e = max(max(np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max()) for a, b in zip(res[name][0], res["cpu"][0]))
        el = np.abs(res[name][1] - res["cpu"][1]).max() / (1.0 + np.abs(res["cpu"][1]).max())
        worst_mpc = max(worst_mpc, e, el)
        if not (ok and e < 1e-5 and el < 1e-6 and np.array_equal(res[name][2], res["cpu"][2])):
            fails += 1
            print(f"FAIL mpc case {case}: dims ({nx},{nu}) T={T} B={B} active={active} seed={seed} {name}: counts "
                  f"{[a[2] for a in res[name][0]]} vs {[b[2] for b in res['cpu'][0]]} e={e:.2e} el={el:.2e}")
print(f"{n_mpc} MPC cases (2 calls each, reference exit) x 2 GPU variants vs CPU host logic: worst deviation {worst_mpc:.2e}")
print(f"{n_cases} cases x 2 variants: worst relative deviation team {worst['team']:.2e}, quad {worst['quad']:.2e}; failures: {fails}")
sys.exit(1 if fails else 0)
