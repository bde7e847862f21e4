#!/usr/bin/env python3
"""Throughput of the quadrotor / flying-cartpole providers (value + Jacobian launch), and of one Newton direction with
obstacle rows at B = 4096 on the team and on the quad step kernel.  gpurun -- python tools/bench_dyn_rigid.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deq_mpc_corl_amd import FlyingCartpoleDynamics, RexQuadrotorDynamics, synthetic_problem
from deq_mpc_corl_amd.backend import default_backend

dev = "cuda:0"
out = {}


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, dyn, nx in (("rex_quadrotor", RexQuadrotorDynamics(), 12), ("flying_cartpole", FlyingCartpoleDynamics(), 14)):
    for dt in (torch.float64, torch.float32):
        K = 1 << 20
        x = 0.3 * torch.randn(K, nx, dtype=dt, device=dev)
        u = 0.2 * torch.randn(K, 4, dtype=dt, device=dev)
        tj = timed(lambda: dyn.jac(x, u))
        tv = timed(lambda: dyn(x, u))
        words = nx + 4 + nx + nx * (nx + 4)
        out[f"{name}_{'f64' if dt == torch.float64 else 'f32'}"] = {
            "points": K, "jac_ms": 1e3 * tj, "value_ms": 1e3 * tv, "jac_points_per_s": K / tj,
            "algorithmic_GBps": K * words * x.element_size() / tj / 1e9}
# obstacle rows: one Newton direction at B = 4096, (20, 13, 4)
be = default_backend()
B, T, nx, nu, nobs = 4096, 20, 13, 4, 4
for dt in (torch.float32, torch.float64):
    p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev, active=True)
    z = p.z0.clone()
    xn = (torch.einsum("btij,btj->bti", p.F, z[:, :-1]) + p.c).contiguous()
    lam = torch.zeros(B, T * nx + T * (2 * nu + nobs), dtype=dt, device=dev)
    rho = torch.ones(B, dtype=dt, device=dev)
    pos = (z[:, :, None, :3] + 0.25 * torch.randn(B, T, nobs, 3, dtype=dt, device=dev)).contiguous()
    d = torch.empty_like(z)
    ws = be.new_workspace((B, T, nx, nu), z)
    for variant, w in (("team", None), ("quad", ws)):
        t = timed(lambda: be.newton_step((B, T, nx, nu), z, xn, p.F, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi, 0, 0, d,
                                         obs=(pos, 0.3), workspace=w), reps=20)
        out[f"obstacle_step_B4096_{variant}_{'f64' if dt == torch.float64 else 'f32'}_ms"] = 1e3 * t
print(json.dumps(out, indent=1))
