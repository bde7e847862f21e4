#!/usr/bin/env python3
"""Which device kernels one drop-in MPC.__call__ launches at the reference's batch size, and their device time
(torch profiler): python tools/profile_call.py [B] [f32|f64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost, synthetic_problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
T, nx, nu = 20, 13, 4
dev = "cuda:0"
p = synthetic_problem(B, T, nx, nu, seed=0, dtype=dt, device=dev)
dyn = AffineDynamics(p.F, p.c)
cost = QuadCost(torch.diag_embed(p.Qd), p.q, torch.zeros(B, T, device=dev, dtype=dt))
mpc = MPC(nx, nu, T, u_lower=p.u_lo, u_upper=p.u_hi, n_batch=B, dtype=dt, exit_mode="fixed")
xi, ui = p.z0[..., :nx].clone(), p.z0[..., nx:].clone()
def call():
    mpc.reinitialize(p.x0, None)
    return mpc(p.x0, cost, dyn, dyn.jac, x_init=xi, u_init=ui)
for _ in range(5):
    call()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        call()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dtm = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if dtm and e.device_type.name != "CPU":
        rows.append((dtm / 10.0, e.count / 10.0, e.key[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"device kernels per call: {sum(r[1] for r in rows):.1f}, device time per call {tot:.1f} us")
for r in rows[:25]:
    print(f"  {r[0]:8.1f} us  x{r[1]:4.1f}  {r[2]}")
