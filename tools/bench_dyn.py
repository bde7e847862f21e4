#!/usr/bin/env python3
"""Throughput of the pendulum1l dynamics provider kernel (GPU box): points/s and achieved HBM
bandwidth, beside the same step + Jacobian written in PyTorch (GPU, autograd) and the C oracle."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deq_mpc_corl_amd import Pendulum1lDynamics

dev = "cuda:0"
out = []
for dt, name, sz in ((torch.float64, "f64", 8), (torch.float32, "f32", 4)):
    K = 16384 * 19 * 32   # 0.9 GB of traffic in fp64: past the 256 MB Infinity Cache
    x = torch.randn(K, 2, dtype=dt, device=dev)
    u = torch.randn(K, 1, dtype=dt, device=dev)
    dyn = Pendulum1lDynamics(dt=0.05)
    for _ in range(3):
        dyn.jac(x, u)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        dyn.jac(x, u)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = K * 11 * sz   # read x (2), u (1); write xnext (2), F (6)
    r = {"kernel": "k_dyn_pendulum1l", "dtype": name, "points": K, "ms": ms, "points_per_s": K / ms * 1e3,
         "GB_per_s": nbytes / ms / 1e6, "hbm_frac_of_8TBs": nbytes / ms / 1e6 / 8000.0}
    print(json.dumps(r)); out.append(r)
from deq_mpc_corl_amd.backend import default_backend
be = default_backend()
for dt, name, sz in ((torch.float64, "f64", 8), (torch.float32, "f32", 4)):
    K = 16384 * 19 * 16
    x = torch.randn(K, 4, dtype=dt, device=dev)
    tau = torch.randn(K, 2, dtype=dt, device=dev)
    for _ in range(3):
        be.dyn_cartpole1l(x, tau, 0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 20
    for _ in range(reps):
        be.dyn_cartpole1l(x, tau, 0.05)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = K * (4 + 2 + 4 + 24) * sz
    r = {"kernel": "k_dyn_cartpole1l", "dtype": name, "points": K, "ms": ms, "points_per_s": K / ms * 1e3,
         "GB_per_s": nbytes / ms / 1e6, "hbm_frac_of_8TBs": nbytes / ms / 1e6 / 8000.0}
    print(json.dumps(r)); out.append(r)
for dt, name, sz in ((torch.float64, "f64", 8), (torch.float32, "f32", 4)):
    K = 16384 * 19 * 8
    x = torch.randn(K, 6, dtype=dt, device=dev)
    tau = torch.randn(K, 3, dtype=dt, device=dev)
    for _ in range(3):
        be.dyn_cartpole2l(x, tau, 0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        be.dyn_cartpole2l(x, tau, 0.05)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = K * (6 + 3 + 6 + 54) * sz
    r = {"kernel": "k_dyn_cartpole2l", "dtype": name, "points": K, "ms": ms, "points_per_s": K / ms * 1e3,
         "GB_per_s": nbytes / ms / 1e6, "hbm_frac_of_8TBs": nbytes / ms / 1e6 / 8000.0}
    print(json.dumps(r)); out.append(r)
# C oracle (one thread) on a bounded sample
from oracle import dyn_py
xs, us = np.random.randn(200000, 2), np.random.randn(200000, 1)
dyn_py.pendulum1l(xs[:10], us[:10], 0.05)
t0 = time.perf_counter(); dyn_py.pendulum1l(xs, us, 0.05); el = time.perf_counter() - t0
r = {"cpu_oracle_points_per_s": 200000 / el, "cores": 1}
print(json.dumps(r)); out.append(r)
json.dump(out, open("gpurun_out/dyn_providers.json", "w"), indent=1)
