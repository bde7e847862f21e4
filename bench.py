#!/usr/bin/env python3
"""Benchmark of the hot path: batched AL MPC/QP solves per second.

Metric (BASELINE.json): QP solves/sec at (B, T=20, nx=13, nu=4); one "step" is one
fused solve (al_iter=2 x 4 Newton steps x 20-point line search + 2 dual updates,
starting from rho=1, lam=0, z0=x_ref - SURVEY.md 8d) of one batch of B=16384
synthetic instances per GPU, inputs resident in HBM. Weak scaling: every rank owns
its own 16384-instance shard, there is no collective in the data path
(exit_mode='fixed' takes no batch-global decision).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy
KERNEL_OF = {"team": "k_solve_lin", "quad": "k_solve_lin_quad"}


def words_per_solve(T, nx, nu):
    """Algorithmic words moved per fused solve: read Qd,q,z0 / write z,
    read F,c,x0, read+write lam, rho in/out + status."""
    n = nx + nu
    return 4 * T * n + (T - 1) * nx * (n + 1) + nx + 2 * (T * nx + 2 * T * nu) + 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--batch", type=int, default=16384, help="instances per GPU")
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--nx", type=int, default=13)
    ap.add_argument("--nu", type=int, default=4)
    ap.add_argument("--al-iter", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", default="auto", choices=["auto", "team", "quad"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_DIST_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks (ranks
    # then share devices); the driver's multi-GPU runs use the default: RCCL ("nccl") over xGMI
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd import _lib
    from deq_mpc_corl_amd.backend import default_backend

    be = default_backend()
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    B, T, nx, nu = args.batch, args.T, args.nx, args.nu
    n = nx + nu
    M = T * nx + 2 * T * nu
    dims = (B, T, nx, nu)
    # each rank owns a different shard of the global batch (seed offset = rank * chunks)
    p = synthetic_problem(B, T, nx, nu, seed=1000 * rank, dtype=dt, device=dev)

    total = args.steps + args.warmup
    zs = [p.z0.clone() for _ in range(total)]
    lams = [torch.zeros(B, M, dtype=dt, device=dev) for _ in range(total)]
    rhos = [torch.ones(B, dtype=dt, device=dev) for _ in range(total)]
    phi = torch.zeros(B, dtype=dt, device=dev)
    rn2 = torch.zeros(B, dtype=dt, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.uint8, device=dev)
    flags = _lib.ALQP_INIT_MERIT | _lib.ALQP_DUAL_UPDATE

    def step(i):
        be.solve_lin(dims, p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, zs[i], lams[i], rhos[i],
                     phi, rn2, info, status, al_iter=args.al_iter, max_newton=4, n_ls=20, flags=flags,
                     rho_scale=10.0, variant=args.variant)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step(args.warmup + i)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    ok = bool((info == 0).all().item()) and bool((status == 1).all().item())

    if world > 1:
        tmax = torch.tensor([elapsed, 0.0 if ok else 1.0], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        ok = float(tmax[1].item()) == 0.0

    variant = be.last_variant  # what 'auto' resolved to (quad from B = 4096 per GPU, team below)
    if rank == 0:
        value = world * B * args.steps / elapsed
        sz = 4 if args.dtype == "f32" else 8
        alg_bytes = words_per_solve(T, nx, nu) * sz * B
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                key = f"{variant}_{args.dtype}_B{B}_T{T}_nx{nx}_nu{nu}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "QP solves/sec at (B,T=20,nx=13,nu=4)",
            "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{ {(2, 1): 'pendulum', (8, 2): 'flying_cartpole2d', (13, 4): 'rex_quadrotor'}.get((nx, nu), 'synthetic') }"
                                   f"-shaped (nx={nx},nu={nu}) T={T} B={B}/GPU fused AL solve, "
                                   f"al_iter={args.al_iter}x4 Newton, 20-pt line search, exit_mode=fixed",
                       "global_batch": world * B, "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "kernel_variant": variant},
            "all_instances_ok": ok,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": KERNEL_OF[variant], "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "achieved = algorithmic bytes / kernel time (BASELINE metric). The quad kernel "
                                 "deliberately streams its per-stage factor through an HBM workspace: 'traffic' "
                                 "is the measured HBM volume per launch (rocprofv3 FETCH_SIZE*2+WRITE_SIZE), "
                                 "see DESIGN.md section 5"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, T, nx, nu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, T, nx, nu):
    """The CPU oracle (banded C restatement, OpenMP over instances) on a bounded sample
    of the same workload, all host threads. A reported baseline, not the target."""
    import torch

    from deq_mpc_corl_amd import synthetic_problem
    from oracle import oracle_py as orc

    dt = torch.float32 if args.dtype == "f32" else torch.float64
    # use the cores this process may actually run on (the GPU box gives a 1-GPU job a CPU
    # share; OpenMP's default of one thread per host core oversubscribes it badly)
    avail = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = min(avail, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    threads = max(1, min(orc.max_threads(), avail, 16))
    orc.set_threads(threads)
    Bs = 2048
    p = synthetic_problem(Bs, T, nx, nu, seed=0, dtype=dt)
    c = lambda a: a.numpy()
    arrs = [c(a) for a in (p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, p.z0)]
    orc.solve_lin(args.dtype, *arrs, al_iter=args.al_iter, exit_mode="fixed")  # warm
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 12.0:
        orc.solve_lin(args.dtype, *arrs, al_iter=args.al_iter, exit_mode="fixed")
        reps += 1
    el = time.perf_counter() - t0
    return {"value": reps * Bs / el, "unit": "QP solves/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x {Bs} instances of the same workload ({el:.1f} s), oracle/alqp_oracle.c "
                      f"banded path, OpenMP {threads} threads"}


if __name__ == "__main__":
    main()
