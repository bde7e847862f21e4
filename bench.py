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


VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}   # MI355X vector (non-matrix) peaks, MI355X_MICROARCH.md


def flops_per_solve(T, nx, nu, al_iter):
    """Estimate: per Newton step and stage one n x n LDL' panel (n^3/3), the nx
    rows of -rho F_t through it (nx n^2), F'F (nx n^2 / 2 sym.), the Schur complement (nx^2 n / 2 sym.),
    two triangular solves (2 n^2) and the 20-candidate merits (~40 n); fma = 2 flops."""
    n = nx + nu
    per_stage = n ** 3 / 3 + nx * n * n + nx * n * n / 2 + nx * nx * n / 2 + 2 * n * n + 40 * n
    return 2.0 * per_stage * T * 4 * al_iter


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run)
    BEFORE this process touches the GPU, and pass their JSON line through. Never re-execs a process that
    initialised HIP. Fewer devices than ranks: refuse, unless BENCH_DIST_BACKEND=gloo asks for the CPU-
    rendezvous rehearsal in which ranks share devices."""
    import socket
    import subprocess

    import torch
    ndev = torch.cuda.device_count()     # does not initialise the GPU on this image
    env = dict(os.environ)
    if ndev < args.gpus and env.get("BENCH_DIST_BACKEND", "nccl") != "gloo":
        sys.exit(f"bench.py: --gpus {args.gpus} but {ndev} device(s) visible. Launch on a node with {args.gpus} GPUs, "
                 "or set BENCH_DIST_BACKEND=gloo to rehearse with ranks sharing devices.")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def timed(torch, fn, steps, warmup, barrier):
    """W untimed + K timed calls of fn(i), bracketed by barrier + synchronize; HIP events per call on the
    launching stream (torch's current stream is the one every entry point is given)."""
    for i in range(warmup):
        fn(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        fn(warmup + i)
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, sum(a.elapsed_time(b) for a, b in ev) / max(steps, 1)


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
    ap.add_argument("--no-extras", action="store_true", help="skip the f64 / newton_step / ipm sub-records")
    ap.add_argument("--variant", default="auto", choices=["auto", "team", "quad"])
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
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
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from deq_mpc_corl_amd import synthetic_problem
    from deq_mpc_corl_amd import _lib
    from deq_mpc_corl_amd.backend import default_backend

    be = default_backend()
    B, T, nx, nu = args.batch, args.T, args.nx, args.nu
    n = nx + nu
    M = T * nx + 2 * T * nu
    dims = (B, T, nx, nu)
    flags = _lib.ALQP_INIT_MERIT | _lib.ALQP_DUAL_UPDATE

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_solve(dtype_name, steps, warmup):
        """K fused solves of the rank's shard in `dtype_name`; returns (elapsed, kernel_ms, ok)."""
        dt = torch.float32 if dtype_name == "f32" else torch.float64
        # each rank owns a different shard of the global batch (seed offset = rank * chunks)
        p = synthetic_problem(B, T, nx, nu, seed=1000 * rank, dtype=dt, device=dev)
        total = steps + warmup
        zs = [p.z0.clone() for _ in range(total)]
        lams = [torch.zeros(B, M, dtype=dt, device=dev) for _ in range(total)]
        rhos = [torch.ones(B, dtype=dt, device=dev) for _ in range(total)]
        phi = torch.zeros(B, dtype=dt, device=dev)
        rn2 = torch.zeros(B, dtype=dt, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.uint8, device=dev)

        def step(i):
            be.solve_lin(dims, p.Qd, p.q, p.F, p.c, p.x0, p.u_lo, p.u_hi, 0, 0, zs[i], lams[i], rhos[i],
                         phi, rn2, info, status, al_iter=args.al_iter, max_newton=4, n_ls=20, flags=flags,
                         rho_scale=10.0, variant=args.variant)

        elapsed, kernel_ms = timed(torch, step, steps, warmup, barrier)
        ok = bool((info == 0).all().item()) and bool((status == 1).all().item())
        return elapsed, kernel_ms, ok, p

    elapsed, kernel_ms, ok, p = run_solve(args.dtype, args.steps, args.warmup)
    variant = be.last_variant  # what 'auto' resolved to (quad from B = 4096 per GPU, team below)

    if world > 1:
        tmax = torch.tensor([elapsed, 0.0 if ok else 1.0], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        ok = float(tmax[1].item()) == 0.0

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        extras = bench_extras(args, torch, be, _lib, dev, dims, p, run_solve, barrier)

    if rank == 0:
        value = world * B * args.steps / elapsed
        sz = 4 if args.dtype == "f32" else 8
        alg_bytes = words_per_solve(T, nx, nu) * sz * B
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                key = f"{variant}_{args.dtype}_B{B}_T{T}_nx{nx}_nu{nu}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
                traffic_src = f"profiles/hbm_traffic.json[{key}] <- {rec.get(key, {}).get('source')}"
            except Exception:
                traffic = None
        flops = flops_per_solve(T, nx, nu, args.al_iter) * B
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "valu_frac": flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[args.dtype],
                         # achieved solves/s over what the vector peak allows for this arithmetic (65 M solves/s in fp32 at
                         # (20,13,4)): the ceiling of the fused unit, which the algorithmic HBM roof (300 M solves/s) is not
                         "valu_capped_frac": flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[args.dtype],
                         "valu_cap_solves_per_s": B * VALU_PEAK_TFLOPS[args.dtype] * 1e12 / flops,
                         "valu_peak_tflops": VALU_PEAK_TFLOPS[args.dtype], "flops_per_launch_est": flops,
                         "kernel": KERNEL_OF[variant], "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "achieved = algorithmic bytes / kernel time (BASELINE metric); traffic = measured HBM "
                                 "bytes per launch (rocprofv3 FETCH_SIZE*2+WRITE_SIZE, committed profile); "
                                 "hbm_traffic_frac = that volume / kernel time / 8 TB/s; valu_frac = estimated flops / "
                                 "kernel time / vector peak. The fused solve is bound by VALU issue + its own workspace "
                                 "traffic, not by the algorithmic bytes (DESIGN.md section 5)"},
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, T, nx, nu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_extras(args, torch, be, _lib, dev, dims, p, run_solve, barrier):
    """Sub-records of the one JSON line (rank 0, N = 1): the other dtype of the same workload, the
    per-Newton-step unit of the nonlinear-caller mode (24 960 B per step at (20,13,4) fp32 - SURVEY 8d)
    and the interior-point QP of the qp_wrapper path. Short runs; each is timed like the headline."""
    B, T, nx, nu = dims
    n = nx + nu
    out = {}
    other = "f64" if args.dtype == "f32" else "f32"
    try:
        el, kms, ok2, _ = run_solve(other, max(3, args.steps // 4), 2)
        steps2 = max(3, args.steps // 4)
        sz = 8 if other == "f64" else 4
        out[other] = {"value": B * steps2 / el, "unit": "QP solves/s", "ms_per_step": 1e3 * el / steps2,
                      "kernel_ms": kms, "all_instances_ok": ok2,
                      "frac": words_per_solve(T, nx, nu) * sz * B / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "valu_frac": flops_per_solve(T, nx, nu, args.al_iter) * B / (kms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[other]}
    except Exception as e:  # pragma: no cover
        out[other] = {"error": repr(e)}
    # ---- one Newton direction (alqp_newton_step): z, Qd, q, F, xnext, lam, rho in; d out
    try:
        dt = p.z0.dtype
        M = T * nx + 2 * T * nu
        xn = (torch.einsum("btij,btj->bti", p.F, p.z0[:, :-1]) + p.c).contiguous()
        lam = torch.zeros(B, M, dtype=dt, device=dev)
        rho = torch.ones(B, dtype=dt, device=dev)
        d = torch.empty(B, T, n, dtype=dt, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        k = max(5, args.steps // 2)
        # the drop-in class takes the quad kernels (16 instances per wavefront) from B = 4096, the team kernels below
        wsq = be._workspace(dims, p.z0)[0] if B >= be.QUAD_MIN_BATCH else None
        el, kms = timed(torch, lambda i: be.newton_step(dims, p.z0, xn, p.F, p.x0, lam, rho, p.Qd, p.q, p.u_lo, p.u_hi,
                                                         0, 0, d, info=info, workspace=wsq), k, 2, barrier)
        words = 3 * T * n + (T - 1) * nx * n + T * nx + M + 1 + T * n
        sz = p.z0.element_size()
        out["newton_step"] = {"value": B * k / el, "unit": "Newton steps/s", "kernel_ms": kms,
                              "algorithmic_bytes_per_step": words * sz,
                              "achieved_gbs": words * sz * B / (kms * 1e-3) / 1e9,
                              "frac": words * sz * B / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "kernel": getattr(be, "last_step_kernel", "k_newton_step"), "dtype": args.dtype}
    except Exception as e:  # pragma: no cover
        out["newton_step"] = {"error": repr(e)}
    # ---- interior-point QP (qp_wrapper path), exit mode "fixed": 20 iterations in one launch
    try:
        dt = torch.float64   # the ip path's callers run fp64 (policies.py:1141)
        p64 = p if p.z0.dtype == dt else None
        from deq_mpc_corl_amd import synthetic_problem
        Bi = min(B, 8192)
        q64 = synthetic_problem(Bi, T, nx, nu, seed=0, dtype=dt, device=dev)
        tm = lambda a: a.transpose(0, 1).contiguous()
        Cd, c, F, f = tm(q64.Qd), tm(q64.q), tm(q64.F), tm(q64.c)
        k = 3
        res = {}

        def ipm(i):
            res["o"] = be.ipm_solve((Bi, T, nx, nu), Cd, c, F, f, q64.x0, q64.u_hi, q64.u_lo, exit_mode="fixed")

        el, kms = timed(torch, ipm, k, 1, barrier)
        nk = n * T + 4 * T * nu + T * nx
        out["ipm"] = {"value": Bi * k / el, "unit": "QPs/s (20 interior-point iterations each)", "kernel_ms": kms,
                      "batch": Bi, "dtype": "f64", "kkt_order": nk,
                      "max_best_residual": float(res["o"]["resid"].max().item()),
                      "kernel": "k_ipm_g4 (register/LDS-resident; AlqpIpmParams.variant auto)",
                      "algorithmic_bytes_per_qp": 8 * (2 * T * n + (T - 1) * nx * (n + 1) + nx + 2 * T * n + 4 * T * nu + T * nx),
                      "measured_hbm_bytes_per_qp": 140e3, "traffic_source": "profiles/r03/ipm_f64_summary.json",
                      "note": "replaces 40 dense LU factorisations of order %d per QP" % nk}
    except Exception as e:  # pragma: no cover
        out["ipm"] = {"error": repr(e)}
    return out


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
