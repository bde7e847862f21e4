"""N > 1 path on CPU: world_size-2 gloo processes, each owning a slab of the batch.
The TEST-ONLY oracle backend stands in for the GPU; what is tested is the sharding
logic: slabs, the 8-byte all-reduce that keeps the batch-global exit decision of the
reference, and the all-gather of the solution."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import golden_util as gu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, exit_mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deq_mpc_corl_amd import AffineDynamics, QuadCost
    from deq_mpc_corl_amd.sharding import gather_batch, make_sharded_mpc, shard
    from tests.oracle_backend import OracleBackend
    g = gu.load(name)
    dt = torch.float64
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    t = lambda a: shard(torch.as_tensor(np.ascontiguousarray(a)).to(dt), rank, world)
    mpc = make_sharded_mpc(nx, nu, T, torch.as_tensor(g["u_lo"]).to(dt), torch.as_tensor(g["u_hi"]).to(dt),
                           B, rank, world, dtype=dt, exit_mode=exit_mode, backend=OracleBackend())
    x0 = t(g["x0"])
    mpc.reinitialize(x0, None)
    mpc.al_iter = g["al_iter"]
    dyn = AffineDynamics(t(g["F"]), t(g["c"]))
    Bl = x0.shape[0]
    cost = QuadCost(torch.diag_embed(t(g["Qd"])), t(g["q"]), torch.zeros(Bl, T, dtype=dt))
    z0 = t(g["z0"])
    x, u, _ = mpc(x0, cost, dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    xg = gather_batch(x, B)
    ug = gather_batch(u, B)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), x=xg.numpy(), u=ug.numpy(),
                 npa=np.array(mpc.last_newton_per_al))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,B_note", [("pend_f64_al2", "even slabs"), ("pend_active_f64_al6", "even slabs")])
def test_sharded_reference_exit_matches_unsharded_reference(name, B_note, tmp_path):
    """pend_* goldens are the cases where the reference's batch-global early exit fires
    ([4,3] Newton steps): a rank deciding on its own slab would take a different number
    of steps; with the all-reduce both ranks reproduce the reference exactly."""
    g = gu.load(name)
    mp.spawn(_worker, args=(2, _free_port(), name, "reference", str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npz")
    assert list(out["npa"]) == list(g["newton_per_al"])
    assert np.abs(out["x"] - g["x"]).max() < 2e-5
    assert np.abs(out["u"] - g["u"]).max() < 2e-5


def test_sharded_fixed_mode_equals_unsharded_fixed_mode(tmp_path):
    """exit_mode='fixed' takes no batch-global decision: slabs need no collective and the
    gathered result equals the single-process result bit for bit."""
    from deq_mpc_corl_amd import MPC, AffineDynamics, QuadCost
    from tests.oracle_backend import OracleBackend
    name = "cart_f64_al2"
    g = gu.load(name)
    mp.spawn(_worker, args=(2, _free_port(), name, "fixed", str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npz")
    dt = torch.float64
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    mpc = MPC(nx, nu, T, u_lower=t(g["u_lo"]), u_upper=t(g["u_hi"]), n_batch=B, dtype=dt, exit_mode="fixed",
              backend=OracleBackend())
    mpc.reinitialize(t(g["x0"]), None)
    dyn = AffineDynamics(t(g["F"]), t(g["c"]))
    z0 = t(g["z0"])
    x, u, _ = mpc(t(g["x0"]), QuadCost(torch.diag_embed(t(g["Qd"])), t(g["q"]), torch.zeros(B, T, dtype=dt)),
                  dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    assert np.array_equal(out["x"], x.numpy()) and np.array_equal(out["u"], u.numpy())


def test_shard_range_covers_batch():
    from deq_mpc_corl_amd.sharding import shard_range
    for B in (1, 7, 8, 65536, 16385):
        for w in (1, 2, 3, 8):
            rs = [shard_range(B, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def _stream_worker(rank, world, port, name, out_dir):
    """reinitialize -> call -> warm_start_initialize -> stream calls on this rank's slab (tests/test_stream_golden.py
    does the same un-sharded)."""
    from types import SimpleNamespace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from deq_mpc_corl_amd import PendulumDynamics, QuadCost
    from deq_mpc_corl_amd.sharding import gather_batch, make_sharded_mpc, shard
    from tests.oracle_backend import OracleBackend
    g = gu.load(name)
    dt = torch.float64
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    t = lambda a, d=dt: shard(torch.as_tensor(np.ascontiguousarray(a)).to(d), rank, world)
    mpc = make_sharded_mpc(nx, nu, T, torch.as_tensor(g["u_lo"]).to(dt), torch.as_tensor(g["u_hi"]).to(dt),
                           B, rank, world, dtype=dt, exit_mode="reference", backend=OracleBackend())
    x0, Qd = t(g["x0"]), t(g["Qd"])
    Bl = x0.shape[0]
    zeros = torch.zeros(Bl, T, dtype=dt)
    dyn = PendulumDynamics()
    mpc.reinitialize(x0, None)
    mpc.al_iter = int(g["al_iter_first"])
    z0 = t(g["z0"])
    mpc(x0, QuadCost(torch.diag_embed(Qd), t(g["q0"]), zeros), dyn, dyn.jac, x_init=z0[..., :nx].clone(), u_init=z0[..., nx:].clone())
    mpc.warm_start_initialize(t(g["x_warm"], torch.float32), t(g["u_warm"], torch.float32),
                              SimpleNamespace(rho_init_max=float(g["rho_init_max"])))
    mpc.linearize_once = bool(int(g["linearize_once"]))
    n_al, xs, us, sts = [], [], [], []
    for ci in range(int(g["stream_calls"])):
        mpc.al_iter = int(g["al_iter_stream"])
        q = torch.as_tensor(np.ascontiguousarray(g["q"][ci])).to(dt)
        x, u, st = mpc(x0, QuadCost(torch.diag_embed(Qd), shard(q, rank, world), zeros), dyn, dyn.jac)
        n_al.append(len(mpc.last_newton_per_al))
        xs.append(gather_batch(x, B).numpy()); us.append(gather_batch(u, B).numpy()); sts.append(int(st))
    if rank == 0:
        np.savez(os.path.join(out_dir, "stream.npz"), x=np.stack(xs), u=np.stack(us), n_al=np.array(n_al), status=np.array(sts))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_linearize_once_breaks_in_the_same_iteration_on_every_rank(tmp_path):
    """The stream loop's break test is a BATCH mean (AL_mpc.py:406-408) and rho.max() (:412): with the batch
    sharded both are all-reduced (MPC._global_mean / _global_max), so every rank leaves the loop in the
    iteration the un-sharded reference did; a rank-local mean would leave the ranks with different
    iteration counts and the next all-reduce would hang."""
    name = "pend_stream_lin_f64"
    g = gu.load(name)
    mp.spawn(_stream_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "stream.npz")
    assert out["n_al"].tolist() == g["n_al"].tolist()
    assert out["status"].tolist() == g["status"].tolist()
    assert np.abs(out["x"] - g["x"]).max() < 2e-5 and np.abs(out["u"] - g["u"]).max() < 2e-5


def _ip_problem(B):
    """Nonlinear pendulum SQP problems whose first half starts AT the solution (x0 = 0, zero linear cost: the very
    first QP returns du = 0) while the second half has real work to do."""
    rng = np.random.default_rng(3)
    T, nx, nu = 6, 2, 1
    x0 = rng.standard_normal((B, nx)) * 0.5
    c = rng.standard_normal((T, B, nx + nu)) * 0.3
    x0[: B // 2] = 0.0
    c[:, : B // 2] = 0.0
    Cd = np.ones((T, B, nx + nu))
    return T, nx, nu, x0, Cd, c


def _ip_run(x0, Cd, c, T, nx, nu, process_group=None):
    from deq_mpc_corl_amd import PendulumDynamics
    from deq_mpc_corl_amd.qpth import qp_wrapper as ip
    from tests.oracle_backend import OracleBackend
    dt = torch.float64
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
    dyn = PendulumDynamics()
    B = x0.shape[0]
    mpc = ip.MPC(nx, nu, T, u_lower=t([-2.0]), u_upper=t([2.0]), qp_iter=4, exit_unconverged=False, eps=1e-5, n_batch=B,
                 backprop=False, verbose=0, u_init=torch.zeros(T, B, nu, dtype=dt), grad_method=ip.GradMethods.ANALYTIC,
                 solver_type="dense", single_qp_solve=False, exit_mode="fixed", backend=OracleBackend(),
                 process_group=process_group)
    x, u = mpc(t(x0), ip.QuadCost(torch.diag_embed(t(Cd)), t(c)), dyn, dyn.jac)
    return x, u, mpc.last_sqp_iters


def _ip_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    B = 8
    T, nx, nu, x0, Cd, c = _ip_problem(B)
    lo, hi = rank * B // world, (rank + 1) * B // world        # rank 0 owns the instances that are converged from the start
    x, u, n_it = _ip_run(x0[lo:hi], Cd[:, lo:hi], c[:, lo:hi], T, nx, nu, process_group=dist.group.WORLD)
    np.savez(os.path.join(out_dir, f"ip_{rank}.npz"), x=x.detach().numpy(), u=u.detach().numpy(), n_it=n_it)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ip_sqp_loop_leaves_in_the_same_iteration_on_every_rank(tmp_path):
    """qp_wrapper.MPC.solve_nonlin ends its SQP loop on the step norm of the WHOLE batch (reference
    qp_wrapper.py:355, 372). Rank 0's shard has du = 0 from the first QP on; with a rank-local test it would leave the
    loop while rank 1 is still iterating and the line search's all-reduces would pair up out of order (wrong decisions,
    then a hang). With the norm all-reduced both ranks run the iterations the un-sharded batch runs, and the
    concatenated result equals the un-sharded one."""
    mp.spawn(_ip_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    B = 8
    T, nx, nu, x0, Cd, c = _ip_problem(B)
    x, u, n_it = _ip_run(x0, Cd, c, T, nx, nu)
    r = [np.load(tmp_path / f"ip_{k}.npz") for k in range(2)]
    assert int(r[0]["n_it"]) == int(r[1]["n_it"]) == n_it and n_it > 1
    assert np.abs(np.concatenate([r[0]["x"], r[1]["x"]], axis=1) - x.detach().numpy()).max() < 1e-9
    assert np.abs(np.concatenate([r[0]["u"], r[1]["u"]], axis=1) - u.detach().numpy()).max() < 1e-9
