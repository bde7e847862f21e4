"""Sharded batch in the interior-point path: two processes (gloo for the 24-byte all-reduce), each solving one half of
a fixture's batch on the one MI355X of the box, must take the reference's batch-global exit decision together
(batch_LU.py:120-151): same iteration count as the un-sharded reference run, same solution. A rank deciding on its own
half would stop at a different iteration (the halves converge at different speeds)."""
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


def _worker(rank, world, port, name, sharded, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deq_mpc_corl_amd.qpth import qp_wrapper as ip
    g = gu.load(name)
    dt, dev = torch.float64, "cuda:0"
    B, T, nx, nu = g["B"], g["T"], g["nx"], g["nu"]
    lo, hi = rank * B // world, (rank + 1) * B // world
    tb = lambda a: torch.as_tensor(np.ascontiguousarray(a[:, lo:hi])).to(dt).to(dev)      # time-major [T, B, .]
    t0 = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev)
    mpc = ip.MPC(nx, nu, T, u_lower=t0(g["u_lo"]), u_upper=t0(g["u_hi"]), qp_iter=1, exit_unconverged=False, eps=1e-5,
                 n_batch=hi - lo, backprop=False, u_init=tb(g["u_init"]), single_qp_solve=True, exit_mode="reference",
                 process_group=dist.group.WORLD if sharded else None)
    x, u = mpc(t0(g["x0"][lo:hi]), ip.QuadCost(torch.diag_embed(tb(g["Cd"])), tb(g["c"])), ip.LinDx(tb(g["F"]), tb(g["f"])), None)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=x.cpu().numpy(), u=u.cpu().numpy(), iters=mpc.last_ipm["iters"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ip_cart_active_f64", "ip_quad13_active_f64"])
def test_sharded_ip_reference_exit_matches_unsharded_reference(name, tmp_path):
    g = gu.load(name)
    want = int(g["ipm_iters"][-1])
    res = {}
    for sharded in (True, False):
        d = tmp_path / ("sh" if sharded else "own")
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), name, sharded, str(d)), nprocs=2, join=True)
        res[sharded] = [np.load(d / f"r{r}.npz") for r in range(2)]
    B = g["B"]
    for r in range(2):
        lo, hi = r * B // 2, (r + 1) * B // 2
        assert int(res[True][r]["iters"]) == want          # both ranks stop where the un-sharded reference stopped
        assert np.abs(res[True][r]["x"] - g["x"][:, lo:hi]).max() < 1e-7 * max(1.0, np.abs(g["x"]).max())
        assert np.abs(res[True][r]["u"] - g["u"][:, lo:hi]).max() < 1e-7
    # the control experiment: without the all-reduce at least one half stops elsewhere (or the fixture would not test anything)
    assert any(int(res[False][r]["iters"]) != want for r in range(2))
