"""Batch sharding across the GPUs of one node.

Every QP instance is independent, so the batch axis shards with no arithmetic across
ranks: rank r owns the contiguous slab [lo, hi) of every [B, ...] tensor and keeps its
solver state (lamda_prev, rho_prev, x_init, u_init) resident on its own GPU.

Collectives (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in
the CPU tests) appear in exactly two places:
  * exit_mode="reference": an 8-byte all-reduce of sum_b |r+|^2 per Newton step, so every
    rank takes the batch-global exit decision of the un-sharded reference
    (qpth/al_utils.py:486,552,560-564). exit_mode="fixed" needs none;
  * gather_batch(): one all-gather of the solution slabs, only when the caller's outer
    step needs the full batch on every rank.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous slab of rank `rank`; the first B % world ranks get one extra instance."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(t, rank, world):
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi].contiguous()


def gather_batch(t_local, B_total=None, group=None):
    """All-gather the batch axis. Equal slabs use one all_gather_into_tensor (a direct
    exchange over xGMI: each rank pushes its slab to the 7 peers); ragged slabs fall
    back to the list form."""
    world = dist.get_world_size(group)
    if world == 1:
        return t_local
    t_local = t_local.contiguous()
    if B_total is None or B_total % world == 0:
        out = torch.empty((world * t_local.shape[0],) + tuple(t_local.shape[1:]),
                          dtype=t_local.dtype, device=t_local.device)
        dist.all_gather_into_tensor(out, t_local, group=group)
        return out
    sizes = [shard_range(B_total, r, world) for r in range(world)]
    parts = [torch.empty((hi - lo,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
             for lo, hi in sizes]
    dist.all_gather(parts, t_local, group=group)
    return torch.cat(parts, 0)


def make_sharded_mpc(n_state, n_ctrl, T, u_lower, u_upper, n_batch_total, rank, world, group=None,
                     **kw):
    """An MPC for this rank's slab of a global batch of `n_batch_total` instances."""
    from .qpth.AL_mpc import MPC
    lo, hi = shard_range(n_batch_total, rank, world)
    mpc = MPC(n_state, n_ctrl, T, u_lower=u_lower, u_upper=u_upper, n_batch=hi - lo, **kw)
    if world > 1 and kw.get("exit_mode", "reference") == "reference":
        mpc.process_group = group if group is not None else dist.group.WORLD
    return mpc
