"""Multi-GPU sampling: independent trajectories are sharded over ranks; the ONLY collective is the final gather.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).  The
reference has no distributed path at all; trajectories never interact (edges never cross molecules), so
there is no exchange step during integration and none is invented here.  Per-trajectory RNG is keyed by the GLOBAL
trajectory index (``traj_offset``; a trajectory continued by a second call passes ``step_offset``), and ``pin_template`` fixes
the edge-row layout to the one the GLOBAL batch would get, so fixed-step samples are bit-identical for every rank count
(the adaptive ``dopri5`` chooses its steps from rank-local error norms, like the reference does per mini-batch: there the
agreement is to solver tolerance).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total: int, world: int) -> np.ndarray:
    """Contiguous near-equal blocks: bounds[r] .. bounds[r+1] is rank r's slice (first n_total % world ranks get one more)."""
    if world < 1 or n_total < 0:
        raise ValueError("world must be >= 1 and n_total >= 0")
    base, rem = divmod(n_total, world)
    counts = np.full(world, base, np.int64)
    counts[:rem] += 1
    return np.concatenate([[0], np.cumsum(counts)])


def shard_slice(n_total: int, rank: int, world: int) -> slice:
    b = shard_bounds(n_total, world)
    return slice(int(b[rank]), int(b[rank + 1]))


def gather_trajectories(local, n_total: int, group=None):
    """All-gather the per-rank blocks of a [..., n_local, ...]-leading-axis tensor back into global order.

    ``local`` is a torch tensor whose FIRST axis indexes this rank's trajectories (CPU tensor with gloo, GPU tensor with
    nccl/RCCL).  Ranks may own different counts (ragged tail): blocks are padded to the largest count for the collective and
    trimmed afterwards.  Returns the full tensor on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(n_total, world)
    counts = np.diff(bounds)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} trajectories, expected {counts[rank]}")
    cmax = int(counts.max())
    pad = local
    if local.shape[0] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    return torch.cat([blocks[r][: counts[r]] for r in range(world)], dim=0)


def pin_template(engine, n_total: int) -> str:
    """Pin ``engine`` (a PainnEngine) to the edge-row layout a single-process run over all ``n_total`` trajectories would use;
    call it on every rank before the sharded rollout.  Returns the layout's name."""
    which = engine.template_for(n_total)
    engine.set_template(which)
    return which


def rollout_sharded(rollout_fn, x0, cond, group=None):
    """Run ``rollout_fn(x0_local, cond_local, traj_offset) -> end_state_local [n_local, ...]`` on this rank's shard of the
    global batch (x0 / cond indexed by global trajectory on the first axis) and gather the end states of all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = int(x0.shape[0])
    sl = shard_slice(n_total, rank, world)
    local = rollout_fn(x0[sl], None if cond is None else cond[sl], sl.start)
    return gather_trajectories(local, n_total, group)
