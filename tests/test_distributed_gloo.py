"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding and the single final gather (the compute leg
is replaced by a deterministic stand-in, because the sampling path itself has no CPU fallback)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    import importlib
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    ti = importlib.import_module("thermodynamic-interpolation_amd")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0 = torch.arange(n_total * 6, dtype=torch.float32).reshape(n_total, 2, 3)
        cond = torch.arange(n_total, dtype=torch.float32).reshape(n_total, 1)

        def fake_rollout(x_local, cond_local, traj_offset):          # depends on the GLOBAL trajectory index, like the Philox key
            gid = torch.arange(traj_offset, traj_offset + x_local.shape[0], dtype=torch.float32)
            return x_local * 2.0 + cond_local[:, :, None] + gid[:, None, None]

        full = ti.distributed.rollout_sharded(fake_rollout, x0, cond)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_sharded_rollout_gathers_global_order(n_total):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    x0 = np.arange(n_total * 6, dtype=np.float32).reshape(n_total, 2, 3)
    want = x0 * 2.0 + np.arange(n_total, dtype=np.float32)[:, None, None] * 2.0
    for r in range(world):
        np.testing.assert_array_equal(outs[r], want)


def test_shard_bounds():
    ti = pkg()
    d = ti.distributed
    np.testing.assert_array_equal(d.shard_bounds(65536, 8), np.arange(9) * 8192)
    np.testing.assert_array_equal(d.shard_bounds(10, 4), [0, 3, 6, 8, 10])
    np.testing.assert_array_equal(d.shard_bounds(2, 4), [0, 1, 2, 2, 2])
    assert d.shard_slice(10, 3, 4) == slice(8, 10)
    with pytest.raises(ValueError):
        d.shard_bounds(4, 0)
