"""N > 1 path on CPU (gloo, world_size 2): gradient averaging and batch sharding used by DataParallelTrainer / bench.py.
The data path itself has no collective (independent waveforms); the only exchange is the gradient all-reduce."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tqdne_amd.trainer import allreduce_mean_, shard_batch

    # 1. bucketed mean all-reduce == mean of the per-rank gradients, for sizes that do not divide the bucket
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(100003, generator=g)
    mine = flat.clone()
    allreduce_mean_(flat, world, bucket_elems=4096)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ok1 = torch.allclose(flat, sum(gathered) / world, atol=1e-6)

    # 2. data parallel equivalence on a toy quadratic "model": sharded batch + averaged grads == full-batch grads
    gg = torch.Generator().manual_seed(7)
    w = torch.randn(16, generator=gg)
    batch = {"signal": torch.randn(8, 16, generator=gg), "cond": torch.randn(8, 5, generator=gg)}
    local = shard_batch(batch, rank, world)
    assert local["signal"].shape[0] == 4 and local["cond"].shape[0] == 4

    def grad_of(x):
        ww = w.clone().requires_grad_(True)
        ((x @ ww) ** 2).mean().backward()
        return ww.grad

    gl = grad_of(local["signal"])
    allreduce_mean_(gl, world)
    ok2 = torch.allclose(gl, grad_of(batch["signal"]), atol=1e-6)
    ret[rank] = bool(ok1 and ok2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_gradient_allreduce_and_sharding():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
