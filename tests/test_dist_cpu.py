"""N>1 path on CPU: two gloo ranks exercise env sharding, the flat gradient bucket all-reduce and the
sum-reduction bookkeeping (no GPU, no kernels)."""
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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emloco_amd.dist import FlatGradBucket, all_reduce_sum_count, init_from_env, shard_range
    r, lr, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    start, per = shard_range(4096, r, w)
    torch.manual_seed(0)                                   # same initial weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(100, 49), torch.nn.ReLU(), torch.nn.Linear(49, 24), torch.nn.ReLU(), torch.nn.Linear(24, 1))
    bucket = FlatGradBucket(net.parameters())
    assert bucket.flat.numel() == 6174                      # the LocoVal gradient bucket (24.7 KB)
    torch.manual_seed(100 + rank)                           # different data per rank
    x, y = torch.randn(8 + rank, 100), torch.rand(8 + rank, 1)
    loss = torch.nn.functional.mse_loss(net(x), y, reduction="sum")
    loss.backward()
    local = bucket.flat.clone()
    bucket.all_reduce(average=False)
    gl, gc = all_reduce_sum_count(loss, x.shape[0])
    out.put((rank, start, per, local, bucket.flat.clone(), float(gl), float(gc), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_gradient_allreduce():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, p0, l0, g0, gl0, gc0, ls0), (r1, s1, p1, l1, g1, gl1, gc1, ls1) = res
    assert (s0, p0, s1, p1) == (0, 2048, 2048, 2048)        # contiguous shards, global ids = rank*E_local + i
    assert torch.allclose(g0, l0 + l1) and torch.equal(g0, g1)   # one flat bucket, summed, identical on both ranks
    assert gc0 == gc1 == 17.0 and abs(gl0 - (ls0 + ls1)) < 1e-4


def test_shard_range_rejects_ragged_split():
    from emloco_amd.dist import shard_range
    with pytest.raises(ValueError):
        shard_range(4097, 0, 8)
    assert shard_range(4096, 7, 8) == (3584, 512)
