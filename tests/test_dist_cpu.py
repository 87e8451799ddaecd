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


def _sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emloco_amd.dist import all_reduce_mean_scalar, broadcast_parameters, init_from_env, sync_running_mean_std
    from emloco_amd.utils.running_mean_std import RunningMeanStd
    init_from_env("gloo")
    torch.manual_seed(10 + rank)                            # run.py seeds with base + rank: replicas differ until the broadcast
    net = torch.nn.Linear(7, 3)
    rms = RunningMeanStd((5,))
    rms.running_mean += rank
    broadcast_parameters(net, rms)
    w = net.weight.detach().numpy().copy()
    mean_after_bcast = rms.running_mean.numpy().copy()
    # per-rank statistics of different samples, different counts
    g = torch.Generator().manual_seed(3)
    data = torch.randn(300, 5, generator=g, dtype=torch.float64) * 2 + 1
    mine = data[:100] if rank == 0 else data[100:]
    rms.running_mean.copy_(mine.mean(0)); rms.running_var.copy_(mine.var(0, unbiased=False)); rms.count.fill_(float(len(mine)))
    sync_running_mean_std(rms)
    kl = all_reduce_mean_scalar(0.25 if rank == 0 else 0.75)
    out.put((rank, w, mean_after_bcast, rms.running_mean.numpy().copy(), rms.running_var.numpy().copy(), float(rms.count), kl))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_statistics_sync_and_kl_mean():
    """The reference's other three exchange steps (common_agent.py:165-166,179-180; amp_continuous.py:287-288): parameter
    broadcast at start-up, running-mean-std synchronisation per epoch (pooled mean / variance of unequal shards), KL average."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, m0, mu0, var0, c0, kl0), (_, w1, m1, mu1, var1, c1, kl1) = res
    import numpy as np
    assert np.array_equal(w0, w1) and np.array_equal(m0, m1) and (m1 == 0).all()        # rank 0's values everywhere
    g = torch.Generator().manual_seed(3)
    data = (torch.randn(300, 5, generator=g, dtype=torch.float64) * 2 + 1).numpy()
    np.testing.assert_allclose(mu0, data.mean(0), rtol=1e-12)
    np.testing.assert_allclose(var0, data.var(0), rtol=1e-10)
    assert np.array_equal(mu0, mu1) and np.array_equal(var0, var1) and c0 == c1 == 150.0
    assert kl0 == kl1 == 0.5
