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


def _eight_rank_worker(rank, world, port, out, steps):
    """One rank of the LocoVal exchange pattern: every step every rank joins ONE all-reduce of the flat bucket
    [gradient | loss sum | episode count], whatever its own episodes did, and the AdamW commit is gated by the GLOBAL count."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from emloco_amd.dist import FlatGradBucket, broadcast_parameters, init_from_env, sync_running_mean_std
    from emloco_amd.learning.flat_adamw import GatedFlatAdamW
    from emloco_amd.utils.running_mean_std import RunningMeanStd
    init_from_env("gloo")
    torch.manual_seed(50 + rank)                            # replicas differ until the broadcast (run.py:65 seeds with base + rank)
    net = torch.nn.Sequential(torch.nn.Linear(100, 49), torch.nn.ReLU(), torch.nn.Linear(49, 24), torch.nn.ReLU(), torch.nn.Linear(24, 1))
    broadcast_parameters(net)
    bucket = FlatGradBucket(net.parameters(), extra=2)
    opt = GatedFlatAdamW(net.parameters(), bucket.grads, lr=1e-3, weight_decay=1e-4)
    rms = RunningMeanStd((6,))
    g = torch.Generator().manual_seed(900 + rank)
    seen = []
    commits = 0
    for t in range(steps):
        bucket.zero()
        # rank r finishes episodes only on steps t with t % world == r (disjoint steps), rank 0 never on its own; steps with
        # t % (world + 3) >= world have no finished episode anywhere: the collective still runs, nothing is committed
        slot = t % (world + 3)
        n = (2 + rank) if (slot == rank and rank > 0) else 0
        if n:
            x, y = torch.randn(n, 100, generator=g), torch.rand(n, 1, generator=g)
            loss = torch.nn.functional.mse_loss(net(x), y, reduction="sum")
            loss.backward()
            bucket.tail[0], bucket.tail[1] = loss.detach(), float(n)
        bucket.all_reduce(average=False)                    # unconditional: one collective per step on every rank
        cnt = bucket.tail[1].clone()
        bucket.grads.div_(cnt.clamp(min=1.0))               # MSELoss(reduction='sum') / global count (common_agent.py:96)
        opt.step(gate=cnt > 0)
        commits += int(cnt.item() > 0)
        seen.append(torch.randn(4 + rank, 6, generator=g, dtype=torch.float64) * (1 + rank) + rank)   # this rank's observations
    # per-rank observation statistics (the training-mode update itself is a HIP launch, emloco_rms_update: set here directly)
    data = torch.cat(seen)
    rms.running_mean.copy_(data.mean(0)); rms.running_var.copy_(data.var(0, unbiased=False)); rms.count.fill_(float(len(data)))
    mine = (rms.running_mean.clone(), rms.running_var.clone(), float(rms.count))
    sync_running_mean_std(rms)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    # (numpy, not tensors: a tensor travels through the queue as a shared-memory handle that must outlive the sender)
    out.put((rank, flat.numpy().copy(), commits, float(opt.steps), (mine[0].numpy().copy(), mine[1].numpy().copy(), mine[2]),
             rms.running_mean.numpy().copy(), rms.running_var.numpy().copy(), float(rms.count)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gated_fit_with_episodes_on_disjoint_steps():
    """DESIGN section 6's no-deadlock claim at the driver's rank count: 8 gloo ranks, each finishing episodes on its own steps only
    (rank 0 never, some steps nobody): every rank issues the same number of collectives, the device-side gate commits AdamW on
    exactly the steps where the GLOBAL episode count is positive, the replicas stay bit-identical and moved, and the epoch's
    running-mean-std synchronisation pools eight unequal shards by the law of total variance."""
    world, steps = 8, 33
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = sum(1 for t in range(steps) if 1 <= t % (world + 3) < world)       # slots 1..7 have a finishing rank; 0, 8, 9, 10 none
    torch.manual_seed(50)
    ref = torch.nn.Sequential(torch.nn.Linear(100, 49), torch.nn.ReLU(), torch.nn.Linear(49, 24), torch.nn.ReLU(), torch.nn.Linear(24, 1))
    start = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    import numpy as np
    for r, flat, commits, nstep, _mine, _mu, _var, _c in res:
        assert np.array_equal(flat, res[0][1]), r                               # replicas identical after 33 steps
        assert commits == expect and nstep == float(expect), (r, commits, nstep)
    assert not np.array_equal(res[0][1], start.numpy())                         # ... and they moved away from rank 0's initial weights
    # pooled statistics: weights count_r / sum(count)
    cs = np.array([m[4][2] for m in res], np.float64)
    mus = np.stack([m[4][0].astype(np.float64) for m in res])
    vs = np.stack([m[4][1].astype(np.float64) for m in res])
    w = (cs / cs.sum())[:, None]
    mu = (w * mus).sum(0)
    var = (w * (vs + (mus - mu) ** 2)).sum(0)
    for m in res:
        np.testing.assert_allclose(m[5], mu, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(m[6], var, rtol=1e-6, atol=1e-7)
        assert abs(m[7] - float(cs.mean())) < 1e-3
