"""Multi-GPU plumbing: one process per GPU, envs sharded across ranks, gradients all-reduced over RCCL.

The rollout shards naturally (no inter-env contacts, humanoid.py:838-841; terrain replicated): rank r owns envs
[r*E_local, (r+1)*E_local) and seeds its generators with base+rank (run.py:65).  The exchange steps are the ones the
reference issues through Horovod:
  * parameters broadcast from rank 0 at start-up                 common_agent.py:165-166   (hvd.setup_algo)
  * the gradient of whatever is trained, one flat fp32 bucket     a2c_common / amp_continuous.py:515-598
    per optimiser step (LocoVal 6 174 floats, predictor 3.2 M, policy + critic + discriminator ~11 M)
  * the running-mean-std statistics once per epoch                common_agent.py:179-180   (hvd.sync_stats)
  * the epoch's KL scalar                                         amp_continuous.py:287-288 (hvd.average_value)
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests and by the two-ranks-on-one-GPU tests (device tensors are
staged through the host there: gloo's device support depends on the build).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("EMLOCO_FORCE_COLLECTIVES") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def is_distributed():
    """More than one rank -- or EMLOCO_FORCE_COLLECTIVES=1 with an initialised group of ONE: every collective of the product then runs
    (an all-reduce over one rank is the identity) so that the RCCL path -- stream-ordered launches on the stream the caller is on --
    can be exercised on a one-GPU box (tests/test_gpu_dist.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("EMLOCO_FORCE_COLLECTIVES") == "1"


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def get_rank():
    return dist.get_rank() if is_distributed() else 0


rank = get_rank


def barrier():
    """All ranks arrive before any leaves; no-op on one rank."""
    if is_distributed():
        dist.barrier()


def shard_range(num_envs_global, rank, world):
    """Contiguous env shard of this rank; global ids stay rank*E_local + i for bit-exact mask comparison."""
    per = num_envs_global // world
    if per * world != num_envs_global:
        raise ValueError(f"num_envs {num_envs_global} is not divisible by world size {world}")
    return rank * per, per


def _staged(t):
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_(t, op=None):
    """In-place sum (default) all-reduce of a tensor; no-op on one rank."""
    if not is_distributed():
        return t
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def broadcast_(t, src=0):
    if not is_distributed():
        return t
    if _staged(t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def broadcast_parameters(*modules_or_tensors, src=0):
    """Every rank takes rank `src`'s values for all parameters and buffers of the given modules (or plain tensors), packed
    into one flat buffer per dtype: what hvd.setup_algo does before training (common_agent.py:165-166).  Replicas that are
    seeded with base + rank (run.py:65) start from different initialisations without it."""
    tensors = []
    for m in modules_or_tensors:
        if isinstance(m, torch.nn.Module):
            tensors += [p.data for p in m.parameters()] + [b.data for b in m.buffers()]
        elif m is not None:
            tensors.append(m.data if isinstance(m, torch.nn.Parameter) else m)
    if not is_distributed() or not tensors:
        return
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for group in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in group])
            broadcast_(flat, src)
            o = 0
            for t in group:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()


class FlatGradBucket:
    """One flat fp32 buffer aliased by the .grad of every parameter: a single all-reduce per step.  `extra` appends that many
    scalars behind the gradients (`tail`): loss sums / sample counts that travel in the same collective."""

    def __init__(self, params, extra=0, align=1):
        """align: every parameter's slice starts at a multiple of `align` elements (zeros in between) -- 64 (256 bytes) keeps the
        16-byte-load kernels usable on views of a flat PARAMETER buffer laid out the same way (predictor/fused_adam.py)."""
        self.params = [p for p in params if p.requires_grad]
        self.align = int(align)
        self.offsets = []
        o = 0
        for p in self.params:
            o = -(-o // self.align) * self.align
            self.offsets.append(o)
            o += p.numel()
        n = -(-o // self.align) * self.align
        dev = self.params[0].device
        self.flat = torch.zeros(n + extra, dtype=torch.float32, device=dev)
        self.grads = self.flat[:n]
        self.tail = self.flat[n:]
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def release(self):
        """Ahead of a backward pass: detach every parameter's .grad from the bucket.  With .grad aliasing its slice autograd's accumulation
        node ADDS each incoming gradient into it -- one elementwise launch per parameter and step; with .grad = None it keeps the
        incoming tensor and `gather` copies them all into their slices in one launch per 96 tensors (`emloco_gather_flat`)."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """Behind the backward pass that followed `release`: the gradients autograd left on the parameters -> their slices of the flat
        buffer (a parameter that received none keeps the zeros of `zero()`), and every .grad aliases its slice again."""
        got = [(p.grad, o) for p, o in zip(self.params, self.offsets) if p.grad is not None]
        if got and self.flat.is_cuda:
            import ctypes as C
            from .predictor import ops
            src = [g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous() for g, _ in got]
            n = len(src)
            ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in src])
            numel = (C.c_int64 * n)(*[t.numel() for t in src])
            offs = (C.c_int64 * n)(*[o for _, o in got])
            ops._chk(ops._lib().emloco_gather_flat(n, ptrs, numel, offs, C.c_void_p(self.flat.data_ptr()), ops._st(self.flat)), "emloco_gather_flat")
        else:                                                 # host tensors (the multi-process CPU tests)
            for g, o in got:
                self.flat[o:o + g.numel()].copy_(g.reshape(-1))
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def all_reduce(self, average=True):
        if is_distributed():
            all_reduce_(self.flat)
            if average:
                self.grads.div_(dist.get_world_size())
        return self.flat


def all_reduce_sum_count(loss_sum, count):
    """MSELoss(reduction='sum') semantics across ranks (common_agent.py:96): sum-reduce, then divide by the global count."""
    if is_distributed():
        t = torch.stack([loss_sum.detach().float().reshape(()), torch.as_tensor(float(count), device=loss_sum.device)])
        all_reduce_(t)
        return t[0], t[1]
    return loss_sum.detach(), torch.as_tensor(float(count))


def all_reduce_mean_scalar(x):
    """hvd.average_value (amp_continuous.py:287-288): the mean of a python / tensor scalar over the ranks."""
    if not is_distributed():
        return x
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def sync_running_mean_std(*rms_modules):
    """hvd.sync_stats (common_agent.py:179-180): once per epoch every rank's observation statistics become the statistics
    of the pooled samples.  Each RunningMeanStd holds (count, mean, var) of what its rank saw; with weights w_r = count_r /
    sum(count): mean = sum w_r mean_r, var = sum w_r (var_r + (mean_r - mean)^2) (the law of total variance -- rl_games
    averages the three buffers, which drops the between-rank term), count = mean of the counts (so it keeps tracking the
    number of updates, not ranks x updates).  One flat fp64 all-reduce for all modules."""
    mods = [m for m in rms_modules if m is not None]
    if not is_distributed() or not mods:
        return
    with torch.no_grad():
        parts = []
        for m in mods:
            c = m.count.double().reshape(1)
            mu = m.running_mean.double().reshape(-1)
            ex2 = m.running_var.double().reshape(-1) + mu * mu
            parts += [c, c * mu, c * ex2]
        flat = torch.cat(parts)
        all_reduce_(flat)
        o = 0
        w = dist.get_world_size()
        for m in mods:
            n = m.running_mean.numel()
            c = flat[o]; s1 = flat[o + 1:o + 1 + n]; s2 = flat[o + 1 + n:o + 1 + 2 * n]
            o += 1 + 2 * n
            mu = s1 / c
            var = (s2 / c - mu * mu).clamp_(min=0.0)
            m.running_mean.copy_(mu.view_as(m.running_mean).to(m.running_mean.dtype))
            m.running_var.copy_(var.view_as(m.running_var).to(m.running_var.dtype))
            m.count.copy_((c / w).to(m.count.dtype).view_as(m.count))
