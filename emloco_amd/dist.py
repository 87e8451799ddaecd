"""Multi-GPU plumbing: one process per GPU, envs sharded across ranks, gradients all-reduced over RCCL.

The rollout shards naturally (no inter-env contacts, humanoid.py:838-841; terrain replicated): rank r owns envs
[r*E_local, (r+1)*E_local) and seeds its generators with base+rank (run.py:65).  The only exchange is the
gradient all-reduce of whatever is being trained (LocoVal: 6 174 floats; predictor: 3.2 M floats), one flat
fp32 bucket per optimiser step.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(num_envs_global, rank, world):
    """Contiguous env shard of this rank; global ids stay rank*E_local + i for bit-exact mask comparison."""
    per = num_envs_global // world
    if per * world != num_envs_global:
        raise ValueError(f"num_envs {num_envs_global} is not divisible by world size {world}")
    return rank * per, per


class FlatGradBucket:
    """One flat fp32 buffer aliased by the .grad of every parameter: a single all-reduce per step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, average=True):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.flat


def all_reduce_sum_count(loss_sum, count):
    """MSELoss(reduction='sum') semantics across ranks (common_agent.py:96): sum-reduce, then divide by the global count."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.stack([loss_sum.detach().float().reshape(()), torch.as_tensor(float(count), device=loss_sum.device)])
        dist.all_reduce(t)
        return t[0], t[1]
    return loss_sum.detach(), torch.as_tensor(float(count))
