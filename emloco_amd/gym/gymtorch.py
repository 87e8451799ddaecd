"""gymtorch: zero-copy hand-off between simulator tensors and torch (reference: isaacgym/python/isaacgym/gymtorch.py).

`wrap_tensor` returns a NON-OWNING alias of simulator memory (gymtorch.py:60-70); `unwrap_tensor` requires a
contiguous tensor (gymtorch.py:97-107) and returns a descriptor the gym setters accept.
"""
import torch

from . import gymapi


def wrap_tensor(gym_tensor, offsets=None, counts=None):
    t = gym_tensor._t
    if offsets is not None or counts is not None:
        idx = tuple(slice(o, o + c) for o, c in zip(offsets or [0] * t.dim(), counts or t.shape))
        t = t[idx]
    return t


def unwrap_tensor(torch_tensor):
    if not torch_tensor.is_contiguous():
        raise Exception("Input tensor must be contiguous")
    if torch_tensor.dtype not in (torch.float32, torch.uint8, torch.int16, torch.int32, torch.int64):
        raise Exception("Unsupported Gym tensor dtype")
    return gymapi.Tensor(torch_tensor)
