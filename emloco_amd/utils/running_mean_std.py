"""RunningMeanStd: the observation / AMP-observation normaliser of the policy and the learners.

Stands where /root/reference/pacer/pacer/utils/running_mean_std.py:9-97 stands: same constructor, the same three float64
buffers under the same names (`running_mean`, `running_var`, `count` -- reference checkpoints load), the same freeze switches
and `forward(input, unnorm=False)` contract.  Both halves of forward run in libemloco_hip.so for the batches the rollout
produces (2-D float32 on the device): the normalisation in `emloco_obs_normalize` (policy hot path, SURVEY.md section 8 row
A19), the statistics update of training mode in `emloco_rms_update` -- one launch that folds the batch's per-column mean and
unbiased variance into the running moments with the parallel-variance rule.  The per-channel image layouts of the reference
class are not part of this path and are rejected.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from ..sim import current_stream_handle


def obs_normalize(x, mean32, var32, eps, clip=5.0, split=None, out0=None, out1=None):
    """clamp((x - mean) / sqrt(var + eps), -clip, clip) on the device; optional column split into two buffers."""
    lib = L.require_device()
    rows, cols = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    if split is None:
        split = cols
    if out0 is None:
        out0 = torch.empty((rows, split), dtype=torch.float32, device=x.device)
    if out1 is None and split < cols:
        out1 = torch.empty((rows, cols - split), dtype=torch.float32, device=x.device)
    vp = C.c_void_p
    rc = lib.emloco_obs_normalize(rows, cols, vp(x.data_ptr()), x.stride(0), vp(mean32.data_ptr()), vp(var32.data_ptr()),
                                  C.c_float(eps), C.c_float(clip), split, vp(out0.data_ptr()), out0.stride(0),
                                  vp(out1.data_ptr()) if out1 is not None else None,
                                  out1.stride(0) if out1 is not None else 0, current_stream_handle(x.device))
    if rc != 0:
        raise L.EmlocoError(f"emloco_obs_normalize failed with code {rc}")
    return out0, out1


def rms_update(x, mean64, var64, count64, first_col=0, scratch=None):
    """Fold the batch x (rows, cols) into the running moments in place (emloco_rms_update); returns nothing.  `count64` is a
    0-dim float64 tensor; the kernel writes the new count to `scratch` and it is copied back on the same stream."""
    lib = L.require_device()
    rows, cols = x.shape
    if scratch is None:
        scratch = torch.empty((), dtype=torch.float64, device=x.device)
    vp = C.c_void_p
    if rows >= 512:                   # tall batch (the learner's minibatches): chunked partial moments + in-order fold, two launches
        lib.emloco_rms_update_workspace.restype = C.c_int64
        ws = torch.empty(int(lib.emloco_rms_update_workspace(rows, cols)) // 8, dtype=torch.float64, device=x.device)
        rc = lib.emloco_rms_update_chunked(rows, cols, vp(x.data_ptr()), x.stride(0), vp(mean64.data_ptr()), vp(var64.data_ptr()),
                                           vp(count64.data_ptr()), vp(scratch.data_ptr()), int(first_col), vp(ws.data_ptr()),
                                           current_stream_handle(x.device))
    else:
        rc = lib.emloco_rms_update(rows, cols, vp(x.data_ptr()), x.stride(0), vp(mean64.data_ptr()), vp(var64.data_ptr()), vp(count64.data_ptr()),
                                   vp(scratch.data_ptr()), int(first_col), current_stream_handle(x.device))
    if rc != 0:
        raise L.EmlocoError(f"emloco_rms_update failed with code {rc}")
    count64.copy_(scratch)


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        if per_channel:
            raise NotImplementedError("RunningMeanStd(per_channel=True) normalises image tensors; the rollout path has flat observations only")
        self.insize, self.epsilon, self.norm_only, self.per_channel = insize, epsilon, norm_only, False
        self.axis = [0]
        self.register_buffer("running_mean", torch.zeros(insize, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(insize, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))
        # the reference spells these two attributes this way; checkpoints and callers that poke them keep working
        self.forzen = False
        self.forzen_partial = False
        self.diff = 0

    def freeze(self):
        self.forzen = True

    def unfreeze(self):
        self.forzen = False

    def freeze_partial(self, diff):
        """Only the last `diff` columns keep learning (the future part of the observation); the count advances for all."""
        self.forzen_partial, self.diff = True, int(diff)

    def _on_device_batch(self, t):
        return t.dim() == 2 and t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and not t.requires_grad

    def forward(self, input, unnorm=False):
        mean32, var32 = self.running_mean.float(), self.running_var.float()
        if unnorm:                                   # back from normalised values: sigma * clamp(y) + mu
            out = torch.sqrt(var32 + self.epsilon) * input.clamp(-5.0, 5.0) + mean32
        elif self.norm_only:
            out = input / torch.sqrt(var32 + self.epsilon)
        elif self._on_device_batch(input):
            out, _ = obs_normalize(input, mean32, var32, self.epsilon)
        else:                                        # gradients wanted, or a layout the kernel does not take: elementwise torch on the device
            out = ((input - mean32) / torch.sqrt(var32 + self.epsilon)).clamp(-5.0, 5.0)
        # the moments learn from the batch AFTER it has been normalised with the old ones (running_mean_std.py:85-95)
        if self.training and not self.forzen:
            if not self._on_device_batch(input.detach()):
                raise L.EmlocoError("RunningMeanStd statistics update needs a 2-D float32 batch on the GPU (emloco_rms_update); there is no CPU path")
            first = input.shape[1] - self.diff if self.forzen_partial else 0
            rms_update(input.detach(), self.running_mean, self.running_var, self.count, first_col=first)
        return out
