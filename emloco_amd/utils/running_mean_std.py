"""RunningMeanStd -- mirror of /root/reference/pacer/pacer/utils/running_mean_std.py:9-97.

Same constructor, buffers (float64 `running_mean`, `running_var`, `count`), freeze switches and `forward(input,
unnorm=False)` semantics.  The eval-mode normalisation of a 2-D batch (the policy-input hot path, SURVEY.md section 8
row A19) runs in `emloco_obs_normalize`; the statistics update (training only) and the rarely used branches stay in
torch on the device.
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


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        self.insize = insize
        self.epsilon = epsilon
        self.norm_only = norm_only
        self.per_channel = per_channel
        if per_channel:
            self.axis = {3: [0, 2, 3], 2: [0, 2], 1: [0]}[len(self.insize)]
            in_size = self.insize[0]
        else:
            self.axis = [0]
            in_size = insize
        self.register_buffer("running_mean", torch.zeros(in_size, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(in_size, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))
        self.forzen = False            # (sic) attribute names as in the reference
        self.forzen_partial = False

    def freeze(self):
        self.forzen = True

    def unfreeze(self):
        self.forzen = False

    def freeze_partial(self, diff):
        self.forzen_partial = True
        self.diff = diff

    def _update_mean_var_count_from_moments(self, mean, var, count, batch_mean, batch_var, batch_count):
        delta = batch_mean - mean
        tot_count = count + batch_count
        new_mean = mean + delta * batch_count / tot_count
        m_a = var * count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + delta ** 2 * count * batch_count / tot_count
        return new_mean, M2 / tot_count, tot_count

    def forward(self, input, unnorm=False):
        if self.per_channel:
            shape = [1, self.insize[0]] + [1] * (len(self.insize) - 1)
            current_mean = self.running_mean.view(shape).expand_as(input)
            current_var = self.running_var.view(shape).expand_as(input)
        else:
            current_mean, current_var = self.running_mean, self.running_var
        if unnorm:
            y = torch.clamp(input, min=-5.0, max=5.0)
            y = torch.sqrt(current_var.float() + self.epsilon) * y + current_mean.float()
        elif self.norm_only:
            y = input / torch.sqrt(current_var.float() + self.epsilon)
        elif (not self.per_channel) and input.dim() == 2 and input.is_cuda and input.dtype == torch.float32 \
                and input.stride(1) == 1 and not input.requires_grad:
            y, _ = obs_normalize(input, current_mean.float(), current_var.float(), self.epsilon)
        else:
            y = (input - current_mean.float()) / torch.sqrt(current_var.float() + self.epsilon)
            y = torch.clamp(y, min=-5.0, max=5.0)
        # statistics are updated AFTER normalising, as in the reference (running_mean_std.py:85-95)
        if self.training and not self.forzen:
            mean = input.mean(self.axis)
            var = input.var(self.axis)
            new_mean, new_var, new_count = self._update_mean_var_count_from_moments(
                self.running_mean, self.running_var, self.count, mean, var, input.size()[0])
            if self.forzen_partial:
                self.running_mean[-self.diff:], self.running_var[-self.diff:], self.count = \
                    new_mean[-self.diff:], new_var[-self.diff:], new_count
            else:
                self.running_mean, self.running_var, self.count = new_mean, new_var, new_count
        return y
