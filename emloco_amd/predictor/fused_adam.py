"""clip_grad_norm_ + Adam for the predictor as three launches on flat buffers (emloco_adam_clip_flat).

The reference's loops call `torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)` and `optimizer.step()` of a
`torch.optim.Adam(model.parameters(), lr)` (train_jta.py:317-318,411; train_jrdb.py likewise): with torch's foreach implementations that
is ~25 launches over the model's 118 tensors.  `FlatClipAdam` IS a torch.optim.Adam -- same constructor, same `param_groups`, the same
`state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter: checkpoints travel both ways) -- whose parameters, gradients and
moments are views into four flat fp32 buffers:

* parameters are moved into one buffer at construction (`p.data` becomes a view of it; in-place loads -- `load_state_dict`,
  `broadcast_parameters` -- keep the aliasing, `model.to(...)` afterwards does not and is refused at the next step);
* gradients alias a `FlatGradBucket` (the one the data-parallel all-reduce uses when there is one);
* `step(max_grad_norm=...)` clips and updates in one call; the total norm clip_grad_norm_ would have returned stays on the device
  (`last_norm`).

One param group, no amsgrad / maximize (what the reference uses); anything else raises.

Two deliberate differences from torch.optim.Adam, both outside what the reference's loops do: (1) a parameter whose gradient was never
written is treated as having a ZERO gradient (its slice of the bucket is zero: step count and moment decay advance), where torch skips
parameters whose `.grad` is None -- every parameter of the predictor receives a gradient every step; (2) a NaN / Inf gradient norm
makes the clip coefficient NaN and with it every gradient, parameter and moment (as `clip_grad_norm_` does: loud, not partial).
"""
import ctypes as C
import math

import torch

from ..dist import FlatGradBucket
from . import ops


class FlatClipAdam(torch.optim.Adam):
    ALIGN = 64

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, bucket=None, counted=False):
        """counted: the step count lives on the device (`emloco_adam_clip_flat_counted`) so that a step captured in a HIP graph replays
        as the NEXT step (the PPO learner's graphed optimiser step); the per-parameter `step` entries of the state are then views of
        that one device scalar, as torch's capturable Adam keeps them."""
        params = [p for p in params if p.requires_grad]
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, capturable=False)
        self._counted = bool(counted)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatClipAdam: one parameter group")
        dev = params[0].device
        if dev.type != "cuda" or any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise RuntimeError("FlatClipAdam: fp32 parameters on one GPU (the update is a HIP kernel; there is no CPU path)")
        self._params = params
        # every slice starts on a 256-byte boundary (ALIGN elements; zeros in between: their gradient, moments and update stay zero):
        # the GEMM kernels load weights 16 bytes at a time
        self.bucket = bucket if bucket is not None else FlatGradBucket(params, align=self.ALIGN)
        if [id(p) for p in self.bucket.params] != [id(p) for p in params] or self.bucket.align % 4:
            raise ValueError("FlatClipAdam: the gradient bucket must be built over the same parameters in the same order, 16-byte aligned slices")
        self._offsets = list(self.bucket.offsets)
        n = self.bucket.grads.numel()
        self._n = n
        self._flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self._flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self._flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._count = torch.zeros(1, dtype=torch.float32, device=dev) if self._counted else None
        with torch.no_grad():
            for p, o in zip(params, self._offsets):
                k = p.numel()
                self._flat_p[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self._flat_p[o:o + k].view_as(p)
                self.state[p] = {"step": self._count[0] if self._counted else torch.tensor(0.0),
                                 "exp_avg": self._flat_m[o:o + k].view_as(p), "exp_avg_sq": self._flat_v[o:o + k].view_as(p)}
        self._ws = torch.zeros(ops._lib().emloco_adam_clip_flat_workspace(n), dtype=torch.float32, device=dev)
        self._t = 0

    @property
    def last_norm(self):
        """Total gradient norm of the last clipped step (a device scalar: reading it synchronises)."""
        return self._ws[0]

    def zero_grad(self, set_to_none=False):
        self.bucket.zero()                       # (set_to_none would drop the aliasing)

    def _check_aliasing(self):
        base_p, base_g = self._flat_p.data_ptr(), self.bucket.grads.data_ptr()
        for p, o in zip(self._params, self._offsets):
            if p.data_ptr() != base_p + 4 * o or p.grad is None or p.grad.data_ptr() != base_g + 4 * o:
                raise RuntimeError("FlatClipAdam: a parameter or its .grad no longer aliases the flat buffers (model.to(), zero_grad("
                                   "set_to_none=True) or a re-assigned .grad after the optimiser was built)")

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=0.0):
        if closure is not None:
            raise NotImplementedError("FlatClipAdam: no closure")
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise NotImplementedError("FlatClipAdam: amsgrad / maximize")
        self._check_aliasing()
        if self._counted:
            P = lambda t: C.c_void_p(t.data_ptr())
            b1, b2 = g["betas"]
            ops._chk(ops._lib().emloco_adam_clip_flat_counted(self._n, P(self._flat_p), P(self.bucket.grads), P(self._flat_m), P(self._flat_v),
                                                              float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                                              float(max_grad_norm or 0.0), P(self._ws), P(self._count), ops._st(self._flat_p)),
                     "emloco_adam_clip_flat_counted")
            return
        self._t += 1
        b1, b2 = g["betas"]
        bc1, bc2s = 1.0 - b1 ** self._t, math.sqrt(1.0 - b2 ** self._t)
        P = lambda t: C.c_void_p(t.data_ptr())
        ops._chk(ops._lib().emloco_adam_clip_flat(self._n, P(self._flat_p), P(self.bucket.grads), P(self._flat_m), P(self._flat_v),
                                                  float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                                  float(bc1), float(bc2s), float(max_grad_norm or 0.0), P(self._ws), ops._st(self._flat_p)),
                 "emloco_adam_clip_flat")
        for p in self._params:
            self.state[p]["step"] += 1           # (host tensors, as torch's non-capturable Adam keeps them)

    def load_state_dict(self, state_dict):
        """A torch.optim.Adam / FlatClipAdam state dict: the loaded moments are copied INTO the flat buffers."""
        super().load_state_dict(state_dict)
        steps = set()
        with torch.no_grad():
            for p, o in zip(self._params, self._offsets):
                k, st = p.numel(), self.state.get(p, {})
                m, v = self._flat_m[o:o + k].view_as(p), self._flat_v[o:o + k].view_as(p)
                if "exp_avg" in st:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    steps.add(int(round(float(st["step"]))))
                else:
                    m.zero_(); v.zero_()
                    steps.add(0)
                self.state[p] = {"step": self._count[0] if self._counted else torch.tensor(float(max(steps) if steps else 0)),
                                 "exp_avg": m, "exp_avg_sq": v}
        if len(steps) > 1:
            raise NotImplementedError("FlatClipAdam: parameters with different step counts")
        self._t = steps.pop() if steps else 0
        if self._counted:
            self._count.fill_(float(self._t))
