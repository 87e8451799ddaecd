"""Linear warm-up + cosine annealing of the LocoVal learning rate.

Mirror of pacer/pacer/learning/scheduler.py:4-75 (class CosineAnnealingLR; the chainable recurrence of
pytorch-lightning-bolts' LinearWarmupCosineAnnealingLR): same constructor, stepped once per training epoch
(common_agent.py:95,209).  Epoch 0 runs at warmup_start_lr, the rate rises linearly to the base rate over warmup_epochs,
then follows half a cosine down to eta_min at max_epochs.
"""
import math

import torch


class CosineAnnealingLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.00001, eta_min=0.00001, last_epoch=-1):
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.warmup_start_lr = warmup_start_lr
        self.eta_min = eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t, w, T = self.last_epoch, self.warmup_epochs, self.max_epochs
        groups = self.optimizer.param_groups
        if t == 0:
            return [self.warmup_start_lr] * len(self.base_lrs)
        if t < w:                                             # one linear increment per epoch
            return [g["lr"] + (b - self.warmup_start_lr) / (w - 1) for b, g in zip(self.base_lrs, groups)]
        if t == w:
            return list(self.base_lrs)
        if (t - 1 - T) % (2 * (T - w)) == 0:                  # restart point of the periodic continuation
            return [g["lr"] + (b - self.eta_min) * (1 - math.cos(math.pi / (T - w))) / 2 for b, g in zip(self.base_lrs, groups)]
        num = 1 + math.cos(math.pi * (t - w) / (T - w))
        den = 1 + math.cos(math.pi * (t - w - 1) / (T - w))
        return [num / den * (g["lr"] - self.eta_min) + self.eta_min for g in groups]
