"""AdamW over one flat parameter / gradient buffer with a device-side gate.

The LocoVal fit steps its optimiser only on rollout steps where some episode finished (amp_continuous_value.py:123-140).
Deciding that on the host costs a device read per step; here the update is computed unconditionally and committed through
`torch.where(gate, new, old)` -- parameters, both moments and the step counter -- so a closed gate leaves the state exactly
as it was (AdamW with a zero gradient would still decay the weights and the moments).  The arithmetic is torch.optim.AdamW's
(decoupled weight decay, bias-corrected moments, eps outside the square root), so an open-gate step equals `AdamW.step()`.
"""
import torch


class GatedFlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        """`params`: the parameters, `flat_grad`: the flat buffer their .grad tensors alias (dist.FlatGradBucket.grads), in the
        same order."""
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat_grad = flat_grad
        n = flat_grad.numel()
        assert n == sum(p.numel() for p in params)
        dev = flat_grad.device
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.steps = torch.zeros((), device=dev)
        self._params = params

    def _flat_params(self):
        return torch.cat([p.detach().reshape(-1) for p in self._params])

    @torch.no_grad()
    def step(self, gate=None):
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        grad = self.flat_grad
        if gate is None:
            gate = torch.ones((), dtype=torch.bool, device=grad.device)
        p = self._flat_params()
        steps = self.steps + 1.0
        p_new = p * (1.0 - lr * wd)
        m = torch.lerp(self.exp_avg, grad, 1.0 - b1)
        v = self.exp_avg_sq * b2 + (1.0 - b2) * grad * grad
        bc1 = 1.0 - torch.pow(torch.as_tensor(b1, device=grad.device, dtype=torch.float64), steps.double())
        bc2 = 1.0 - torch.pow(torch.as_tensor(b2, device=grad.device, dtype=torch.float64), steps.double())
        step_size = (lr / bc1).float()
        denom = v.sqrt() / bc2.sqrt().float() + eps
        p_new = p_new - step_size * (m / denom)
        p = torch.where(gate, p_new, p)
        self.exp_avg = torch.where(gate, m, self.exp_avg)
        self.exp_avg_sq = torch.where(gate, v, self.exp_avg_sq)
        self.steps = torch.where(gate, steps, self.steps)
        o = 0
        for q in self._params:
            q.copy_(p[o:o + q.numel()].view_as(q))
            o += q.numel()
