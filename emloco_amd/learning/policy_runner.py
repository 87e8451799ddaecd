"""Frozen-policy forward for the rollout.

What the reference does per env.step (amp_players.py / common_player.py:166-169 -> rl_games PpoPlayerContinuous
.get_action): `running_mean_std(obs)` (utils/running_mean_std.py:81-83), `a2c_network.eval_actor`
(amp_network_sept_builder.py:79-110: task MLP 1054->512->256 on the task part, concat with the 368 self
observations, actor MLP 624->2048->1024, `mu` head ->69, sigma = const e^-2.9), action = mu (deterministic) or
mu + sigma * N(0,1), clamped to [-1, 1] (`clip_actions`).

Here: 6 launches on the caller's stream, no torch.cat, no intermediate copies:
  1. `emloco_obs_normalize`: self part -> columns [0,368) of the actor input, task part -> a zero-padded,
     16-byte-aligned (E,1056) operand;
  2-3. task MLP: GEMM+bias+ReLU; the second writes straight into columns [368,624) of the actor input (ldc = 624);
  4-6. actor MLP and `mu` head on `emloco_gemm_f32` (fp32 MFMA, split-K where the tile grid would not fill 256 CUs).
Weights are packed once (`_task_mlp.0.weight` padded from K=1054 to 1056 so every operand takes the 16-byte load path).
"""
import torch

from ..predictor import ops
from ..utils.running_mean_std import obs_normalize

_PAD = 4


_KSPLIT_TARGET = int(__import__("os").environ.get("EMLOCO_KSPLIT_TARGET", "512"))       # workgroups a launch should reach (256 CUs)


def _ksplit(m, n, k):
    tiles = ((m + 127) // 128) * ((n + 31) // 32 if n <= 32 else (n + 127) // 128)
    return int(max(1, min(_KSPLIT_TARGET // max(tiles, 1), k // 128, 16)))


def _frozen_image(owner, w, n, k):
    """The piece image of a FROZEN weight (ops.split_image: the matrix cut into the split mode's bf16 pieces once, when first used,
    instead of by every workgroup of every launch of the rollout); None where the image does not apply (narrow heads).  The weights
    of these runners are private copies made at construction and never written again."""
    if n <= 32 or not ops._IMAGE_FROZEN or ops._matmul_precision[0] != "fp32_split":
        return None
    cache = owner.__dict__.setdefault("_images", {})
    key = (w.data_ptr(), n, k)
    if key not in cache:
        cache[key] = ops.split_image(w, n, k, w.stride(0), 0)
    return cache[key]


class FrozenPolicy:
    def __init__(self, network, mean_std, num_envs, device, clip_actions=1.0):
        self.E, self.device, self.clip = num_envs, torch.device(device), float(clip_actions)
        self.self_size, self.task_size = network.self_obs_size, network.task_obs_size
        sd = {k: v.detach().to(self.device, torch.float32).clone().contiguous() for k, v in network.state_dict().items()}      # private copies: frozen for good (the piece images below are cut from them once)

        def layers(prefix):
            idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith(prefix + ".") and k.endswith(".weight")})
            return [(sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"]) for i in idx]

        self.task_layers, self.actor_layers = layers("_task_mlp"), layers("actor_mlp")
        self.mu_w, self.mu_b, self.sigma = sd["mu.weight"], sd["mu.bias"], sd["sigma"]
        # pad the first task layer's K to a multiple of 4 floats (16-byte loads)
        w0, b0 = self.task_layers[0]
        self.task_k = (self.task_size + _PAD - 1) // _PAD * _PAD
        w0p = torch.zeros((w0.shape[0], self.task_k), dtype=torch.float32, device=self.device)
        w0p[:, :self.task_size] = w0
        self.task_layers[0] = (w0p, b0)
        self.mean32 = mean_std.running_mean.to(self.device).float().contiguous()
        self.var32 = mean_std.running_var.to(self.device).float().contiguous()
        self.eps = float(mean_std.epsilon)
        self.exp_sigma = torch.exp(self.sigma)                 # the frozen policy's standard deviation: constant
        self.actor_in_size = self.self_size + self.task_layers[-1][0].shape[0]
        assert self.actor_layers[0][0].shape[1] == self.actor_in_size and self.actor_in_size % _PAD == 0
        E = num_envs
        f = dict(dtype=torch.float32, device=self.device)
        self.task_in = torch.zeros((E, self.task_k), **f)          # pad columns stay zero
        self.actor_in = torch.empty((E, self.actor_in_size), **f)
        self.task_h = [torch.empty((E, w.shape[0]), **f) for w, _ in self.task_layers[:-1]]
        self.actor_h = [torch.empty((E, w.shape[0]), **f) for w, _ in self.actor_layers]
        self.mu = torch.empty((E, self.mu_w.shape[0]), **f)
        self.flops_per_env = 2 * (sum(w.shape[0] * w.shape[1] for w, _ in self.task_layers + self.actor_layers) + self.mu_w.numel())

    @classmethod
    def from_checkpoint(cls, path, network, mean_std, num_envs, device, **kw):
        """rl_games checkpoint: {'model': {'a2c_network.<key>': tensor, ...}, 'running_mean_std': {...}} (common_agent.py:252)."""
        ck = torch.load(path, map_location="cpu")
        sd = {k[len("a2c_network."):]: v for k, v in ck["model"].items() if k.startswith("a2c_network.")}
        network.load_state_dict(sd, strict=False)
        if "running_mean_std" in ck:
            mean_std.load_state_dict(ck["running_mean_std"])
        return cls(network, mean_std, num_envs, device, **kw)

    def _linear(self, x, k, w, b, out, ldc=None, c_off=0, relu=True):
        m, n = x.shape[0], w.shape[0]
        ops.gemm(1, m, n, k, x, x.stride(0), 0, 0, w, w.stride(0), 0, 0, out, ldc if ldc is not None else out.stride(0), 0,
                 bias=b, flags=ops.GEMM_BIAS | (ops.GEMM_RELU if relu else 0), ksplit=_ksplit(m, n, k), c_off=c_off,
                 b_image=_frozen_image(self, w, n, k))

    def act_mean(self, obs):
        """obs (E, self+task) fp32 on the device -> mu (E, actions); the returned tensor is reused by the next call."""
        assert obs.shape == (self.E, self.self_size + self.task_size)
        obs_normalize(obs, self.mean32, self.var32, self.eps, 5.0, split=self.self_size, out0=self.actor_in, out1=self.task_in)
        x, k = self.task_in, self.task_k
        for i, (w, b) in enumerate(self.task_layers):
            if i == len(self.task_layers) - 1:          # last task layer lands in the actor input, columns [self_size, ...)
                self._linear(x, k, w, b, self.actor_in, ldc=self.actor_in_size, c_off=self.self_size)
            else:
                self._linear(x, k, w, b, self.task_h[i])
                x, k = self.task_h[i], w.shape[0]
        x, k = self.actor_in, self.actor_in_size
        for i, (w, b) in enumerate(self.actor_layers):
            self._linear(x, k, w, b, self.actor_h[i])
            x, k = self.actor_h[i], w.shape[0]
        self._linear(x, k, self.mu_w, self.mu_b, self.mu, relu=False)
        return self.mu

    def act(self, obs, deterministic=True, generator=None):
        mu = self.act_mean(obs)
        if deterministic:
            return torch.clamp(mu, -self.clip, self.clip)
        noise = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device, generator=generator)
        return torch.addcmul(mu, self.exp_sigma, noise).clamp_(-self.clip, self.clip)       # three launches on the chain, not five


class FrozenDisc:
    """The AMP discriminator's style reward for the LocoVal rollout (config 2), as the frozen policy above: what the reference
    computes per step with `_amp_input_mean_std(amp_obs)`, `a2c_network.eval_disc` (amp_network_builder.py:81-84: MLP 3090 -> 1024
    -> 512, logits -> 1) and `-log(max(1 - sigmoid(logit), 1e-4)) * disc_reward_scale` (amp_continuous.py:675-692), here as one
    normalise launch into a 16-byte-aligned operand, three GEMM launches with the bias / ReLU epilogue on pre-packed weights and
    preallocated buffers, and the scalar transform on the (E,) logits -- no module dispatch, no per-step allocation."""

    def __init__(self, network, amp_mean_std, num_envs, device, disc_reward_scale=2.0, normalize=True):
        self.E, self.device, self.scale, self.normalize = num_envs, torch.device(device), float(disc_reward_scale), bool(normalize)
        sd = {k: v.detach().to(self.device, torch.float32).clone().contiguous() for k, v in network.state_dict().items()}      # private copies: frozen for good (the piece images below are cut from them once)
        idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("_disc_mlp.") and k.endswith(".weight")})
        self.layers = [(sd[f"_disc_mlp.{i}.weight"], sd[f"_disc_mlp.{i}.bias"]) for i in idx]
        self.logit_w, self.logit_b = sd["_disc_logits.weight"], sd["_disc_logits.bias"]
        w0, b0 = self.layers[0]
        self.in_size = w0.shape[1]
        self.in_k = (self.in_size + _PAD - 1) // _PAD * _PAD
        w0p = torch.zeros((w0.shape[0], self.in_k), dtype=torch.float32, device=self.device)
        w0p[:, :self.in_size] = w0
        self.layers[0] = (w0p, b0)
        self.mean32 = amp_mean_std.running_mean.to(self.device).float().contiguous()
        self.var32 = amp_mean_std.running_var.to(self.device).float().contiguous()
        self.eps = float(amp_mean_std.epsilon)
        f = dict(dtype=torch.float32, device=self.device)
        self.x = torch.zeros((num_envs, self.in_k), **f)              # pad columns stay zero
        # the staged halves below hand the operand from `stage` (caller's stream) to `reward_staged` (possibly another stream, later):
        # a ring of operands, written and read in FIFO order, so that a stage never overwrites what a lagging reward_staged still reads
        self._xs, self._xw, self._xr = [self.x], 0, 0
        self.h = [torch.empty((num_envs, w.shape[0]), **f) for w, _ in self.layers]
        self.logits = torch.empty((num_envs, 1), **f)
        self.floor = torch.tensor(0.0001, device=self.device)

    def _linear(self, x, k, w, b, out, relu):
        m, n = x.shape[0], w.shape[0]
        ops.gemm(1, m, n, k, x, x.stride(0), 0, 0, w, w.stride(0), 0, 0, out, out.stride(0), 0, bias=b,
                 flags=ops.GEMM_BIAS | (ops.GEMM_RELU if relu else 0), ksplit=_ksplit(m, n, k), b_image=_frozen_image(self, w, n, k))

    def logits_of(self, amp_obs):
        x = amp_obs.reshape(amp_obs.shape[0], -1)
        assert x.shape == (self.E, self.in_size) and x.dtype == torch.float32
        if self.normalize:
            self._normalize_padded(x.contiguous(), self.x)
        else:
            self.x[:, :self.in_size].copy_(x)
        return self._logits_from(self.x)

    def _logits_from(self, operand):
        cur, k = operand, self.in_k
        for (w, b), out in zip(self.layers, self.h):
            self._linear(cur, k, w, b, out, True)
            cur, k = out, w.shape[0]
        self._linear(cur, k, self.logit_w, self.logit_b, self.logits, False)
        return self.logits

    def _normalize_padded(self, x, out):
        # emloco_obs_normalize writes `split` columns to out0 with its own leading dimension: the padded operand takes them all
        obs_normalize(x, self.mean32, self.var32, self.eps, 5.0, split=self.in_size, out0=out)

    def reward(self, amp_obs):
        return self._reward_of(self.logits_of(amp_obs))

    def _reward_of(self, logits):
        """-log(max(1 - sigmoid(logit), 1e-4)) * scale (amp_continuous.py:675-692) in one launch (emloco_disc_reward)."""
        import ctypes as C
        from ..sim import current_stream_handle
        out = torch.empty(self.E, dtype=torch.float32, device=self.device)
        ops._chk(ops._lib().emloco_disc_reward(self.E, C.c_void_p(logits.data_ptr()), C.c_float(self.scale), C.c_void_p(out.data_ptr()),
                                               current_stream_handle(self.device)), "emloco_disc_reward")
        return out

    # The same reward in two halves, for a rollout that keeps the discriminator off the chain between two rigid-body steps: `stage`
    # (one launch on the caller's stream) takes what it needs out of the step's AMP observations -- the normalised, padded GEMM
    # operand -- before the resets overwrite them; `reward_staged` (three GEMMs + the scalar transform, any stream ordered behind the
    # stage) reads nothing of the task's.  Same launches on the same values as `reward`.
    # The operand is a ring (`set_stage_ring(n)`; one buffer by default): stage number t writes buffer t mod n, the t-th reward_staged
    # reads it.  A caller that runs reward_staged on another stream sizes the ring to the number of stages it lets run ahead of the
    # rewards (LocoValRollout: its ring of staging sets, whose hand-back -- the host waits for the fit of n steps ago before the flags
    # launch -- then covers this operand as well; with ONE buffer, stage(t + 1) on the main stream raced the first GEMM of
    # reward_staged(t) on the side stream whenever the side stream lagged: round-4 review).
    def set_stage_ring(self, n):
        n = max(1, int(n))
        while len(self._xs) < n:
            self._xs.append(torch.zeros_like(self.x))              # pad columns stay zero
        del self._xs[n:]
        self._xw = self._xr = 0

    def stage(self, amp_obs):
        x = amp_obs.reshape(amp_obs.shape[0], -1)
        assert x.shape == (self.E, self.in_size) and x.dtype == torch.float32
        dst = self._xs[self._xw % len(self._xs)]
        self._xw += 1
        if self.normalize:
            self._normalize_padded(x.contiguous(), dst)
        else:
            dst[:, :self.in_size].copy_(x)

    def reward_staged(self):
        assert self._xr < self._xw, "reward_staged without a stage"
        src = self._xs[self._xr % len(self._xs)]
        self._xr += 1
        return self._reward_of(self._logits_from(src))
