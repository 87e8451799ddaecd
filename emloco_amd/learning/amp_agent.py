"""AMPAgent -- PPO + adversarial-motion-prior training of the PACER policy (configs[1]) without rl_games.

Mirror of /root/reference/pacer/pacer/learning/amp_continuous.py (AMPAgent: play_steps :98-189, train_epoch :222-321,
calc_gradients :335-479, losses :515-598, disc reward :675-692) over common_agent.py (GAE :573-587, bound loss :594-603,
actor / critic losses :657-683, advantages :685-696, prepare_dataset :426-468), amp_datasets.py, replay_buffer.py and
amp_models.py (model forward incl. the per-joint AMP dropout :62-90).  The pieces that come from rl_games==1.1.4 in the
reference (A2CBase buffers, ModelA2CContinuousLogStd's neglogp / entropy, policy_kl, Adam set-up) are restated from their
textbook definitions -- parity with rl_games internals is UNPINNED; the losses are pinned by a
plain-torch double-backward restatement in tests/test_gpu_policy.py.

Every Linear of actor / critic / discriminator runs on `emloco_gemm_f32` through `predictor.ops.linear` (forward and
backward).  The discriminator gradient penalty needs d/dtheta of |dD/dx|^2; instead of a double backward through custom
autograd Functions the input gradient is written out as its own small network (ReLU masks are constants a.e., so the
result is identical):  dD/dx = W1^T (m1 o (W2^T (m2 o w3)))  -- three more GEMMs whose autograd reaches W1, W2, w3.
Multi-GPU: one flat gradient bucket all-reduced (averaged) per minibatch over RCCL, KL averaged, as the reference does
with Horovod (amp_continuous.py:436-444, common_agent.py:179-180).
"""
import contextlib
import os
import math
import time

import numpy as np
import torch
import torch.nn as nn

from ..dist import FlatGradBucket, all_reduce_mean_scalar, broadcast_parameters, sync_running_mean_std
from ..predictor import ops
from ..utils.running_mean_std import RunningMeanStd
from . import ppo_heads
from .amp_network_sept_builder import AMPSeptBuilder


class ReplayBuffer:
    """replay_buffer.py:3-92."""

    def __init__(self, buffer_size, device):
        self._head, self._total_count, self._buffer_size, self._device = 0, 0, buffer_size, device
        self._data_buf = None
        self._sample_idx = torch.randperm(buffer_size)
        self._sample_head = 0

    def get_buffer_size(self):
        return self._buffer_size

    def get_total_count(self):
        return self._total_count

    def store(self, data_dict):
        if self._data_buf is None:
            self._data_buf = {k: torch.zeros((self._buffer_size,) + tuple(v.shape[1:]), dtype=v.dtype, device=self._device)
                              for k, v in data_dict.items()}
        n = next(iter(data_dict.values())).shape[0]
        assert n <= self._buffer_size
        for key, buf in self._data_buf.items():
            store_n = min(n, self._buffer_size - self._head)
            buf[self._head:self._head + store_n] = data_dict[key][:store_n]
            if n - store_n > 0:
                buf[0:n - store_n] = data_dict[key][store_n:]
        self._head = (self._head + n) % self._buffer_size
        self._total_count += n

    def sample(self, n):
        idx = torch.arange(self._sample_head, self._sample_head + n) % self._buffer_size
        rand_idx = self._sample_idx[idx]
        if self._total_count < self._buffer_size:
            rand_idx = rand_idx % self._head
        rand_idx = rand_idx.to(self._device)
        samples = {k: v[rand_idx] for k, v in self._data_buf.items()}
        self._sample_head += n
        if self._sample_head >= self._buffer_size:
            self._sample_idx[:] = torch.randperm(self._buffer_size)
            self._sample_head = 0
        return samples


def neglogp(x, mean, std, logstd):
    """ModelA2CContinuousLogStd.neglogp (rl_games 1.1.4)."""
    return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)


_COEF_CACHE = {}


def _weighted_sum(terms, coef):
    """sum_i coef_i * terms_i of 0-dim tensors as one stack and one dot; the coefficient vector is made once per (values, device)."""
    key = (tuple(float(c) for c in coef), str(terms[0].device))
    c = _COEF_CACHE.get(key)
    if c is None:
        c = _COEF_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=terms[0].device)
    return torch.dot(torch.stack([t.float().reshape(()) for t in terms]), c)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """torch_ext.policy_kl (rl_games 1.1.4), reduced."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    return (c1 + c2 - 0.5).sum(dim=-1).mean()


def amp_dropout_draw(B, num_masks=3, num_joints=19):
    """The host draw of amp_dropout_mask: (19, B, num_masks) uniforms in the reference's stream order (amp_models.py:85-89)."""
    if (B * num_masks) % 16 == 0:
        return torch.rand(num_joints, B, num_masks)
    return torch.stack([torch.rand(B, num_masks) for _ in range(num_joints)])


def amp_dropout_expand(u, steps, feat=206, dropout_rate=0.3):
    """(19, B, M) uniforms ON THE DEVICE -> the (B, steps * 206, M) keep mask (device ops only: capturable in a HIP graph)."""
    assert feat == 206
    dof_off, num_joints = 12, 19
    vel_off = dof_off + num_joints * 6
    B, M, device = u.shape[1], u.shape[2], u.device
    keep = (u > dropout_rate).permute(1, 0, 2).float()
    keep = torch.cat([keep, torch.ones(B, 1, M, device=device)], dim=1)
    key = str(device)
    if key not in _AMP_COL:
        col = torch.full((feat,), num_joints, dtype=torch.long)
        for j in range(num_joints):
            col[dof_off + j * 6:dof_off + j * 6 + 6] = j
            col[vel_off + j * 3:vel_off + j * 3 + 3] = j
        _AMP_COL[key] = col.to(device)
    return keep[:, _AMP_COL[key], :].repeat(1, steps, 1)


def amp_dropout_mask(B, steps, feat, num_masks=3, dropout_rate=0.3, device="cpu"):
    """amp_models.py:62-90 for the 206-wide AMP row: whole joints (6 rotation + 3 velocity values) are dropped together.
    The draws are the reference's (19 x torch.rand(B, num_masks) from the CPU generator, in its order); the (B, steps * 206,
    num_masks) mask itself is expanded ON THE DEVICE from the B x 19 x num_masks keep bits -- the reference assembles the 76 MB
    tensor of a 2 048-row minibatch on the host and uploads it every optimiser step (measured here: 60 - 200 ms of an 80 ms step)."""
    return amp_dropout_expand(amp_dropout_draw(B, num_masks).to(device), steps, feat, dropout_rate)


_AMP_COL = {}


def disc_first_layer(net):
    """The discriminator's first Linear and its weight with the input width padded to a multiple of four (the AMP observation is 3 090
    wide; the 16-byte-load GEMM kernels want 3 092): ONE pad per optimiser step, shared by every evaluation of the step (ops.linear pads x
    and W per call otherwise -- two fill + copy pairs per evaluation)."""
    lin0 = next(m for m in net._disc_mlp if isinstance(m, nn.Linear))
    pad = (-lin0.in_features) % 4
    Wp = torch.nn.functional.pad(lin0.weight, (0, pad)) if pad else lin0.weight
    return lin0, Wp


def disc_eval_padded(net, xp, first):
    """net.eval_disc on an input that is already padded to the first layer's padded width (pad columns zero)."""
    lin0, Wp = first
    lin = [m for m in net._disc_mlp if isinstance(m, nn.Linear)]
    h = ops.linear(xp, Wp, lin0.bias, relu=True)
    for m in lin[1:]:
        h = ops.linear(h, m.weight, m.bias, relu=True)
    return ops.linear(h, net._disc_logits.weight, net._disc_logits.bias)


def disc_forward_with_grad_penalty(net, amp_obs_demo, input_mask=None, padded=None):
    """Discriminator logits on the demo batch and mean |dD/dx|^2 (amp_continuous.py:560-583) with the input gradient
    written as an explicit network (see module docstring).  `input_mask` is the AMP dropout mask applied to the input.
    padded = (first, xp): the step's padded first-layer weight (`disc_first_layer`) and a zero-padded buffer xp (B, K + pad) the masked
    input is written into -- no per-call padding of the operands."""
    acts = [m for m in net._disc_mlp if not isinstance(m, nn.Linear)]
    if any(not isinstance(m, nn.ReLU) for m in acts):
        # the explicit gradient network below is the ReLU one (the shipped config, amp_humanoid_smpl_sept_task.yaml: disc
        # activation relu); another activation needs its own derivative here
        raise NotImplementedError("disc_forward_with_grad_penalty: the discriminator activation must be relu, got "
                                  + ", ".join(sorted({type(m).__name__ for m in acts})))
    lin = [m for m in net._disc_mlp if isinstance(m, nn.Linear)]
    K = lin[0].in_features
    if padded is not None:
        (lin0, Wp), xp = padded
        if input_mask is None:
            xp[:, :K].copy_(amp_obs_demo)
        else:
            torch.mul(amp_obs_demo, input_mask, out=xp[:, :K])
        x, W0 = xp, Wp
    else:
        x = amp_obs_demo if input_mask is None else amp_obs_demo * input_mask
        W0 = lin[0].weight
    hs, h = [], x
    for i, m in enumerate(lin):
        h = ops.linear(h, W0 if i == 0 else m.weight, m.bias, relu=True)
        hs.append(h)
    logits = ops.linear(h, net._disc_logits.weight, net._disc_logits.bias)
    g = ops.relu_mask(net._disc_logits.weight.expand_as(hs[-1]), hs[-1])     # (B, units[-1])
    for i in range(len(lin) - 1, 0, -1):
        g = ops.relu_mask(ops.linear(g, lin[i].weight.t()), hs[i - 1])
    gx = ops.linear(g, W0.t())
    if padded is not None:
        gx = gx[:, :K]                                       # (the pad columns are exact zeros: W0's pad columns are)
    if input_mask is not None:
        gx = gx * input_mask
    return logits, torch.mean(torch.sum(torch.square(gx), dim=-1))


class AMPAgent:
    def __init__(self, vec_env, cfg_train, seed=0):
        self.vec_env = vec_env
        env = vec_env.env if hasattr(vec_env, "env") else vec_env
        self.env, self.task = env, env.task
        if getattr(self.task, "amp_ring", False):
            self.task.enable_amp_ring(False)     # this learner reads infos["amp_obs"] every step: the reference's layout (a rollout loop may have left the ring on)
        task = self.task
        self.device = torch.device(task.device)
        params = cfg_train["params"]
        c = self.config = params["config"]
        self.num_actors = task.num_envs
        self.horizon_length = int(c["horizon_length"])
        self.batch_size = self.horizon_length * self.num_actors
        self.minibatch_size = min(int(c["minibatch_size"]), self.batch_size)
        assert self.batch_size % self.minibatch_size == 0
        self.mini_epochs_num = int(c["mini_epochs"])
        self.gamma, self.tau, self.e_clip = c["gamma"], c["tau"], c["e_clip"]
        self.critic_coef, self.entropy_coef, self.bounds_loss_coef = c["critic_coef"], c["entropy_coef"], c["bounds_loss_coef"]
        self.clip_value, self.truncate_grads, self.grad_norm = c["clip_value"], c["truncate_grads"], c["grad_norm"]
        self.normalize_input, self.normalize_value = c["normalize_input"], c["normalize_value"]
        self.normalize_advantage = c["normalize_advantage"]
        self.last_lr = float(c["learning_rate"])
        self._task_reward_w, self._disc_reward_w = c["task_reward_w"], c["disc_reward_w"]
        self._amp_batch_size = int(c["amp_batch_size"])
        self._amp_minibatch_size = min(int(c["amp_minibatch_size"]), self.minibatch_size)
        self._disc_coef, self._disc_logit_reg = c["disc_coef"], c["disc_logit_reg"]
        self._disc_grad_penalty, self._disc_weight_decay = c["disc_grad_penalty"], c["disc_weight_decay"]
        self._disc_reward_scale = c["disc_reward_scale"]
        self._normalize_amp_input = c.get("normalize_amp_input", True)
        self._amp_dropout = c.get("amp_dropout", False)
        self.motion_sym_loss = bool(getattr(task, "motion_sym_loss", False))
        self.sym_loss_coef = task.cfg["env"].get("sym_loss_coef", 1) if hasattr(task, "cfg") else 1
        obs_size, self_size, amp_size = task.get_obs_size(), task.get_self_obs_size(), task.get_num_amp_obs()
        self.actions_num = task.num_actions
        self.running_mean_std = RunningMeanStd((obs_size,)).to(self.device)
        self.value_mean_std = RunningMeanStd((1,)).to(self.device)
        self._amp_input_mean_std = RunningMeanStd((amp_size,)).to(self.device)
        torch.manual_seed(seed)
        b = AMPSeptBuilder()
        b.load(params["network"])
        self.a2c_network = b.build("amp", actions_num=self.actions_num, input_shape=(obs_size,), num_seqs=self.num_actors, value_size=1,
                                   amp_input_shape=(amp_size,), self_obs_size=self_size, task_obs_size=obs_size - self_size,
                                   task_obs_size_detail=task.get_task_obs_size_detail(), mean_std=self.running_mean_std).to(self.device)
        # hvd.setup_algo (common_agent.py:165-166): every rank starts from rank 0's networks and (initial) statistics
        broadcast_parameters(self.a2c_network, self.running_mean_std, self.value_mean_std, self._amp_input_mean_std)
        # The optimiser step as ONE HIP graph (use_graph): an update step is ~650 launches of 4 - 80 us each -- 8 ms of kernel time that
        # the host needs 40 ms to issue through autograd, torch ops and ctypes.  Captured once (static minibatch buffers, the dropout
        # draw and the shuffled indices filled from the host outside the graph, Adam with device-side step counters) and replayed
        # for the epoch's remaining minibatches and every later epoch.  Single rank only: a gloo all-reduce cannot be captured.
        self.use_graph = (self.device.type == "cuda" and os.environ.get("EMLOCO_PPO_GRAPH", "1") != "0")
        self._graph, self._g_in, self._g_u, self._g_acc, self._g_keys = None, None, None, None, None
        # the arms' streams exist from construction on (round 6): which hardware queue a stream lands on depends on how many streams
        # the process created before it, so they are made here, in a fixed order, not at the first captured step
        self._g_branch = tuple(torch.cuda.Stream(device=self.device) for _ in range(3)) if self.use_graph else None
        # clip_grad_norm_ + Adam (common_agent.py:573-603 through amp_continuous.py:440-445) on flat buffers: 4 launches
        # (`emloco_adam_clip_flat_counted`: the step count lives on the device, so the captured step replays as the next one) where
        # torch's foreach implementations issue ~30 passes over the 11 M parameters.  EMLOCO_PPO_FLAT_ADAM=0 / a CPU device: torch's own.
        self._flat_adam = self.device.type == "cuda" and os.environ.get("EMLOCO_PPO_FLAT_ADAM", "1") != "0"
        # the loss heads as fused launches (learning/ppo_heads.py); EMLOCO_PPO_HEADS=0 / CPU tensors: the torch expressions
        self._fused_heads = self.device.type == "cuda" and os.environ.get("EMLOCO_PPO_HEADS", "1") != "0"
        self._head_kl = None
        # EMLOCO_PPO_GATHER_GRADS=1: autograd keeps the step's gradients and one launch per 96 tensors gathers them into the flat bucket
        # (dist.FlatGradBucket.release / gather, `emloco_gather_flat`: what the predictor's train step does) instead of one accumulation
        # launch per parameter.  Off here: the large weights already accumulate in their GEMMs' epilogues (mark_direct_grad below) and the
        # captured step pays nothing per launch -- measured 3.14-3.17 ms per optimiser step without, 3.23-3.24 with
        # (profiles/r06_ab_ppo_gather.txt)
        self._gather_grads = self.device.type == "cuda" and os.environ.get("EMLOCO_PPO_GATHER_GRADS", "0") == "1"
        if self._flat_adam:
            from ..predictor.fused_adam import FlatClipAdam
            trainable = [p for p in self.a2c_network.parameters() if p.requires_grad]
            self.bucket = FlatGradBucket(trainable, align=FlatClipAdam.ALIGN)
            self.optimizer = FlatClipAdam(trainable, float(self.last_lr), eps=1e-08, weight_decay=0.0, bucket=self.bucket, counted=True)
        else:
            self.optimizer = torch.optim.Adam(self.a2c_network.parameters(), float(self.last_lr), eps=1e-08, weight_decay=0.0,
                                              capturable=self.use_graph)
            self.bucket = FlatGradBucket([p for p in self.a2c_network.parameters() if p.requires_grad])
        # Round 6: one stacked actor evaluation per step (EMLOCO_PPO_STACK_ACTOR=0: three, as the reference), and the weight gradients of
        # the layers that are used on ONE stream and by nothing but their GEMMs -- actor and critic trunks, the value head -- accumulate
        # straight into the flat bucket (ops.mark_direct_grad; EMLOCO_PPO_DIRECT_GRAD=0: through autograd).  Not the task MLP (evaluated
        # on the actor's and on the critic's stream) and not the discriminator (weight decay, logit regulariser and the gradient-penalty
        # network reach its weights through autograd's accumulation, which runs on a stream of its own choosing).
        self._stack_actor = os.environ.get("EMLOCO_PPO_STACK_ACTOR", "1") != "0"
        # (the discriminator's operands born padded -- persistent zero-padded input buffers, one weight pad per step: measured, no gain
        # with the arms graph (3.46-3.53 against 3.40-3.48 ms; the one-chain graph 4.43 -> 4.37): the discriminator's arm is not the
        # critical one.  Off unless EMLOCO_PPO_DISC_PADDED=1)
        self._disc_padded = os.environ.get("EMLOCO_PPO_DISC_PADDED", "0") == "1"
        if self.device.type == "cuda" and os.environ.get("EMLOCO_PPO_DIRECT_GRAD", "1") != "0" and (self._stack_actor or not self.motion_sym_loss):
            from ..predictor import ops as _ops
            net_ = self.a2c_network
            direct = [m.weight for m in list(net_.actor_mlp.modules()) + list(net_.critic_mlp.modules()) if isinstance(m, nn.Linear)]
            _ops.mark_direct_grad(direct + [net_.value.weight])
        self._amp_obs_demo_buffer = ReplayBuffer(int(c["amp_obs_demo_buffer_size"]), self.device)
        self._amp_replay_buffer = ReplayBuffer(int(c["amp_replay_buffer_size"]), self.device)
        self._amp_replay_keep_prob = c["amp_replay_keep_prob"]
        H, E, f = self.horizon_length, self.num_actors, dict(dtype=torch.float32, device=self.device)
        self.buf = {"obses": torch.zeros((H, E, obs_size), **f), "next_obses": torch.zeros((H, E, obs_size), **f),
                    "flip_obs": torch.zeros((H, E, obs_size), **f), "rewards": torch.zeros((H, E, 1), **f),
                    "values": torch.zeros((H, E, 1), **f), "next_values": torch.zeros((H, E, 1), **f),
                    "neglogpacs": torch.zeros((H, E), **f), "dones": torch.zeros((H, E), dtype=torch.uint8, device=self.device),
                    "actions": torch.zeros((H, E, self.actions_num), **f), "mus": torch.zeros((H, E, self.actions_num), **f),
                    "sigmas": torch.zeros((H, E, self.actions_num), **f), "amp_obs": torch.zeros((H, E, amp_size), **f)}
        self.current_rewards = torch.zeros((E, 1), **f)
        self.current_lengths = torch.zeros(E, **f)
        self.done_indices = torch.arange(E, device=self.device)
        self._idx_buf = torch.randperm(self.batch_size)
        self.epoch_num, self.frame = 0, 0
        self.train_result = {}
        self._init_amp_demo_buf()

    # ------------------------------------------------------------------ helpers
    def set_eval(self):
        self.a2c_network.eval(); self.running_mean_std.eval(); self.value_mean_std.eval(); self._amp_input_mean_std.eval()

    def set_train(self):
        self.a2c_network.train(); self.running_mean_std.train(); self.value_mean_std.train(); self._amp_input_mean_std.train()

    def _preproc_obs(self, obs):
        return self.running_mean_std(obs) if self.normalize_input else obs

    def _preproc_amp_obs(self, amp_obs):
        return self._amp_input_mean_std(amp_obs) if self._normalize_amp_input else amp_obs

    def _fetch_amp_obs_demo(self, n):
        return self.env.fetch_amp_obs_demo(n)

    def _init_amp_demo_buf(self):
        for _ in range(int(np.ceil(self._amp_obs_demo_buffer.get_buffer_size() / self._amp_batch_size))):
            self._amp_obs_demo_buffer.store({"amp_obs": self._fetch_amp_obs_demo(self._amp_batch_size)})

    def get_action_values(self, obs):
        """A2CBase.get_action_values + ModelA2CContinuousLogStd.forward(is_train=False)."""
        net = self.a2c_network
        with torch.no_grad():
            x = self._preproc_obs(obs)
            mu, logstd = net.eval_actor(x)
            value = net.eval_critic(x)
            sigma = torch.exp(logstd)
            action = mu + sigma * torch.randn_like(mu)
            res = {"neglogpacs": neglogp(action, mu, sigma, logstd), "values": value, "actions": action, "mus": mu, "sigmas": sigma}
            if self.normalize_value:
                res["values"] = self.value_mean_std(res["values"], True)
        return res

    def _eval_critic(self, obs):
        with torch.no_grad():
            value = self.a2c_network.eval_critic(self._preproc_obs(obs))
            return self.value_mean_std(value, True) if self.normalize_value else value

    def _calc_disc_rewards(self, amp_obs):
        with torch.no_grad():
            shp = amp_obs.shape
            logits = self.a2c_network.eval_disc(self._preproc_amp_obs(amp_obs.reshape(-1, shp[-1]))).reshape(*shp[:-1], 1)
            prob = 1 / (1 + torch.exp(-logits))
            return -torch.log(torch.maximum(1 - prob, torch.tensor(0.0001, device=self.device))) * self._disc_reward_scale

    def discount_values(self, mb_fdones, mb_values, mb_rewards, mb_next_values):
        lastgaelam = 0
        mb_advs = torch.zeros_like(mb_rewards)
        for t in reversed(range(self.horizon_length)):
            not_done = (1.0 - mb_fdones[t]).unsqueeze(1)
            delta = mb_rewards[t] + self.gamma * mb_next_values[t] - mb_values[t]
            lastgaelam = delta + self.gamma * self.tau * not_done * lastgaelam
            mb_advs[t] = lastgaelam
        return mb_advs

    # ------------------------------------------------------------------ rollout
    def play_steps(self):
        self.set_eval()
        env, task, buf = self.env, self.task, self.buf
        terminated_flags = torch.zeros(self.num_actors, device=self.device)
        reward_raw = None
        with torch.no_grad():
            for n in range(self.horizon_length):
                if self.done_indices.numel():
                    env.reset(self.done_indices)
                obs = task.obs_buf
                buf["obses"][n] = obs
                res = self.get_action_values(obs)
                for k in ("neglogpacs", "values", "actions", "mus", "sigmas"):
                    buf[k][n] = res[k]
                obs, rewards, dones, infos = self.vec_env.step(torch.clamp(res["actions"], -1.0, 1.0))
                buf["rewards"][n] = rewards.unsqueeze(-1) if rewards.dim() == 1 else rewards
                buf["next_obses"][n] = obs
                buf["dones"][n] = dones.to(torch.uint8)
                buf["amp_obs"][n] = infos["amp_obs"]
                if self.motion_sym_loss:
                    buf["flip_obs"][n] = infos["flip_obs"]
                terminated = infos["terminate"].float()
                terminated_flags += terminated
                rr = infos["reward_raw"].mean(dim=0)
                reward_raw = rr if reward_raw is None else reward_raw + rr
                next_vals = self._eval_critic(obs) * (1.0 - terminated.unsqueeze(-1))
                buf["next_values"][n] = next_vals
                self.current_rewards += buf["rewards"][n]
                self.current_lengths += 1
                self.done_indices = dones.nonzero(as_tuple=False).flatten()
                not_dones = 1.0 - dones.float()
                self.current_rewards = self.current_rewards * not_dones.unsqueeze(1)
                self.current_lengths = self.current_lengths * not_dones
            mb_fdones = buf["dones"].float()
            amp_rewards = self._calc_disc_rewards(buf["amp_obs"])
            mb_rewards = self._task_reward_w * buf["rewards"] + self._disc_reward_w * amp_rewards
            mb_advs = self.discount_values(mb_fdones, buf["values"], mb_rewards, buf["next_values"])
            mb_returns = mb_advs + buf["values"]
            flat = lambda t: t.transpose(0, 1).reshape(t.shape[0] * t.shape[1], *t.shape[2:])      # swap_and_flatten01
            batch = {k: flat(v) for k, v in buf.items()}
            batch["returns"] = flat(mb_returns)
            batch["disc_rewards"] = flat(amp_rewards)
            batch["played_frames"] = self.batch_size
            batch["terminated_flags"] = terminated_flags
            batch["reward_raw"] = reward_raw / self.horizon_length
        return batch

    # ------------------------------------------------------------------ losses
    def _actor_loss(self, old_logp, logp, advantage, e_clip):
        ratio = torch.exp(old_logp - logp)
        a_loss = torch.max(-advantage * ratio, -advantage * torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip))
        return {"actor_loss": a_loss, "actor_clipped": (torch.abs(ratio - 1.0) > e_clip).detach()}

    def _critic_loss(self, value_preds, values, e_clip, returns, clip_value):
        if clip_value:
            clipped = value_preds + (values - value_preds).clamp(-e_clip, e_clip)
            return {"critic_loss": torch.max((values - returns) ** 2, (clipped - returns) ** 2)}
        return {"critic_loss": (returns - values) ** 2}

    def bound_loss(self, mu):
        if self.bounds_loss_coef is None:
            return torch.zeros(mu.shape[0], device=mu.device)
        return (torch.clamp_max(mu + 1.0, 0.0) ** 2 + torch.clamp_min(mu - 1.0, 0.0) ** 2).sum(dim=-1)

    def _sym_loss(self, flip_obs, orig_obs):
        B = flip_obs.shape[0]
        if getattr(self, "_sym_idx", None) is None or self._sym_idx.device != flip_obs.device:
            self._sym_idx = torch.as_tensor(self.task.left_to_right_index_action, dtype=torch.long, device=flip_obs.device)   # (an index list would be uploaded every step)
        idx = self._sym_idx
        # (one evaluation of the 2 B stacked rows: the network is row-wise, the GEMMs twice as tall -- amp_continuous.py evaluates twice)
        both, _ = self.a2c_network.eval_actor(torch.cat([flip_obs, orig_obs], dim=0))
        return self._sym_from_actions(both[:B], both[B:])

    def _sym_from_actions(self, flip_a, orig_a):
        """The symmetry loss (amp_continuous.py:405-418) from the actor's means on the flipped and on the next observations."""
        B = flip_a.shape[0]
        if getattr(self, "_sym_idx", None) is None or self._sym_idx.device != flip_a.device:
            self._sym_idx = torch.as_tensor(self.task.left_to_right_index_action, dtype=torch.long, device=flip_a.device)
        idx = self._sym_idx
        if getattr(self, "_sym_sign", None) is None or self._sym_sign.device != orig_a.device:
            self._sym_sign = torch.tensor([-1.0, 1.0, -1.0], device=orig_a.device)      # (made once: no host-to-device copy in the step)
        orig_a = orig_a.view(B, -1, 3) * self._sym_sign
        orig_a = orig_a[..., idx, :]
        return {"sym_loss": (orig_a.reshape(B, -1) - flip_a).pow(2).mean(dim=-1) * 50}

    def _disc_loss(self, disc_agent_logit, disc_demo_logit, grad_penalty):
        if self._fused_heads and disc_agent_logit.is_cuda:
            bce_agent, agent_acc, bce_demo, demo_acc = ppo_heads.disc_head(disc_agent_logit, disc_demo_logit)
            logit_loss = torch.sum(torch.square(self.a2c_network.get_disc_logit_weights()))
            terms, coef = [bce_agent, bce_demo, logit_loss, grad_penalty], [0.5, 0.5, self._disc_logit_reg, self._disc_grad_penalty]
            if self._disc_weight_decay != 0:
                terms += [torch.sum(torch.square(w)) for w in self.a2c_network.get_disc_weights()]
                coef += [self._disc_weight_decay] * (len(terms) - 4)
            return {"disc_loss": _weighted_sum(terms, coef), "disc_grad_penalty": grad_penalty.detach(), "disc_logit_loss": logit_loss.detach(),
                    "disc_agent_acc": agent_acc.detach(), "disc_demo_acc": demo_acc.detach()}
        else:
            bce = torch.nn.BCEWithLogitsLoss()
            disc_loss = 0.5 * (bce(disc_agent_logit, torch.zeros_like(disc_agent_logit)) + bce(disc_demo_logit, torch.ones_like(disc_demo_logit)))
            agent_acc, demo_acc = (disc_agent_logit < 0).float().mean(), (disc_demo_logit > 0).float().mean()
        logit_loss = torch.sum(torch.square(self.a2c_network.get_disc_logit_weights()))
        disc_loss = disc_loss + self._disc_logit_reg * logit_loss + self._disc_grad_penalty * grad_penalty
        if self._disc_weight_decay != 0:
            # (the sum of the per-matrix sums: the reference concatenates the flattened weights first -- 3.7 M floats copied every step)
            wd = sum(torch.sum(torch.square(w)) for w in self.a2c_network.get_disc_weights())
            disc_loss = disc_loss + self._disc_weight_decay * wd
        return {"disc_loss": disc_loss, "disc_grad_penalty": grad_penalty.detach(), "disc_logit_loss": logit_loss.detach(),
                "disc_agent_acc": agent_acc.detach(), "disc_demo_acc": demo_acc.detach()}

    def compute_loss(self, d, dropout_masks=None, branch_streams=None):
        """The scalar of calc_gradients (amp_continuous.py:335-425) for one minibatch dict `d`.

        branch_streams = (critic, discriminator, symmetry-loss stream): the networks are independent until their losses are added -- the
        critic, the discriminator (three evaluations + the gradient penalty) and the symmetry loss (two more actor evaluations) are
        issued on streams of their own, forked from and joined to the caller's, and only their scalar losses cross.  Autograd runs a node's backward on the stream of its forward, so
        the backward passes fork the same way.  A 2 048-row minibatch is 128-256 tiles per GEMM on 256 CUs: one chain leaves half the
        chip idle, three fill it (inside the captured optimiser step the branches are parallel arms of the graph)."""
        net = self.a2c_network
        main = torch.cuda.current_stream(self.device) if branch_streams is not None else None
        s_c, s_d, s_s = branch_streams if branch_streams is not None else (None, None, None)

        def on(stream):
            if stream is None:
                return contextlib.nullcontext()
            stream.wait_stream(main)
            return torch.cuda.stream(stream)

        if self._amp_dropout and dropout_masks is None:
            steps = self.task._num_amp_obs_steps
            dropout_masks = amp_dropout_mask(self._amp_minibatch_size, steps, d["amp_obs"].shape[1] // steps, device=d["amp_obs"].device)
        # In training mode a RunningMeanStd UPDATES its statistics with the batch it is handed, so the order of the calls on ONE normaliser
        # is part of the arithmetic (the reference's: obs, the three AMP batches, flipped, next: amp_continuous.py:345-352,405) -- and two
        # streams updating one normaliser at once would be a race: each normaliser is called from one stream only, in that order.
        # Round 6 -- order of issue.  The two normalisers are independent objects: the AMP one sees its three batches in the reference's
        # order on the DISCRIMINATOR's arm (the only reader of its results), the observation one sees obs, flipped, next in the
        # reference's order on the caller's stream.  Each arm forks as soon as its inputs exist: the discriminator before anything
        # else, the critic behind the normalised observations and ahead of the flipped / next ones (before: all six normaliser chains --
        # 36 launches -- on the stream every arm forks from, ahead of every fork).
        n_amp = self._amp_minibatch_size
        stack_actor = self.motion_sym_loss and self._stack_actor and d["obs"].is_cuda
        if branch_streams is not None and dropout_masks is not None:      # (allocated on the caller's stream, read on an arm's)
            dropout_masks.record_stream(s_d)
        with on(s_d):
            amp_obs = self._preproc_amp_obs(d["amp_obs"][0:n_amp])
            amp_replay = self._preproc_amp_obs(d["amp_obs_replay"][0:n_amp])
            amp_demo = self._preproc_amp_obs(d["amp_obs_demo"][0:n_amp])
            m = (lambda i: dropout_masks[..., i]) if dropout_masks is not None else (lambda i: None)
            mul = lambda x, k: x if k is None else x * k
            K_amp = amp_obs.shape[1]
            if amp_obs.is_cuda and K_amp % 4 and self._disc_padded:
                # (round 6) the discriminator's operands are BORN padded: persistent zero-padded buffers the masked rows are written into
                # (agent | replay stacked, demo), the first layer's weight padded once per step -- was: F.pad of x and of W inside every
                # ops.linear call on the 3 090-wide input, and a torch.cat for the stacking
                Kp = K_amp + (-K_amp) % 4
                if getattr(self, "_disc_xs", None) is None or self._disc_xs.shape != (2 * n_amp, Kp) or self._disc_xs.device != amp_obs.device:
                    self._disc_xs = torch.zeros(2 * n_amp, Kp, device=amp_obs.device)
                    self._disc_xd = torch.zeros(n_amp, Kp, device=amp_obs.device)
                xs, first = self._disc_xs, disc_first_layer(net)
                for rows, src, k in ((xs[:n_amp, :K_amp], amp_obs, m(0)), (xs[n_amp:, :K_amp], amp_replay, m(1))):
                    if k is None:
                        rows.copy_(src)
                    else:
                        torch.mul(src, k, out=rows)
                disc_agent_replay_logit = disc_eval_padded(net, xs, first)
                disc_demo_logit, grad_pen = disc_forward_with_grad_penalty(net, amp_demo, m(2), padded=(first, self._disc_xd))
            else:
                # (agent and replay rows through the discriminator as ONE stacked batch: row-wise network, the loss wants them stacked anyway)
                disc_agent_replay_logit = net.eval_disc(torch.cat([mul(amp_obs, m(0)), mul(amp_replay, m(1))], dim=0))
                disc_demo_logit, grad_pen = disc_forward_with_grad_penalty(net, amp_demo, m(2))
            disc_info = self._disc_loss(disc_agent_replay_logit, disc_demo_logit, grad_pen)
        obs = self._preproc_obs(d["obs"])
        if branch_streams is not None:
            obs.record_stream(s_c)
        heads = self._fused_heads and obs.is_cuda
        with on(s_c):
            values = net.eval_critic(obs)
            if heads:
                c_loss = ppo_heads.critic_head(values, d["old_values"], d["returns"], self.e_clip, self.clip_value)
            else:
                c_info = self._critic_loss(d["old_values"], values, self.e_clip, d["returns"], self.clip_value)
                c_loss = torch.mean(c_info["critic_loss"])
        flip_n = next_n = None
        if self.motion_sym_loss:
            flip_n, next_n = self._preproc_obs(d["flip_obs"]), self._preproc_obs(d["next_obses"])
        s_loss = None
        if self.motion_sym_loss and not stack_actor:         # (two more actor evaluations, on an arm of their own)
            if branch_streams is not None:
                flip_n.record_stream(s_s); next_n.record_stream(s_s)
            with on(s_s):
                s_loss = torch.mean(self._sym_loss(flip_n, next_n)["sym_loss"])
        # one evaluation of the actor on the stacked rows [policy | flipped | next] (the network is row-wise; amp_continuous.py evaluates
        # it three times): a third of the actor's launches, one weight-gradient product per layer instead of three and their sums
        if stack_actor:
            B = obs.shape[0]
            mu_all, logstd_all = net.eval_actor(torch.cat([obs, flip_n, next_n], dim=0))
            mu, logstd = mu_all[:B], logstd_all[:B]
            s_loss = torch.mean(self._sym_from_actions(mu_all[B:2 * B], mu_all[2 * B:])["sym_loss"])
        else:
            mu, logstd = net.eval_actor(obs)
        sigma = torch.exp(logstd)
        self._head_kl = None
        if heads and self.bounds_loss_coef is not None:
            # neglogp, surrogate, entropy, bound loss, clip fraction and the step's KL in one forward launch (learning/ppo_heads.py)
            a_loss, entropy, b_loss, clip_frac, self._head_kl = ppo_heads.actor_head(
                mu, logstd, d["actions"], d["old_logp_actions"], d["advantages"], self.e_clip, d.get("mu"), d.get("sigma"))
        else:
            action_log_probs = neglogp(d["actions"], mu, sigma, logstd)
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            a_info = self._actor_loss(d["old_logp_actions"], action_log_probs, d["advantages"], self.e_clip)
            a_loss = torch.mean(a_info["actor_loss"])
            b_loss, entropy = torch.mean(self.bound_loss(mu)), torch.mean(entropy)
            clip_frac = a_info["actor_clipped"].float().mean()
        if branch_streams is not None:
            # join: what crosses is a handful of scalars (allocated on the branch streams: tell the allocator who else reads them)
            for st, ts in ((s_c, [c_loss]), (s_d, list(disc_info.values())), (s_s, [s_loss] if (s_loss is not None and not stack_actor) else [])):
                main.wait_stream(st)
                for t in ts:
                    t.record_stream(main)
        info = {"actor_loss": a_loss.detach(), "critic_loss": c_loss.detach(), "b_loss": b_loss.detach(), "entropy": entropy.detach(),
                "actor_clip_frac": clip_frac.detach(), **{k: v.detach() for k, v in disc_info.items()}}
        if heads:
            # the weighted sum as ONE stack + dot (two launches forward, two backward) instead of a multiply and an add per term, each a
            # launch of its own in both directions
            terms = [a_loss, c_loss, entropy, b_loss, disc_info["disc_loss"]] + ([s_loss] if s_loss is not None else [])
            coef = [1.0, self.critic_coef, -self.entropy_coef, self.bounds_loss_coef, self._disc_coef] + ([self.sym_loss_coef] if s_loss is not None else [])
            loss = _weighted_sum(terms, coef)
        else:
            loss = a_loss + self.critic_coef * c_loss - self.entropy_coef * entropy + self.bounds_loss_coef * b_loss \
                + self._disc_coef * disc_info["disc_loss"]
            if s_loss is not None:
                loss = loss + s_loss * self.sym_loss_coef
        if s_loss is not None:
            info["sym_loss"] = s_loss.detach()
        return loss, info, mu.detach(), sigma.detach()

    def calc_gradients(self, d):
        self.set_train()
        loss, info, mu, sigma = self.compute_loss(d)
        self.bucket.zero()                                         # the .grad of every parameter aliases the flat bucket
        if self._gather_grads:
            self.bucket.release()                                  # (autograd keeps the gradients; one launch gathers them: dist.FlatGradBucket)
        loss.backward()
        if self._gather_grads:
            self.bucket.gather()
        self.bucket.all_reduce(average=True)                       # no-op on one rank
        self._clip_and_step()
        with torch.no_grad():
            info["kl"] = self._head_kl if self._head_kl is not None else policy_kl(mu, sigma, d["mu"], d["sigma"])
        info["loss"] = loss.detach()
        self.train_result = info
        return info

    def _clip_and_step(self):
        if self._flat_adam:
            self.optimizer.step(max_grad_norm=self.grad_norm if self.truncate_grads else 0.0)
            return
        if self.truncate_grads:
            nn.utils.clip_grad_norm_(self.a2c_network.parameters(), self.grad_norm)
        self.optimizer.step()

    # ------------------------------------------------------------------ the optimiser step as a HIP graph
    def _graph_body(self, part=None):
        """calc_gradients on the static minibatch buffers; every value the epoch averages is added to a static accumulator.
        part None: the whole step (one rank).  Data parallel the step is captured as TWO graphs around the gradient exchange
        (amp_continuous.py:440 `optimizer.synchronize()`): part "grad" = zero the bucket, losses, backward; part "apply" = clip, Adam,
        the epoch's accumulators -- between their replays `bucket.all_reduce` is issued on the same stream (RCCL: stream-ordered)."""
        d = self._g_in
        if part in (None, "grad"):
            masks = None
            if self._amp_dropout:
                masks = amp_dropout_expand(self._g_u, self.task._num_amp_obs_steps)
            self.bucket.zero()                                   # (ahead of the fork: the branches' backward passes write into it)
            if self._gather_grads:
                self.bucket.release()
            loss, info, mu, sigma = self.compute_loss(d, dropout_masks=masks, branch_streams=self._branch_streams())
            loss.backward()
            if self._gather_grads:
                self.bucket.gather()                             # (the engine has joined the arms' streams into this one by now)
            if part == "grad":
                self._g_mid = (loss.detach(), {k: v.detach() for k, v in info.items()}, mu.detach(), sigma.detach())   # static tensors of the graphs' pool
                return None
            loss = loss.detach()
        else:
            loss, info, mu, sigma = self._g_mid
            info = dict(info)
        self._clip_and_step()
        with torch.no_grad():
            info["kl"] = self._head_kl if self._head_kl is not None else policy_kl(mu, sigma, d["mu"], d["sigma"])
            info["loss"] = loss.detach()
            if self._g_acc is None:
                self._g_keys = sorted(info)
                self._g_acc = torch.zeros(len(self._g_keys), dtype=torch.float32, device=self.device)
            self._g_acc += torch.stack([info[k].float().reshape(()) for k in self._g_keys])
        return info

    def _branch_streams(self):
        """Streams of the critic / discriminator / symmetry-loss arms of the optimiser step, or None for one chain (`_g_arms`)."""
        if not getattr(self, "_g_arms", False):
            return None
        if getattr(self, "_g_branch", None) is None:
            self._g_branch = tuple(torch.cuda.Stream(device=self.device) for _ in range(3))
        return self._g_branch

    def _graph_fill(self, i):
        """Host side of a graphed step: gather minibatch i into the static buffers, draw the AMP dropout uniforms."""
        start, end = i * self.minibatch_size, (i + 1) * self.minibatch_size
        # the shuffled row ids live on the device for a whole pass (one upload per reshuffle): a per-step upload from pageable host
        # memory makes the host wait for the stream -- i.e. for the previous optimiser step -- before it can issue this one
        if getattr(self, "_idx_dev", None) is None:
            self._idx_dev = self._idx_buf.to(self.device)
        idx = self._idx_dev[start:end]
        if self._g_in is None:
            self._g_in = {k: torch.empty((self.minibatch_size,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device)
                          for k, v in self.dataset.items() if v is not None}
            self._g_u = torch.empty(19, self._amp_minibatch_size, 3, device=self.device)
        if self._fused_heads:                                # every fp32 table of the dataset in one gather launch (13 index_select launches otherwise)
            f32 = [k for k, buf in self._g_in.items() if buf.dtype == torch.float32 and self.dataset[k].is_contiguous()]
            ppo_heads.gather_rows(idx, [self.dataset[k] for k in f32], [self._g_in[k] for k in f32])
            rest = [k for k in self._g_in if k not in f32]
            if os.environ.get("EMLOCO_PPO_DEBUG") and not getattr(self, "_dbg_fill", False):
                self._dbg_fill = True
                print("graph_fill: fused gather", f32, "| index_select", rest, "|", {k: (tuple(v.shape), v.dtype, v.is_contiguous()) for k, v in self.dataset.items() if v is not None}, flush=True)
        else:
            rest = list(self._g_in)
        for k in rest:
            torch.index_select(self.dataset[k], 0, idx, out=self._g_in[k])
        if end >= self.batch_size:                           # (host draws in the eager step's order: reshuffle, then the dropout uniforms)
            self._idx_buf[:] = torch.randperm(self.batch_size)
            self._idx_dev = None
        if self._amp_dropout:
            self._g_u.copy_(amp_dropout_draw(self._amp_minibatch_size), non_blocking=True)

    # Whether the arms of the step run side by side or get in each other's way is decided by how the HIP runtime happens to map the
    # graph's internal streams onto its hardware queues (GPU_MAX_HW_QUEUES, the streams the process created earlier): measured 4.75 ms
    # with arms against 5.7 as one chain in one process, 9.4 against 6.0 in another (profiles/r04_ppo_hw_queues.txt).  So the step is
    # captured BOTH ways and each graph is timed on its first `_G_TRIALS` steps -- real optimiser steps, the two graphs compute the same
    # update -- and the faster one replays from then on.  EMLOCO_PPO_BRANCHES=0 / 1 pins one chain / the arms.
    _G_TRIALS = 3                                           # (x 4 candidates: one chain + three draws of the arms)

    def _capture(self, arms):
        """The step as a replayable object: one graph on one rank; data parallel a pair (gradient graph, apply graph) sharing one
        memory pool, replayed around the bucket's all-reduce (`_replay`)."""
        from ..dist import is_distributed
        self._g_arms = bool(arms)
        torch.cuda.synchronize(self.device)
        if not is_distributed():                            # (true for one rank with EMLOCO_FORCE_COLLECTIVES=1 as well: the exchange is issued)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_body()
            return g
        # (capture_error_mode "thread_local": the process group's watchdog thread polls the events of earlier collectives while this
        # thread captures -- under the default global mode its hipEventQuery fails the capture with "operation not permitted")
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self._graph_body("grad")
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            self._graph_body("apply")
        return (ga, gb)

    def _replay(self, g):
        if isinstance(g, tuple):
            g[0].replay()
            self.bucket.all_reduce(average=True)            # on the replays' stream: behind the gradient graph, ahead of the apply graph
            g[1].replay()
        else:
            g.replay()

    def _timed_replay(self, g):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._replay(g)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    def _graph_step(self, i):
        """One optimiser step: eager (on a side stream) for the first three calls, then captured (one chain and with arms), the two
        graphs timed on the next steps, then the faster one replayed."""
        self.set_train()
        self._graph_fill(i)
        from ..dist import is_distributed
        if self._graph is not None:
            self._replay(self._graph)
            return
        self._g_warm = getattr(self, "_g_warm", 0) + 1
        if self._g_warm <= 3:                               # warm-up off the default stream, as graph capture asks
            side = getattr(self, "_g_side", None) or torch.cuda.Stream(device=self.device)
            self._g_side = side
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                if not is_distributed():
                    self._graph_body()
                else:                                       # the data-parallel step, eagerly: gradients | exchange | apply
                    self._graph_body("grad")
                    self.bucket.all_reduce(average=True)
                    self._graph_body("apply")
            torch.cuda.current_stream(self.device).wait_stream(side)
            return
        mode = os.environ.get("EMLOCO_PPO_BRANCHES", "auto")
        if is_distributed() and mode == "auto":
            # every rank must replay the SAME number of collectives per step and the timing trial below is rank-local: data parallel the
            # choice is made by what the runtime was initialised with -- the arms when it has 16 hardware queues (what wins there), one
            # chain otherwise (4 queues: the arms serialise, 9.4 against 4.7 ms, profiles/r04_ppo_hw_queues.txt) -- which comes from the
            # environment / the entry point and is the same on every rank
            from .. import hw_queues
            mode = "1" if (hw_queues() or 4) >= 16 else "0"
        if mode in ("0", "1"):
            self._graph = self._capture(mode == "1")        # (capture does not execute: run the step that was just captured)
            self._replay(self._graph)
            return
        if getattr(self, "_g_cand", None) is None:
            # (round 6) the arms are captured up to EMLOCO_PPO_ARMS_TRIES times (default 3): every instantiation draws fresh internal
            # streams, i.e. another mapping of the arms onto the hardware queues -- one unlucky draw (4.5 ms where the lucky one runs 3.2,
            # profiles/r06_bench_default.log's graph_trial_ms) no longer decides the epoch; a stream created in between moves the draw on
            tries = max(1, int(os.environ.get("EMLOCO_PPO_ARMS_TRIES", "3")))
            self._g_cand = [[self._capture(False), [], False]]
            self._g_spare_streams = []
            for _ in range(tries):
                self._g_cand.append([self._capture(True), [], True])
                self._g_spare_streams.append(torch.cuda.Stream(device=self.device))
        # Each candidate takes `_G_TRIALS` CONSECUTIVE real steps, timed as one stretch (one event ahead of the first replay, one behind
        # the last, no synchronisation in between): what is compared is the sustained rate with the host's per-step work between the
        # replays, as the epoch runs -- a replay timed on its own flattered the arms (3.87 ms alone, 4.74 ms per step sustained, against
        # 4.71 / 4.73 for the chain: profiles/r05_bench_default.log).
        for cand in self._g_cand:
            if len(cand[1]) < self._G_TRIALS:
                if not cand[1]:
                    cand.append(torch.cuda.Event(enable_timing=True))
                    cand[3].record()
                self._replay(cand[0])
                cand[1].append(0.0)
                if len(cand[1]) == self._G_TRIALS:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    e1.synchronize()
                    cand[1] = [cand[3].elapsed_time(e1) / self._G_TRIALS] * self._G_TRIALS
                break
        if all(len(c[1]) >= self._G_TRIALS and c[1][-1] > 0.0 for c in self._g_cand):
            best = min(self._g_cand, key=lambda c: c[1][0])
            self._graph, self._g_arms = best[0], best[2]
            self._g_trial_ms = {"one chain": round(min(c[1][0] for c in self._g_cand if not c[2]), 3),
                                "arms": round(min(c[1][0] for c in self._g_cand if c[2]), 3),
                                "arms_draws": [round(c[1][0], 3) for c in self._g_cand if c[2]]}
            self._g_cand = None

    # ------------------------------------------------------------------ epoch
    def prepare_dataset(self, batch):
        advantages = torch.sum(batch["returns"] - batch["values"], axis=1)
        if self.normalize_advantage:
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
        values, returns = batch["values"], batch["returns"]
        if self.normalize_value:
            values = self.value_mean_std(values)
            returns = self.value_mean_std(returns)
        self.dataset = {"old_values": values, "old_logp_actions": batch["neglogpacs"], "advantages": advantages, "returns": returns,
                        "actions": batch["actions"], "obs": batch["obses"], "mu": batch["mus"], "sigma": batch["sigmas"],
                        "amp_obs": batch["amp_obs"], "amp_obs_demo": batch["amp_obs_demo"], "amp_obs_replay": batch["amp_obs_replay"]}
        if self.motion_sym_loss:
            self.dataset["flip_obs"], self.dataset["next_obses"] = batch["flip_obs"], batch["next_obses"]

    def _minibatch(self, i):
        """AMPDataset._get_item (amp_datasets.py:16-33): shuffled index buffer, reshuffled after the last minibatch."""
        start, end = i * self.minibatch_size, (i + 1) * self.minibatch_size
        idx = self._idx_buf[start:end].to(self.device)
        d = {k: v[idx] for k, v in self.dataset.items() if v is not None}
        if end >= self.batch_size:
            self._idx_buf[:] = torch.randperm(self.batch_size)
            self._idx_dev = None
        return d

    def _store_replay_amp_obs(self, amp_obs):
        if self._amp_replay_buffer.get_total_count() > self._amp_replay_buffer.get_buffer_size():
            keep = torch.bernoulli(torch.full((amp_obs.shape[0],), self._amp_replay_keep_prob, device=self.device)) == 1.0
            amp_obs = amp_obs[keep]
        if amp_obs.shape[0] > 0:
            self._amp_replay_buffer.store({"amp_obs": amp_obs[:self._amp_replay_buffer.get_buffer_size()]})

    def train_epoch(self):
        t0 = time.time()
        batch = self.play_steps()
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        self._amp_obs_demo_buffer.store({"amp_obs": self._fetch_amp_obs_demo(self._amp_batch_size)})
        n = batch["amp_obs"].shape[0]
        batch["amp_obs_demo"] = self._amp_obs_demo_buffer.sample(n)["amp_obs"]
        batch["amp_obs_replay"] = batch["amp_obs"] if self._amp_replay_buffer.get_total_count() == 0 \
            else self._amp_replay_buffer.sample(n)["amp_obs"]
        self.set_train()
        self.prepare_dataset(batch)
        infos = []
        from ..dist import world_size
        # (round 5: the graphed step also runs data parallel -- two captured segments around the bucket's all-reduce, `_capture`)
        graphed = self.use_graph
        n_steps = self.mini_epochs_num * (self.batch_size // self.minibatch_size)
        if graphed and self._g_acc is not None:
            self._g_acc.zero_()
        for _ in range(self.mini_epochs_num):
            for i in range(self.batch_size // self.minibatch_size):
                if graphed:
                    self._graph_step(i)
                else:
                    infos.append(self.calc_gradients(self._minibatch(i)))
        self._store_replay_amp_obs(batch["amp_obs"])
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        self.epoch_num += 1
        self.frame += self.batch_size
        if graphed:
            vals = (self._g_acc / n_steps).tolist()
            out = dict(zip(self._g_keys, vals))
            self.train_result = out
        else:
            out = {k: torch.stack([i[k].float() for i in infos]).mean().item() for k in infos[0]}
        # the reference's per-epoch exchanges: hvd.average_value of the KL (amp_continuous.py:287-288) and hvd.sync_stats of
        # the running statistics (common_agent.py:179-180); no-ops on one rank
        out["kl"] = all_reduce_mean_scalar(out["kl"])
        sync_running_mean_std(self.running_mean_std, self.value_mean_std, self._amp_input_mean_std)
        out.update(play_time=t1 - t0, update_time=t2 - t1, total_time=t2 - t0, fps_step=self.batch_size / (t1 - t0),
                   fps_total=self.batch_size / (t2 - t0), reward_raw=batch["reward_raw"].tolist())
        return out

    # ------------------------------------------------------------------ checkpoints (common_agent.py:252-264 layout)
    def get_full_state_weights(self):
        return {"model": {"a2c_network." + k: v for k, v in self.a2c_network.state_dict().items()},
                "running_mean_std": self.running_mean_std.state_dict(), "reward_mean_std": self.value_mean_std.state_dict(),
                "amp_input_mean_std": self._amp_input_mean_std.state_dict(), "optimizer": self.optimizer.state_dict(),
                "epoch": self.epoch_num, "frame": self.frame}

    def save(self, fn):
        torch.save(self.get_full_state_weights(), fn if fn.endswith(".pth") else fn + ".pth")

    def restore(self, fn):
        ck = torch.load(fn, map_location=self.device)
        self.a2c_network.load_state_dict({k[len("a2c_network."):]: v for k, v in ck["model"].items()})
        self.running_mean_std.load_state_dict(ck["running_mean_std"])
        if "reward_mean_std" in ck:
            self.value_mean_std.load_state_dict(ck["reward_mean_std"])
        if "amp_input_mean_std" in ck:
            self._amp_input_mean_std.load_state_dict(ck["amp_input_mean_std"])
        if "optimizer" in ck:
            self.optimizer.load_state_dict(ck["optimizer"])
        self.epoch_num, self.frame = ck.get("epoch", 0), ck.get("frame", 0)
