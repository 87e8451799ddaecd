"""TEST INFRASTRUCTURE -- torch restatement of the LocoVal rollout bookkeeping and fit.

Only tests/ may import this (and bench.py's cpu_baseline legs): the product path (emloco_amd/learning/locoval_rollout.py) runs
the same arithmetic as HIP kernels (`locoval_returns_kernel`, `locoval_fit_grad_kernel`, `adamw_gated_kernel`) and fails loudly
without them.  Restates pacer/pacer/learning/amp_continuous_value.py:93-118 (return accumulation: unweighted task +
discriminator reward, per-env gamma^t, the `step_to_pred` cut-off masks) and :122-145 (the fit on the rows whose emitted sum
is non-zero, target (G - min) / (max - min), common_agent.py:154-155).  PINNED by tests/golden/locoval_returns.npz, which
tests/golden/gen_golden_a17.py produced by running the reference's own `AMPValueAgent.play_steps` on scripted inputs
(tests/test_locoval_rollout_cpu.py::test_return_accumulator_matches_the_reference_play_steps).
"""
import torch

from emloco_amd.learning.locoval_rollout import LocoValRollout


class ReturnAccumulator:
    """Per-env discounted return of amp_continuous_value.py:93-118, as mask arithmetic on any device.

    update() takes one step's (E,) task rewards (inversion penalty already applied), discriminator rewards and done flags and
    returns what the reference adds to `game_combined_rewards` on that step: the discounted sum of an episode at the step it
    ends if that is within `step_to_pred` control steps, or at step `step_to_pred` if it runs longer (later rewards of the
    episode are accumulated but never emitted), zero elsewhere."""

    def __init__(self, num_envs, step_to_pred, gamma, device):
        z = lambda: torch.zeros(num_envs, device=device)
        self.step_to_pred, self.gamma = step_to_pred, gamma
        self.current_rewards, self.current_lengths, self.current_combined_rewards = z(), z(), z()
        self.discount_coefs = torch.ones(num_envs, device=device)

    def update(self, rewards, amp_rewards, dones):
        dones_b = dones.bool()
        not_dones = 1.0 - dones_b.float()
        self.current_rewards += rewards
        self.current_lengths += 1
        combined = rewards + amp_rewards                                                    # :96, unweighted
        self.current_combined_rewards += combined * self.discount_coefs
        done_early = torch.logical_and(self.current_lengths <= self.step_to_pred, dones_b)
        over_pred = torch.logical_and(self.current_lengths == self.step_to_pred, ~dones_b)
        emitted = self.current_combined_rewards * (done_early | over_pred).float()
        self.current_combined_rewards = self.current_combined_rewards * not_dones
        self.discount_coefs = torch.where(dones_b, torch.ones_like(self.discount_coefs), self.discount_coefs * self.gamma)
        self.current_rewards = self.current_rewards * not_dones
        self.current_lengths = self.current_lengths * not_dones
        return emitted


class TorchLocoValRollout(LocoValRollout):
    """LocoValRollout with the per-step bookkeeping and fit in torch instead of the HIP kernels: the comparison partner of the
    fused path on a GPU, and what the world_size-2 gloo tests drive on the CPU (same exchange code: one flat all-reduce per step,
    the AdamW commit gated on the global episode count)."""

    def __init__(self, *a, **k):
        k["fused"] = False
        super().__init__(*a, **k)

    def _make_return_state(self, E, step_to_pred, gamma, device):
        return ReturnAccumulator(E, step_to_pred, gamma, device)

    def _bookkeeping(self, rewards, amp_rewards, dones, inverted):
        with torch.no_grad():
            rewards = torch.where(inverted, rewards * (-self.inversion_penalty_scale), rewards)      # :63-64
            if amp_rewards is None:
                amp_rewards = torch.zeros(self.num_actors, device=self.device)
            self.game_combined_rewards += self.acc.update(rewards, amp_rewards, dones)
        with torch.enable_grad():
            self._fit()

    def _fit(self):
        """:122-145 as a masked sum over all envs: identical to indexing the finished episodes, without reading their ids."""
        env = self.env
        valid = self.game_combined_rewards != 0
        init_pose = env.get_init_pose().to(self.device)
        waypoint_traj = env.get_waypoint_traj()[:, :13, :].contiguous().to(self.device)
        init_vel = env.get_init_vel().to(self.device)
        pred = self.valuenet(waypoint_traj, init_pose, init_vel).reshape(-1)
        target = (self.game_combined_rewards - self.min_cum_rewards) / (self.max_cum_rewards - self.min_cum_rewards)
        w = valid.float()
        self.bucket.zero()
        loss = (w * (pred - target) ** 2).sum()                                             # MSELoss(reduction='sum') on the valid rows
        loss.backward()
        with torch.no_grad():
            self.bucket.tail[0] = loss.detach()
            self.bucket.tail[1] = w.sum()
            self.bucket.all_reduce(average=False)                                           # unconditional: one collective per step
            tail = self.bucket.tail.double()
            gate = tail[1] > 0.5
            self.vnet_optimizer.step(gate)                                                  # committed on the device iff an episode finished
            g = gate.double()
            self._stats[0:2] = torch.where(gate, tail, self._stats[0:2])
            self._stats[2:4] += tail * g
            self._stats[4] += g
            self.game_combined_rewards = torch.zeros_like(self.game_combined_rewards)
