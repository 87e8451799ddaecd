"""LocoVal training rollout: the bookkeeping of AMPValueAgent.play_steps without rl_games.

Mirror of pacer/pacer/learning/amp_continuous_value.py:34-178 (play_steps) and common_agent.py:89-97,154-155
(LocoVal optimiser / target normalisation): per step reset finished envs, act, step, apply the inversion penalty,
accumulate the discounted combined reward per env up to `step_to_pred` control steps, and when episodes finish fit
LocoVal by sum-MSE on (waypoints, initial pose, initial velocity) -> normalised return with AdamW(1e-3, wd 1e-4).
The policy is pluggable (`policy(obs) -> actions`; default: the frozen policy's exploration noise N(0, e^-2.9)),
so is the AMP discriminator reward (`disc_reward(amp_obs) -> (E,)`, default 0): both networks belong to
rl_games-side code that is listed under 'next' (SURVEY.md 8f.1).  Multi-GPU: gradients are all-reduced as one
flat 6 174-float bucket and divided by the global number of fitted episodes (sum-reduction semantics).
"""
import math

import torch

from ..dist import FlatGradBucket, all_reduce_sum_count
from .value_pose_net import ValuePoseNet


class LocoValRollout:
    def __init__(self, vec_env, use_pose=True, use_vel=True, horizon_length=32, gamma=0.99, inversion_penalty_scale=1.0,
                 task_reward_w=0.5, disc_reward_w=0.5, policy=None, disc_reward=None, min_cum_rewards=-10.0,
                 max_cum_rewards=100.0, lr=1e-3, weight_decay=1e-4):
        self.vec_env = vec_env
        env = vec_env.env if hasattr(vec_env, "env") else vec_env
        self.env = env
        self.task = env.task
        self.device = torch.device(self.task.device)
        self.num_actors = self.task.num_envs
        self.horizon_length, self.gamma = horizon_length, gamma
        self.inversion_penalty_scale = inversion_penalty_scale
        self.task_reward_w, self.disc_reward_w = task_reward_w, disc_reward_w
        self.step_to_pred = self.task.step_to_pred
        self.policy = policy or (lambda obs: torch.randn(self.num_actors, self.task.num_actions, device=self.device) * math.exp(-2.9))
        self.disc_reward = disc_reward or (lambda amp_obs: torch.zeros(self.num_actors, device=self.device))
        self.min_cum_rewards, self.max_cum_rewards = min_cum_rewards, max_cum_rewards     # common_agent.py:154-155
        self.valuenet = ValuePoseNet(use_pose=use_pose, use_vel=use_vel).to(self.device)
        self.vnet_optimizer = torch.optim.AdamW(self.valuenet.parameters(), lr=lr, weight_decay=weight_decay)
        self.bucket = FlatGradBucket(self.valuenet.parameters())
        E = self.num_actors
        z = lambda: torch.zeros(E, device=self.device)
        self.current_rewards, self.current_lengths, self.current_combined_rewards = z(), z(), z()
        self.game_combined_rewards = z()
        self.discount_coefs = torch.ones(E, device=self.device)
        self.done_indices = torch.arange(E, device=self.device)     # first call resets everything
        self.vnet_loss, self.vnet_fits, self.frames = 0.0, 0, 0

    def play_steps(self):
        env, task = self.env, self.task
        for n in range(self.horizon_length):
            with torch.no_grad():
                if self.done_indices.numel():
                    env.reset(self.done_indices)
                obs = task.obs_buf
                actions = self.policy(obs)
                obs, rewards, dones, infos = self.vec_env.step(actions)
                rewards = rewards.clone()
                inverted = task.inverted
                rewards[inverted] *= (-self.inversion_penalty_scale)                       # :63-64
                amp_rewards = self.disc_reward(infos["amp_obs"])
                self.current_rewards += rewards
                self.current_lengths += 1
                combined = self.task_reward_w * rewards + self.disc_reward_w * amp_rewards
                self.current_combined_rewards += combined * self.discount_coefs
                self.done_indices = dones.nonzero(as_tuple=False).flatten()
                not_dones = 1.0 - dones.float()
                done_early = torch.logical_and(self.current_lengths <= self.step_to_pred, dones.bool())
                over_pred = torch.logical_and(self.current_lengths == self.step_to_pred, not_dones.bool())
                self.game_combined_rewards += self.current_combined_rewards * (done_early | over_pred).float()
                self.current_combined_rewards = self.current_combined_rewards * not_dones
                self.discount_coefs = self.discount_coefs * self.gamma
                self.discount_coefs[self.done_indices] = 1.0
                self.current_rewards = self.current_rewards * not_dones
                self.current_lengths = self.current_lengths * not_dones
                self.frames += self.num_actors
            valid = torch.nonzero(self.game_combined_rewards, as_tuple=True)[0]
            if valid.numel() > 0:                                                              # :122-145
                init_pose = env.get_init_pose().to(self.device)
                waypoint_traj = env.get_waypoint_traj()[:, :13, :].contiguous().to(self.device)
                init_vel = env.get_init_vel().to(self.device)
                pred = self.valuenet(waypoint_traj, init_pose, init_vel).squeeze(-1)
                target = (self.game_combined_rewards[valid] - self.min_cum_rewards) / (self.max_cum_rewards - self.min_cum_rewards)
                self.bucket.zero()
                loss = torch.nn.functional.mse_loss(pred[valid], target, reduction="sum")
                loss.backward()
                self.bucket.all_reduce(average=False)
                gl, gc = all_reduce_sum_count(loss, valid.numel())
                self.vnet_optimizer.step()
                self.vnet_loss = float(gl) / max(float(gc), 1.0)
                self.vnet_fits += 1
                self.game_combined_rewards = torch.zeros_like(self.game_combined_rewards)
        return self.vnet_loss

    # ------------------------------------------------------------------ checkpoints (common_agent.py:248-264, finetune branch)
    def save(self, model_output_file, epoch_num=None):
        """`<file>_valuenet.pth`, or `<file>_valuenet_<epoch:08d>.pth` for the intermediate checkpoints: a plain state_dict
        with the reference's keys (`_network.fc{1,2,3}.{weight,bias}`), loadable by train_jta.py / evaluate_jta.py."""
        path = model_output_file + ("_valuenet.pth" if epoch_num is None else "_valuenet_" + str(epoch_num).zfill(8) + ".pth")
        torch.save({k: v.detach().cpu() for k, v in self.valuenet.state_dict().items()}, path)
        return path

    def restore(self, path):
        self.valuenet.load_state_dict(torch.load(path, map_location=self.device))

    def train(self, max_epochs, model_output_file=None, save_freq=200, save_intermediate=True):
        """The finetune loop of CommonAgent.train: one epoch = one play_steps horizon."""
        for epoch_num in range(1, max_epochs + 1):
            self.play_steps()
            if model_output_file and save_freq > 0 and epoch_num % save_freq == 0:
                self.save(model_output_file)
                if save_intermediate and epoch_num % (save_freq * 5) == 0:
                    self.save(model_output_file, epoch_num)
        if model_output_file:
            self.save(model_output_file)
        return self.vnet_loss

