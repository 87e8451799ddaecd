"""Frozen PACER policy + AMP discriminator reward for the LocoVal rollout (config 3).

Assembles what the reference's AMPValueAgent holds at test / LocoVal-collection time: `a2c_network`
(amp_network_sept_builder.py), `running_mean_std` over the 1422 observations (common_agent.py:57-60),
`_amp_input_mean_std` over the 15x206 AMP observations (amp_continuous.py:91-93), and the two functions the rollout
loop calls each step: the action (`FrozenPolicy.act`) and the style reward
`-log(max(1 - sigmoid(D(norm(amp_obs))), 1e-4)) * disc_reward_scale` (amp_continuous.py:675-692).
Checkpoints in the rl_games layout ({'model': {'a2c_network.*'}, 'running_mean_std', 'amp_input_mean_std'};
common_agent.py:252-264, amp_players.py:21-31) load directly; without one the networks keep their initialisation.
"""
import os

import torch
import yaml

from ..utils.running_mean_std import RunningMeanStd
from .amp_network_sept_builder import AMPSeptBuilder
from .policy_runner import FrozenDisc, FrozenPolicy

DEFAULT_CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "cfg", "train", "rlg",
                           "amp_humanoid_smpl_sept_task.yaml")


class AMPPolicyBundle:
    def __init__(self, task, cfg_train=None, checkpoint=None, deterministic=False, seed=0):
        if cfg_train is None:
            cfg_train = yaml.safe_load(open(DEFAULT_CFG))
        params = cfg_train["params"]
        self.config = params.get("config", {})
        self.device = torch.device(task.device)
        self.num_envs = task.num_envs
        obs_size = task.get_obs_size()
        self_size = task.get_self_obs_size()
        task_size = obs_size - self_size
        amp_size = task.get_num_amp_obs()
        self.running_mean_std = RunningMeanStd((obs_size,)).to(self.device).eval()
        self.amp_input_mean_std = RunningMeanStd((amp_size,)).to(self.device).eval()
        builder = AMPSeptBuilder()
        builder.load(params["network"])
        self.a2c_network = builder.build(
            "amp", actions_num=task.num_actions, input_shape=(obs_size,), num_seqs=1, value_size=1, amp_input_shape=(amp_size,),
            self_obs_size=self_size, task_obs_size=task_size, task_obs_size_detail=task.get_task_obs_size_detail(),
            mean_std=self.running_mean_std).to(self.device).eval()
        if checkpoint:
            ck = torch.load(checkpoint, map_location="cpu")
            sd = {k[len("a2c_network."):]: v for k, v in ck["model"].items() if k.startswith("a2c_network.")}
            self.a2c_network.load_state_dict(sd, strict=True)
            if "running_mean_std" in ck:
                self.running_mean_std.load_state_dict(ck["running_mean_std"])
            if "amp_input_mean_std" in ck:
                self.amp_input_mean_std.load_state_dict(ck["amp_input_mean_std"])
        self.frozen = FrozenPolicy(self.a2c_network, self.running_mean_std, self.num_envs, self.device,
                                   clip_actions=float(self.config.get("clip_actions", 1.0)))
        self.deterministic = deterministic
        self.disc_reward_scale = float(self.config.get("disc_reward_scale", 2.0))
        self.frozen_disc = FrozenDisc(self.a2c_network, self.amp_input_mean_std, self.num_envs, self.device,
                                      disc_reward_scale=self.disc_reward_scale, normalize=bool(self.config.get("normalize_amp_input", True)))
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(seed)

    def policy(self, obs):
        return self.frozen.act(obs, deterministic=self.deterministic, generator=self.generator)

    def disc_reward(self, amp_obs):
        """The style reward of a step (amp_continuous.py:675-692) through the packed runner; `disc_reward_modules` is the same through
        the network's modules (what the tests compare it with)."""
        with torch.no_grad():
            return self.frozen_disc.reward(amp_obs)

    def disc_stage(self, amp_obs):
        """First half of `disc_reward` (FrozenDisc.stage): one launch that reads the step's AMP observations."""
        with torch.no_grad():
            self.frozen_disc.stage(amp_obs)

    def disc_stage_ring(self, n):
        """Size the staged operand's ring: the number of `disc_stage` calls that may be issued before the oldest `disc_reward_staged`
        has run (FrozenDisc.set_stage_ring)."""
        self.frozen_disc.set_stage_ring(n)

    def disc_reward_staged(self):
        """Second half (FrozenDisc.reward_staged): the GEMMs and the scalar transform, nothing of the task's is read."""
        with torch.no_grad():
            return self.frozen_disc.reward_staged()

    def disc_reward_modules(self, amp_obs):
        with torch.no_grad():
            x = amp_obs.reshape(amp_obs.shape[0], -1)
            if self.config.get("normalize_amp_input", True):
                x = self.amp_input_mean_std(x)
            logits = self.a2c_network.eval_disc(x)
            prob = 1 / (1 + torch.exp(-logits))
            disc_r = -torch.log(torch.maximum(1 - prob, torch.tensor(0.0001, device=self.device)))
            return (disc_r * self.disc_reward_scale).squeeze(-1)

    def eval_critic(self, obs):
        with torch.no_grad():
            return self.a2c_network.eval_critic(self.running_mean_std(obs))
