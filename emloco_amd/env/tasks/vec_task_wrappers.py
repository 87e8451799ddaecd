"""VecTaskPython / VecTaskPythonWrapper: the (obs, reward, done, info) adapter the agents call.

Mirror of pacer/pacer/env/tasks/vec_task.py:121-142 and vec_task_wrappers.py:28-73, including the EmLoco
getters `get_waypoint_traj / get_init_pose / get_init_vel` (:50-66) that feed the LocoVal fit.
"""
import numpy as np
import torch


class VecTaskPython():
    def __init__(self, task, rl_device, clip_observations=5.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_agents = 1
        self.num_observations = task.num_obs
        self.num_states = task.num_states
        self.num_actions = task.num_actions
        self.clip_obs = clip_observations
        self.rl_device = rl_device

    def get_number_of_agents(self):
        return self.num_agents

    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_obs(self):
        return self.num_observations

    def get_state(self):
        return torch.clamp(self.task.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def _obs(self):
        # clip_observations is read from the wrong dict level in the reference => inf => no clipping (parse_task.py:45)
        if np.isinf(self.clip_obs):
            o = self.task.obs_buf         # the live buffer: a deferred / side-stream observation pass (task.fused_chain, overlap_obs) fills it later
        else:
            if hasattr(self.task, "wait_obs"):
                self.task.wait_obs()      # a clipped COPY must see the finished rows
            o = torch.clamp(self.task.obs_buf, -self.clip_obs, self.clip_obs)
        return o.to(self.rl_device)

    def step(self, actions):
        self.task.step(actions)
        return self._obs(), self.task.rew_buf.to(self.rl_device), self.task.reset_buf.to(self.rl_device), self.task.extras

    def reset(self):
        actions = 0.01 * (1 - 2 * torch.rand([self.task.num_envs, self.task.num_actions], dtype=torch.float32, device=self.rl_device))
        self.task.step(actions)
        return self._obs()


class VecTaskPythonWrapper(VecTaskPython):
    def reset(self, env_ids=None):
        self.task.reset(env_ids)
        return self._obs()

    def reset_done(self):
        """sync-free `reset(dones.nonzero())` (extension; see HumanoidPedestrianTerrain.reset_done)"""
        self.task.reset_done()
        return self._obs()

    def raw_reward(self):
        return self.task.reward_raw.to(self.rl_device)

    def get_waypoint_traj(self):
        waypoint_traj = self.task.waypoint_traj.clone()
        waypoint_traj -= waypoint_traj[:, 0].clone().unsqueeze(1)
        return waypoint_traj

    def get_init_pose(self):
        init_pose = self.task.init_pose.clone()
        init_pose -= init_pose[:, 0].clone().unsqueeze(1)
        return init_pose

    def get_init_vel(self):
        return self.task.init_vel.clone()

    def fetch_amp_obs_demo(self, num_samples):
        return self.task.fetch_amp_obs_demo(num_samples)
