"""HumanoidTraj: trajectory-following task pieces (mirror of pacer/pacer/env/tasks/humanoid_traj.py)."""
import torch

from ...env.util import traj_generator
from ...utils.flags import flags
from .humanoid_amp_task import HumanoidAMPTask


class HumanoidTraj(HumanoidAMPTask):
    def __init__(self, cfg, sim_params, physics_engine, device_type, device_id, headless):
        self._num_traj_samples = cfg["env"]["numTrajSamples"]
        self._traj_sample_timestep = cfg["env"]["trajSampleTimestep"]
        self._speed_min = cfg["env"]["speedMin"]
        self._speed_max = cfg["env"]["speedMax"]
        self._accel_max = cfg["env"]["accelMax"]
        self._sharp_turn_prob = cfg["env"]["sharpTurnProb"]
        self._fail_dist = 4.0
        self.step_to_pred = cfg["env"]["stepToPred"]
        self.inverted = None
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type,
                         device_id=device_id, headless=headless)
        self._build_traj_generator()
        return

    def get_task_obs_size(self):
        return 2 * self._num_traj_samples if self._enable_task_obs else 0

    def _build_traj_generator(self):                              # humanoid_traj.py:110-129
        episode_dur = self.max_episode_length * self.dt
        self._traj_gen = traj_generator.TrajGenerator(self.num_envs, episode_dur, 101, self.device, 2.0,
                                                      self._speed_min, self._speed_max, self._accel_max,
                                                      self._sharp_turn_prob, self._motion_lib,
                                                      hybridInitProb=self._hybrid_init_prob, flags=flags,
                                                      traj_data=self.cfg["env"].get("traj_data", None))
        env_ids = torch.arange(self.num_envs, device=self.device, dtype=torch.long)
        self._traj_gen.reset(env_ids, self._humanoid_root_states[:, 0:3], self._humanoid_root_states[:, 7:10],
                             self._sampled_motion_ids[env_ids], self._motion_start_times[env_ids])
        self.inverted = self._traj_gen.show_inverted()
        return

    def _fetch_traj_samples(self, env_ids=None):                  # humanoid_traj.py:208-224 (host version)
        if (env_ids is None):
            env_ids = torch.arange(self.num_envs, device=self.device, dtype=torch.long)
        timestep_beg = self.progress_buf[env_ids] * self.dt
        timesteps = torch.arange(self._num_traj_samples, device=self.device, dtype=torch.float) * self._traj_sample_timestep
        traj_timesteps = timestep_beg.unsqueeze(-1) + timesteps
        env_ids_tiled = torch.broadcast_to(env_ids.unsqueeze(-1), traj_timesteps.shape)
        flat = self._traj_gen.calc_pos(env_ids_tiled.flatten(), traj_timesteps.flatten())
        return torch.reshape(flat, shape=(env_ids.shape[0], self._num_traj_samples, flat.shape[-1]))
