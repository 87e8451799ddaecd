"""HumanoidAMPTask: task-observation hooks (mirror of pacer/pacer/env/tasks/humanoid_amp_task.py)."""
from .humanoid_amp import HumanoidAMP


class HumanoidAMPTask(HumanoidAMP):
    def __init__(self, cfg, sim_params, physics_engine, device_type, device_id, headless):
        self._enable_task_obs = cfg["env"]["enableTaskObs"]
        self.has_task = True
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type,
                         device_id=device_id, headless=headless)
        return

    def get_obs_size(self):
        obs_size = super().get_obs_size()
        if (self._enable_task_obs):
            obs_size += self.get_task_obs_size()
        return obs_size

    def get_task_obs_size(self):
        return 0

    def pre_physics_step(self, actions):
        super().pre_physics_step(actions)
        self._update_task()
        return

    def _update_task(self):
        return

    def _reset_envs(self, env_ids):
        super()._reset_envs(env_ids)
        self._reset_task(env_ids)
        return

    def _reset_task(self, env_ids):
        return
