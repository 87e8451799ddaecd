"""BaseTask: buffers, sim creation and the env.step skeleton.

Mirror of pacer/pacer/env/tasks/base_task.py (class BaseTask :47-265, `_physics_step` :792-797) reduced to the
headless hot path: viewer, video recording, websocket client and domain randomisation are out of scope.
"""
import torch

from ...gym import gymapi
from ...utils.flags import flags


class BaseTask():
    def __init__(self, cfg, enable_camera_sensors=False):
        self.headless = cfg["headless"]
        self.gym = gymapi.acquire_gym()
        self.paused = False
        self.device_type = cfg.get("device_type", "cuda")
        self.device_id = cfg.get("device_id", 0)
        self.device = "cpu"
        if self.device_type == "cuda" or self.device_type == "GPU":
            self.device = "cuda" + ":" + str(self.device_id)
        self.graphics_device_id = -1
        self.num_envs = cfg["env"]["numEnvs"]
        self.num_obs = cfg["env"]["numObservations"]
        self.num_states = cfg["env"].get("numStates", 0)
        self.num_actions = cfg["env"]["numActions"]
        self.control_freq_inv = cfg["env"].get("controlFrequencyInv", 1)

        # allocate buffers (base_task.py:96-111); int64 progress/reset, reset_buf starts at 1
        self.obs_buf = torch.zeros((self.num_envs, self.num_obs), device=self.device, dtype=torch.float)
        self.states_buf = torch.zeros((self.num_envs, self.num_states), device=self.device, dtype=torch.float)
        self.rew_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.float)
        self.reset_buf = torch.ones(self.num_envs, device=self.device, dtype=torch.long)
        self.progress_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        self.randomize_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        self.extras = {}

        self.create_sim()
        if not self.gym.prepare_sim(self.sim):
            raise RuntimeError("*** Failed to prepare sim")
        self.enable_viewer_sync = True
        self.viewer = None

    def set_sim_params_up_axis(self, sim_params, axis):
        if axis == 'z':
            sim_params.up_axis = gymapi.UP_AXIS_Z
            sim_params.gravity.x, sim_params.gravity.y, sim_params.gravity.z = 0, 0, -9.81
            return 2
        return 1

    def create_sim(self, compute_device, graphics_device, physics_engine, sim_params):
        sim = self.gym.create_sim(compute_device, graphics_device, physics_engine, sim_params)
        if sim is None:
            raise RuntimeError("*** Failed to create sim")   # the reference prints and quit()s (base_task.py:238-241)
        return sim

    def step(self, actions):                                   # base_task.py:245-265
        self.pre_physics_step(actions)
        self._physics_step()
        cb = getattr(self, "after_physics_launch", None)       # (a rollout loop's hook: host work to issue WHILE the rigid-body launch
        if cb is not None:                                      # runs -- side-stream launches that want to start under it)
            cb()
        self.gym.fetch_results(self.sim, True)
        self.post_physics_step()

    def get_states(self):
        return self.states_buf

    def render(self, sync_frame_time=False):
        return

    def pre_physics_step(self, actions):
        raise NotImplementedError

    def _physics_step(self):                                   # base_task.py:792-797, fused into one launch
        if not self.paused and self.enable_viewer_sync:
            self.gym.simulate_n(self.sim, self.control_freq_inv)
        return

    def post_physics_step(self):
        raise NotImplementedError
