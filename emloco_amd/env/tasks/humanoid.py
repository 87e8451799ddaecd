"""Humanoid: SMPL humanoid envs, PD action map, fused post-physics step.

Mirror of pacer/pacer/env/tasks/humanoid.py (class Humanoid :51-1320) for the configuration the BASELINE
configs use (smpl_humanoid asset, PD control, max-coordinate observations, motion_sym_loss).  Per-step
arithmetic (observations, reward, reset masks, AMP rows) runs in ONE fused HIP launch
(emloco_task_post_physics); the `_compute_*` methods below launch the same kernel with the matching mode
bits so the reference's call structure keeps working on reset paths.
"""
import numpy as np
import torch

from ... import _lib as L
from ...gym import gymapi, gymtorch
from ...gym.torch_utils import get_axis_params, to_torch
from ...model import smpl_humanoid
from ...post_physics import BODY_NAMES, LEFT_TO_RIGHT, PostPhysics, dof_subset_indices
from ...utils.flags import flags
from .base_task import BaseTask

ENABLE_MAX_COORD_OBS = True


class Humanoid(BaseTask):
    def __init__(self, cfg, sim_params, physics_engine, device_type, device_id, headless):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.has_task = getattr(self, "has_task", False)
        self.device_type, self.device_id = device_type, device_id
        self.device = "cpu"
        if device_type == "cuda" or device_type == "GPU":
            self.device = "cuda" + ":" + str(device_id)
        self.load_smpl_configs(cfg)
        self._pd_control = self.cfg["env"]["pdControl"]
        self.power_scale = self.cfg["env"]["powerScale"]
        self.plane_static_friction = self.cfg["env"]["plane"]["staticFriction"]
        self.plane_dynamic_friction = self.cfg["env"]["plane"]["dynamicFriction"]
        self.plane_restitution = self.cfg["env"]["plane"]["restitution"]
        self.max_episode_length = self.cfg["env"]["episodeLength"]
        self._local_root_obs = self.cfg["env"]["localRootObs"]
        self._root_height_obs = self.cfg["env"].get("rootHeightObs", True)
        self._enable_early_termination = self.cfg["env"]["enableEarlyTermination"]
        self.key_bodies = self.cfg["env"]["keyBodies"]
        self._setup_character_props(self.key_bodies)
        self.cfg["env"]["numObservations"] = self.get_obs_size()
        self.cfg["env"]["numActions"] = self.get_action_size()
        self.cfg["device_type"], self.cfg["device_id"], self.cfg["headless"] = device_type, device_id, headless
        if not (self._local_root_obs and not self._root_height_obs and self._has_shape_obs
                and self._has_upright_start and not self._has_limb_weight_obs and self._enable_early_termination):
            raise NotImplementedError("emloco fused kernels cover the pacer.yaml observation configuration "
                                      "(localRootObs, rootHeightObs False, has_shape_obs, upright start)")
        super().__init__(cfg=self.cfg)
        self.dt = self.control_freq_inv * sim_params.dt
        self._setup_tensors()
        self.reward_raw = torch.zeros((self.num_envs, 1)).to(self.device)
        return

    # ------------------------------------------------------------------ configuration (humanoid.py:218-338)
    def load_smpl_configs(self, cfg):
        e = cfg["env"]
        self.smpl_humanoid = e["asset"]['assetFileName'] == "mjcf/smpl_humanoid.xml"
        if not self.smpl_humanoid:
            raise NotImplementedError("only mjcf/smpl_humanoid.xml is on the hot path")
        self._has_shape_obs = e.get("has_shape_obs", False)
        self._has_shape_obs_disc = e.get("has_shape_obs_disc", False)
        self._has_limb_weight_obs = e.get("has_weight_obs", False)
        self._has_limb_weight_obs_disc = e.get("has_weight_obs_disc", False)
        self.has_shape_variation = e.get("has_shape_variation", False)
        self._has_self_collision = e.get("has_self_collision", False)
        self._has_jt_limit = e.get("has_jt_limit", True)
        self._has_dof_subset = e.get("has_dof_subset", False)
        self._has_upright_start = e.get("has_upright_start", True)
        self._has_smpl_pd_offset = e.get("has_smpl_pd_offset", False)
        self._real_weight = e.get("real_weight", False)
        self._kp_scale = e.get("kp_scale", 1.0)
        self._kd_scale = e.get("kd_scale", self._kp_scale)
        self._freeze_toe = e.get("freeze_toe", True)
        self._bias_offset = e.get("bias_offset", False)
        self.motion_sym_loss = e.get("motion_sym_loss", False)
        self._body_names_orig = list(BODY_NAMES)
        self._body_names = self._body_names_orig
        self._dof_names = self._body_names[1:]
        self.limb_weight_group = [[self._body_names.index(g) for g in grp] for grp in [
            ['L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe'], ['R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe'],
            ['Pelvis', 'Torso', 'Spine', 'Chest', 'Neck', 'Head'], ['L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand'],
            ['R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']]]
        self.dof_subset = torch.from_numpy(dof_subset_indices().astype(np.int64))
        self.left_to_right_index = list(LEFT_TO_RIGHT)
        self.left_to_right_index_action = [4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11, 12, 18, 19, 20, 21, 22, 13, 14, 15, 16, 17]

    def _setup_character_props(self, key_bodies):                 # humanoid.py:496-571
        self._dof_body_ids = np.arange(1, len(self._body_names))
        self._dof_offsets = np.linspace(0, len(self._dof_names) * 3, len(self._body_names)).astype(int)
        self._dof_obs_size = len(self._dof_names) * 6
        self._num_actions = len(self._dof_names) * 3
        self._num_self_obs = 1 + len(self._body_names) * (3 + 6 + 3 + 3) - 3
        if self._has_shape_obs:
            self._num_self_obs += 11
        if self._has_limb_weight_obs:
            self._num_self_obs += 10
        if not self._root_height_obs:
            self._num_self_obs -= 1
        return

    def get_obs_size(self):
        return self._num_self_obs

    def get_self_obs_size(self):
        return self._num_self_obs

    def get_action_size(self):
        return self._num_actions

    def get_num_actors_per_env(self):
        return self._root_states.shape[0] // self.num_envs

    # ------------------------------------------------------------------ sim + envs (humanoid.py:428-437,643-948)
    def create_sim(self):
        self.up_axis_idx = self.set_sim_params_up_axis(self.sim_params, 'z')
        self.sim = super().create_sim(self.device_id, self.graphics_device_id, self.physics_engine, self.sim_params)
        self._create_ground_plane()
        self._create_envs(self.num_envs, self.cfg["env"]['envSpacing'], int(np.sqrt(self.num_envs)))
        return

    def _create_ground_plane(self):
        plane_params = gymapi.PlaneParams()
        plane_params.normal = gymapi.Vec3(0.0, 0.0, 1.0)
        plane_params.static_friction = self.plane_static_friction
        plane_params.dynamic_friction = self.plane_dynamic_friction
        plane_params.restitution = self.plane_restitution
        self.gym.add_ground(self.sim, plane_params)
        return

    def _synthetic_shapes(self, num_envs):
        """Synthetic AMASS-shaped population: a pool of subjects with limb scale U[0.9,1.1], mass
        scale U[0.7,1.4], betas ~ N(0,1) (observation pass-through), gender in {0,1,2}; envs cycle through it
        like humanoid.py:607 cycles through the AMASS subjects."""
        seed = int(self.cfg["env"].get("shape_seed", 0))
        pool = int(self.cfg["env"].get("num_shapes", 64))
        rng = np.random.default_rng(seed)
        base = smpl_humanoid()
        models, betas = [], []
        for k in range(min(pool, num_envs)):
            if self.has_shape_variation and k > 0:
                models.append(base.scaled(rng.uniform(0.9, 1.1), rng.uniform(0.7, 1.4)))
                b = rng.normal(size=17)
                b[0] = rng.integers(0, 3)
            else:
                models.append(base.scaled(1.0, 1.0))
                b = np.zeros(17)
            betas.append(b)
        return models, np.asarray(betas, np.float32)

    def _create_envs(self, num_envs, spacing, num_per_row):
        lower = gymapi.Vec3(-spacing, -spacing, 0.0)
        upper = gymapi.Vec3(spacing, spacing, spacing)
        self.humanoid_masses, self.humanoid_limb_and_weights = [], []
        asset_options = gymapi.AssetOptions()
        asset_options.angular_damping = 0.01
        asset_options.max_angular_velocity = 100.0
        asset_options.default_dof_drive_mode = gymapi.DOF_MODE_NONE
        models, betas = self._synthetic_shapes(num_envs)
        pool_assets = [self.gym.create_asset_from_model(self.sim, m, asset_options) for m in models]
        self.humanoid_assets = [pool_assets[i % len(pool_assets)] for i in range(num_envs)]
        self.humanoid_betas = torch.from_numpy(betas[np.arange(num_envs) % len(betas)]).float().to(self.device)
        humanoid_asset = self.humanoid_assets[0]
        motor_efforts = [prop.motor_effort for prop in self.gym.get_asset_actuator_properties(humanoid_asset)]
        self.max_motor_effort = max(motor_efforts)
        self.motor_efforts = to_torch(motor_efforts, device=self.device)
        self.torso_index = 0
        self.num_bodies = self.gym.get_asset_rigid_body_count(humanoid_asset)
        self.num_dof = self.gym.get_asset_dof_count(humanoid_asset)
        self.num_joints = self.gym.get_asset_joint_count(humanoid_asset)
        self.humanoid_handles, self.envs = [], []
        for i in range(self.num_envs):
            env_ptr = self.gym.create_env(self.sim, lower, upper, num_per_row)
            self._build_env(i, env_ptr, self.humanoid_assets[i])
            self.envs.append(env_ptr)
        self.humanoid_limb_and_weights = torch.stack(self.humanoid_limb_and_weights).to(self.device)
        dof_prop = self.gym.get_actor_dof_properties(self.envs[0], self.humanoid_handles[0])
        lo, hi = np.minimum(dof_prop['lower'], dof_prop['upper']), np.maximum(dof_prop['lower'], dof_prop['upper'])
        self.dof_limits_lower = to_torch(lo, device=self.device)
        self.dof_limits_upper = to_torch(hi, device=self.device)
        if (self._pd_control):
            self._build_pd_action_offset_scale()
        return

    def _build_env(self, env_id, env_ptr, humanoid_asset):       # humanoid.py:837-948
        col_group = env_id                                        # no inter-environment collision
        col_filter = 0 if self._has_self_collision else 1
        char_h = 0.89
        pos = torch.tensor(get_axis_params(char_h, self.up_axis_idx))
        start_pose = gymapi.Transform()
        start_pose.p = gymapi.Vec3(*pos)
        start_pose.r = gymapi.Quat(0.0, 0.0, 0.0, 1.0)
        humanoid_handle = self.gym.create_actor(env_ptr, humanoid_asset, start_pose, "humanoid", col_group, col_filter, 0)
        self.gym.enable_actor_dof_force_sensors(env_ptr, humanoid_handle)
        mass_ind = [prop.mass for prop in self.gym.get_actor_rigid_body_properties(env_ptr, humanoid_handle)]
        humanoid_mass = np.sum(mass_ind)
        self.humanoid_masses.append(humanoid_mass)
        limb_lengths = torch.norm(torch.tensor(humanoid_asset.model.joint_off, dtype=torch.float32), dim=-1)
        masses = torch.tensor(mass_ind, dtype=torch.float32)
        limb_lengths = [limb_lengths[group].sum() for group in self.limb_weight_group]
        masses = [masses[group].sum() for group in self.limb_weight_group]
        self.humanoid_limb_and_weights.append(torch.tensor(limb_lengths + masses))
        dof_prop = self.gym.get_asset_dof_properties(humanoid_asset)
        if (self._pd_control):
            dof_prop["driveMode"] = gymapi.DOF_MODE_POS
            if self.has_shape_variation:                           # humanoid.py:907-910
                pd_scale = humanoid_mass / self.cfg['env'].get('default_humanoid_mass', 77.0 if self._real_weight else 35.0)
                dof_prop['stiffness'] *= pd_scale * self._kp_scale
                dof_prop['damping'] *= pd_scale * self._kd_scale
        else:
            dof_prop["driveMode"] = gymapi.DOF_MODE_EFFORT
        self.gym.set_actor_dof_properties(env_ptr, humanoid_handle, dof_prop)
        if self._has_self_collision:                               # humanoid.py:917-944: per-shape collision filter bitmasks
            from ...model import SMPL_SHAPE_FILTERS
            props = self.gym.get_actor_rigid_shape_properties(env_ptr, humanoid_handle)
            assert len(SMPL_SHAPE_FILTERS) == len(props)
            for p_idx in range(len(props)):
                props[p_idx].filter = SMPL_SHAPE_FILTERS[p_idx]
            self.gym.set_actor_rigid_shape_properties(env_ptr, humanoid_handle, props)
        self.humanoid_handles.append(humanoid_handle)
        return

    def _build_pd_action_offset_scale(self):                     # humanoid.py:950-1025
        num_joints = len(self._dof_offsets) - 1
        lim_low = self.dof_limits_lower.cpu().numpy().copy()
        lim_high = self.dof_limits_upper.cpu().numpy().copy()
        for j in range(num_joints):
            o, size = self._dof_offsets[j], self._dof_offsets[j + 1] - self._dof_offsets[j]
            if not self._bias_offset and size == 3:
                scale = max(np.max(np.abs(lim_low[o:o + size])), np.max(np.abs(lim_high[o:o + size])))
                scale = min(1.2 * scale, np.pi)
                lim_low[o:o + size], lim_high[o:o + size] = -scale, scale
            else:
                mid = 0.5 * (lim_high[o:o + size] + lim_low[o:o + size])
                sc = 0.7 * (lim_high[o:o + size] - lim_low[o:o + size])
                lim_low[o:o + size], lim_high[o:o + size] = mid - sc, mid + sc
        self._pd_action_offset = to_torch(0.5 * (lim_high + lim_low), device=self.device)
        self._pd_action_scale = to_torch(0.5 * (lim_high - lim_low), device=self.device)
        self._L_knee_dof_idx = self._dof_names.index("L_Knee") * 3 + 1
        self._R_knee_dof_idx = self._dof_names.index("R_Knee") * 3 + 1
        self._pd_action_scale[self._L_knee_dof_idx] = 5
        self._pd_action_scale[self._R_knee_dof_idx] = 5
        if self._has_smpl_pd_offset:
            sgn = np.pi / 2 if self._has_upright_start else np.pi / 6
            self._pd_action_offset[self._dof_names.index("L_Shoulder") * 3] = -sgn
            self._pd_action_offset[self._dof_names.index("R_Shoulder") * 3] = sgn
        zero = np.zeros(self.num_dof, np.uint8)                   # humanoid.py:1190-1196: hands, (frozen) toes
        for n in ["L_Hand", "R_Hand"] + (["L_Toe", "R_Toe"] if self._freeze_toe else []):
            i = self._dof_names.index(n) * 3
            zero[i:i + 3] = 1
        self._pd_zero_mask = torch.from_numpy(zero).to(self.device)
        return

    # ------------------------------------------------------------------ tensors (humanoid.py:135-216)
    def _setup_tensors(self):
        actor_root_state = self.gym.acquire_actor_root_state_tensor(self.sim)
        dof_state_tensor = self.gym.acquire_dof_state_tensor(self.sim)
        rigid_body_state = self.gym.acquire_rigid_body_state_tensor(self.sim)
        contact_force_tensor = self.gym.acquire_net_contact_force_tensor(self.sim)
        dof_force_tensor = self.gym.acquire_dof_force_tensor(self.sim)
        self.dof_force_tensor = gymtorch.wrap_tensor(dof_force_tensor).view(self.num_envs, self.num_dof)
        self._root_states = gymtorch.wrap_tensor(actor_root_state)
        num_actors = self.get_num_actors_per_env()
        self._humanoid_root_states = self._root_states.view(self.num_envs, num_actors, actor_root_state.shape[-1])[..., 0, :]
        self._initial_humanoid_root_states = self._humanoid_root_states.clone()
        self._initial_humanoid_root_states[:, 7:13] = 0
        self._humanoid_actor_ids = num_actors * torch.arange(self.num_envs, device=self.device, dtype=torch.int32)
        self._dof_state = gymtorch.wrap_tensor(dof_state_tensor)
        dofs_per_env = self._dof_state.shape[0] // self.num_envs
        self._dof_pos = self._dof_state.view(self.num_envs, dofs_per_env, 2)[..., :self.num_dof, 0]
        self._dof_vel = self._dof_state.view(self.num_envs, dofs_per_env, 2)[..., :self.num_dof, 1]
        self._initial_dof_pos = torch.zeros_like(self._dof_pos, device=self.device, dtype=torch.float)
        self._initial_dof_vel = torch.zeros_like(self._dof_vel, device=self.device, dtype=torch.float)
        self._rigid_body_state = gymtorch.wrap_tensor(rigid_body_state)
        bodies_per_env = self._rigid_body_state.shape[0] // self.num_envs
        self._rigid_body_state_reshaped = self._rigid_body_state.view(self.num_envs, bodies_per_env, 13)
        self._rigid_body_pos = self._rigid_body_state_reshaped[..., :self.num_bodies, 0:3]
        self._rigid_body_rot = self._rigid_body_state_reshaped[..., :self.num_bodies, 3:7]
        self._rigid_body_vel = self._rigid_body_state_reshaped[..., :self.num_bodies, 7:10]
        self._rigid_body_ang_vel = self._rigid_body_state_reshaped[..., :self.num_bodies, 10:13]
        contact_force_tensor = gymtorch.wrap_tensor(contact_force_tensor)
        self._contact_forces = contact_force_tensor.view(self.num_envs, bodies_per_env, 3)[..., :self.num_bodies, :]
        self._terminate_buf = torch.ones(self.num_envs, device=self.device, dtype=torch.long)
        self._build_termination_heights()
        contact_bodies = self.cfg["env"]["contactBodies"]
        self._key_body_ids = self._build_key_body_ids_tensor(self.key_bodies)
        self._contact_body_ids = self._build_contact_body_ids_tensor(contact_bodies)
        if self.motion_sym_loss:
            self._flip_obs_buf = torch.zeros((self.num_envs, self.num_obs), device=self.device, dtype=torch.float)
        else:
            raise NotImplementedError("emloco fused kernels always produce the mirrored observations (motion_sym_loss: True)")
        # the PD-target buffer IS the simulator's own target tensor when the native sim exists: set_dof_position_target_tensor
        # then has nothing to copy (emloco_sim_set_pd_targets skips an aliasing pointer)
        native = getattr(self.sim, "native", None)
        self._pd_targets = native.pd_target.view(self.num_envs, self.num_dof) if native is not None else \
            torch.zeros((self.num_envs, self.num_dof), device=self.device, dtype=torch.float)
        self._post = PostPhysics(self.device, key_bodies=self.key_bodies, contact_bodies=contact_bodies)
        self._post_bufs = None

    def _build_termination_heights(self):
        self._termination_heights = to_torch(np.array([self.cfg["env"]["terminationHeight"]] * self.num_bodies), device=self.device)
        head_id = self.gym.find_actor_rigid_body_handle(self.envs[0], self.humanoid_handles[0], "head")
        self._termination_heights[head_id] = max(0.3, float(self._termination_heights[head_id]))

    def _build_key_body_ids_tensor(self, key_body_names):
        return to_torch([self._body_names.index(n) for n in key_body_names], device=self.device, dtype=torch.long)

    def _build_contact_body_ids_tensor(self, contact_body_names):
        ids = [self.gym.find_actor_rigid_body_handle(self.envs[0], self.humanoid_handles[0], n) for n in contact_body_names]
        assert all(i != -1 for i in ids)
        return to_torch(ids, device=self.device, dtype=torch.long)

    # ------------------------------------------------------------------ reset (humanoid.py:439-481)
    def reset(self, env_ids=None):
        self.wait_obs()             # a deferred observation pass (fused_chain) goes first: it must see the pre-reset flags and state
        if (env_ids is None):
            env_ids = to_torch(np.arange(self.num_envs), device=self.device, dtype=torch.long)
        self._reset_envs(env_ids)
        return

    def _reset_envs(self, env_ids):
        if (len(env_ids) > 0):
            self._reset_actors(env_ids)
            self._reset_env_tensors(env_ids)
            self._refresh_sim_tensors()
            self._compute_observations(env_ids)
        return

    def _reset_actors(self, env_ids):
        self._humanoid_root_states[env_ids] = self._initial_humanoid_root_states[env_ids]
        self._dof_pos[env_ids] = self._initial_dof_pos[env_ids]
        self._dof_vel[env_ids] = self._initial_dof_vel[env_ids]
        return

    def _reset_env_tensors(self, env_ids):
        env_ids = env_ids.to(self.device)
        env_ids_int32 = self._humanoid_actor_ids[env_ids].contiguous()
        self.gym.set_actor_root_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self._root_states),
                                                     gymtorch.unwrap_tensor(env_ids_int32), len(env_ids_int32))
        self.gym.set_dof_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self._dof_state),
                                              gymtorch.unwrap_tensor(env_ids_int32), len(env_ids_int32))
        self.progress_buf[env_ids] = 0
        self.reset_buf[env_ids] = 0
        self._terminate_buf[env_ids] = 0
        self._contact_forces[env_ids] = 0
        return

    def _refresh_sim_tensors(self):                               # humanoid.py:1039-1047 (live views: no-ops)
        self.gym.refresh_dof_state_tensor(self.sim)
        self.gym.refresh_actor_root_state_tensor(self.sim)
        self.gym.refresh_rigid_body_state_tensor(self.sim)
        self.gym.refresh_force_sensor_tensor(self.sim)
        self.gym.refresh_dof_force_tensor(self.sim)
        self.gym.refresh_net_contact_force_tensor(self.sim)
        return

    # ------------------------------------------------------------------ step (humanoid.py:1184-1232)
    def pre_physics_step(self, actions):
        # the reference clones (humanoid.py:1185): `self.actions` must not alias a buffer the caller goes on writing into (a policy
        # that reuses its output tensor).  Copied into a persistent buffer of the task: same semantics, no allocation per step
        self.wait_obs()             # the observation launch of the last step reads what this step's rigid-body launch overwrites
        if getattr(self, "_actions_buf", None) is None or self._actions_buf.shape != actions.shape:
            self._actions_buf = torch.empty(actions.shape, dtype=torch.float32, device=self.device)
        if not self._pd_control:                                  # humanoid.py:1203-1207: joint torques, effort drives
            self._actions_buf.copy_(actions)
            self.actions = self._actions_buf
            forces = self.actions * self.motor_efforts.unsqueeze(0) * self.power_scale
            if not self.gym.set_dof_actuation_force_tensor(self.sim, gymtorch.unwrap_tensor(forces.contiguous())):
                raise RuntimeError("set_dof_actuation_force_tensor failed (simulator not prepared, or a tensor of the wrong shape / dtype)")
            return
        # pd_tar = offset + scale * a with hands / frozen toes zeroed (humanoid.py:1188-1202,1281-1283) and the task's copy of the
        # actions, one launch
        # (a no-op for a float32 contiguous tensor on the sim device; an rl_device='cpu' / other-GPU tensor is moved first as the
        # reference's `actions.to(self.device).clone()` does: the kernel takes a raw pointer)
        src = actions.to(device=self.device, dtype=torch.float32).contiguous()
        self._post.pd_targets(src, self._pd_action_offset, self._pd_action_scale, self._pd_zero_mask, self._pd_targets,
                              actions_copy=self._actions_buf)
        self.actions = self._actions_buf
        if not self.gym.set_dof_position_target_tensor(self.sim, gymtorch.unwrap_tensor(self._pd_targets)):
            raise RuntimeError("set_dof_position_target_tensor failed (simulator not prepared, or a tensor of the wrong shape / dtype)")
        return

    def _action_to_pd_targets(self, action):
        return self._pd_action_offset + self._pd_action_scale * action

    def _post_mode_step(self):
        return L.POST_ADVANCE | L.POST_OBS | L.POST_REWARD | L.POST_RESET

    # A rollout loop may hand the task its LocoVal return bookkeeping (LocoValRollout.attach: an EmlocoLocoValStep): the launch that
    # computes a step's rewards and reset flags for all envs then also advances the returns (emloco_task_post_physics_returns) and
    # raises `_returns_in_flags` for the loop to see that its own launch is not needed this step.
    _returns_hook = None
    _returns_in_flags = False

    def attach_returns(self, step_struct, before=None):
        """`before` (optional callable) runs right ahead of the launch -- the loop's "staging buffers are free" wait belongs there, behind
        the rigid-body launch, where it never blocks."""
        self._returns_hook = step_struct
        self._returns_before = before if step_struct is not None else None

    def _launch_post(self, mode, env_ids=None):
        if self._post_bufs is None:
            self._post_bufs = self._make_post_bufs()
            if hasattr(self, "_amp_ring_to_bufs"):
                self._amp_ring_to_bufs()
        if (self._returns_hook is not None and env_ids is None and (mode & L.POST_REWARD) and (mode & L.POST_RESET)
                and not (mode & L.POST_SKIP_DONE)):
            if getattr(self, "_returns_before", None) is not None:
                self._returns_before()
            inv = self.inverted if isinstance(getattr(self, "inverted", None), torch.Tensor) else None
            if inv is not None and inv.dtype == torch.bool:
                inv = inv.view(torch.uint8)
            self._post.run(self._post_bufs, mode, None, returns=(self._returns_hook, inv.contiguous() if inv is not None else None))
            self._returns_in_flags = True
            return
        ids = None if env_ids is None else self._humanoid_actor_ids[env_ids.to(self.device)].contiguous()
        self._post.run(self._post_bufs, mode, ids)

    def _make_post_bufs(self):
        raise NotImplementedError("the fused post-physics kernel needs the trajectory / terrain task (HumanoidPedestrianTerrain)")

    # Opt-in (set by a rollout loop that calls wait_obs() before anything reads the observations): the step's launch is split
    # into progress / reward / reset flags on the caller's stream and observations (+ AMP rows) on a side stream, so the
    # observations of the ~4000 live envs are built while the caller's stream already resets the finished ones.  The side
    # launch leaves the finished envs alone (POST_SKIP_DONE): their observation rows are rebuilt by the reset path, as in the
    # reference (humanoid.py:1140-1160 recomputes the observations of the reset envs); their terminal AMP rows come from the
    # flags launch (POST_AMP_DONE_ONLY).
    overlap_obs = False
    # Opt-in (set by a rollout loop that calls reset_done() -- or wait_obs() -- before anything reads the observations of a step):
    # post_physics_step launches only progress / reward / reset flags (+ the terminal AMP rows of the envs that finish); the
    # observations and AMP rows of the live envs are DEFERRED into the next reset_done(), where they ride in the same launch as the
    # reset chain of the finished envs (emloco_task_reset_obs: one launch, no side stream, no event packets).  wait_obs() -- which
    # pre_physics_step calls too -- launches a deferred observation pass on the spot when no reset_done() took it, so a caller that
    # never resets still gets its observations before the state moves on.  Takes precedence over overlap_obs.
    fused_chain = False
    # with fused_chain: keep the AMP history shift / newest AMP row of EVERY env in the flags launch and defer only the observation
    # rows -- for a loop that scores this step's AMP observations (discriminator reward, amp_continuous_value.py:90-96) before it resets
    fused_amp_early = False
    _obs_deferred = 0

    def _make_obs_stream(self):
        # high priority: the short launches that run beside the rigid-body kernel get the first wave slot that frees up
        self._obs_stream = torch.cuda.Stream(device=self.device, priority=-1)
        self._ev_flags, self._ev_obs = torch.cuda.Event(), torch.cuda.Event()

    def wait_obs(self):
        """Make the caller's stream wait for the observation launch of the last step (no-op without overlap_obs); a deferred
        observation pass (fused_chain) that no reset_done() has taken is launched here."""
        if self._obs_deferred:
            mode, self._obs_deferred = self._obs_deferred, 0
            self._launch_post(mode | L.POST_SKIP_DONE)
        if getattr(self, "_obs_pending", False):
            torch.cuda.current_stream(self.device).wait_event(self._ev_obs)
            self._obs_pending = False

    def post_physics_step(self):
        self._refresh_sim_tensors()
        mode = self._post_mode_step()
        if self.fused_chain and torch.device(self.device).type == "cuda" and getattr(self, "_fused_reset", False):
            self.wait_obs()
            if self.fused_amp_early:
                side_mode = mode & L.POST_OBS
                self._launch_post(mode & ~side_mode)
            else:
                side_mode = mode & (L.POST_OBS | L.POST_AMP_SHIFT | L.POST_AMP_ROW)
                amp = mode & (L.POST_AMP_SHIFT | L.POST_AMP_ROW)
                self._launch_post((mode & ~side_mode) | (amp | L.POST_AMP_DONE_ONLY if amp else 0))
            self._obs_deferred = side_mode
        elif self.overlap_obs and torch.device(self.device).type == "cuda":
            if getattr(self, "_obs_stream", None) is None:
                self._make_obs_stream()
            self.wait_obs()                                       # nobody asked for the previous step's observations
            side_mode = mode & (L.POST_OBS | L.POST_AMP_SHIFT | L.POST_AMP_ROW)
            # progress += 1, reward, reset flags -- and the AMP rows of the envs that finish on this step: their terminal AMP
            # observations are scored by the caller (amp_continuous_value.py:90-96) and must be taken from the state the reset
            # path is about to overwrite, so they cannot ride on the side launch
            amp = mode & (L.POST_AMP_SHIFT | L.POST_AMP_ROW)
            self._launch_post((mode & ~side_mode) | (amp | L.POST_AMP_DONE_ONLY if amp else 0))
            main = torch.cuda.current_stream(self.device)
            self._ev_flags.record(main)
            self._obs_stream.wait_event(self._ev_flags)
            with torch.cuda.stream(self._obs_stream):
                self._launch_post(side_mode | L.POST_SKIP_DONE)
                self._ev_obs.record(self._obs_stream)
            self._obs_pending = True
        else:
            self._launch_post(mode)                               # progress += 1, obs, reward, reset [, AMP] in one launch
        self.extras["terminate"] = self._terminate_buf
        self.extras["reward_raw"] = self.reward_raw.detach()
        if self.motion_sym_loss:
            self.extras['flip_obs'] = self._flip_obs_buf
            self.extras['obs'] = self.obs_buf
        return

    def _compute_observations(self, env_ids=None):
        self._launch_post(L.POST_OBS, env_ids)

    def _compute_reward(self, actions):
        self._launch_post(L.POST_REWARD)

    def _compute_reset(self):
        self._launch_post(L.POST_RESET)
