"""HumanoidAMP: AMP observation history and reference-state initialisation.

Mirror of pacer/pacer/env/tasks/humanoid_amp.py (class HumanoidAMP :47-660).  The AMP row and the history
shift run inside the fused post-physics launch; the reference-state reset path is host-side torch plus the
indexed C-ABI setters.  The SMPL-mesh ground-height fix (:321-379, needs the licensed SMPL model) is replaced
by the lowest-collision-geometry fix.
"""
from enum import Enum

import numpy as np
import torch

from ... import _lib as L
from ...gym import gymtorch
from ...gym.torch_utils import quat_apply
from ...utils.flags import flags
from ...utils.motion_lib_synthetic import MotionLibSynthetic
from .humanoid import Humanoid


class HumanoidAMP(Humanoid):
    class StateInit(Enum):
        Default = 0
        Start = 1
        Random = 2
        Hybrid = 3

    def __init__(self, cfg, sim_params, physics_engine, device_type, device_id, headless):
        state_init = cfg["env"]["stateInit"]
        self._state_init = HumanoidAMP.StateInit[state_init]
        self._hybrid_init_prob = cfg["env"]["hybridInitProb"]
        self._num_amp_obs_steps = cfg["env"]["numAMPObsSteps"]
        self._amp_root_height_obs = cfg["env"].get("ampRootHeightObs", False)
        assert (self._num_amp_obs_steps >= 2)
        self._enable_hist_obs = cfg["env"].get("enableHistObs", False)
        self._reset_default_env_ids = []
        self._reset_ref_env_ids = []
        self._state_reset_happened = False
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type,
                         device_id=device_id, headless=headless)
        if self._num_amp_obs_steps != L.AMP_STEPS or self._num_amp_obs_per_step != L.AMP_ROW:
            raise NotImplementedError(f"fused AMP kernel is built for {L.AMP_STEPS} steps x {L.AMP_ROW} features")
        self._motion_start_times = torch.zeros(self.num_envs).to(self.device)
        self._sampled_motion_ids = torch.zeros(self.num_envs).long().to(self.device)
        self._load_motion(cfg['env'].get('motion_file', None))
        self._amp_obs_buf = torch.zeros((self.num_envs, self._num_amp_obs_steps, self._num_amp_obs_per_step),
                                        device=self.device, dtype=torch.float)
        self._curr_amp_obs_buf = self._amp_obs_buf[:, 0]
        self._hist_amp_obs_buf = self._amp_obs_buf[:, 1:]
        self._amp_obs_demo_buf = None
        self._build_lowest_point_tables()
        return

    def _setup_character_props(self, key_bodies):                 # humanoid_amp.py:230-265
        super()._setup_character_props(key_bodies)
        n = 13 + self._dof_obs_size + len(self._dof_names) * 3 + 3 * len(key_bodies)
        if not self._amp_root_height_obs:
            n -= 1
        if self._has_dof_subset:
            n -= (6 + 3) * int((len(self._dof_names) * 3 - len(self.dof_subset)) / 3)
        if self._has_shape_obs_disc:
            n += 11
        if self._has_limb_weight_obs_disc:
            n += 10
        self._num_amp_obs_per_step = n
        if self._enable_hist_obs:
            self._num_self_obs += self._num_amp_obs_steps * self._num_amp_obs_per_step
        return

    def _load_motion(self, motion_file):
        assert (self._dof_offsets[-1] == self.num_dof)
        if motion_file not in (None, "", "synthetic"):
            # an AMASS pickle in the reference's format (convert_amass_isaac.py:309-317): one clip per env on that env's skeleton
            # (humanoid_amp.py:256-271); the height fix through the SMPL mesh is replaced by the lowest-collision-point fix at reset
            from ...utils.motion_lib_smpl import MotionLib
            self._motion_lib = MotionLib(motion_file=motion_file, key_body_ids=self._key_body_ids.cpu().numpy(), device=self.device,
                                         fix_height=False, masterfoot_conifg=None, min_length=self._min_motion_len if hasattr(self, "_min_motion_len") else -1)
            self._motion_lib.load_motions(skeleton_trees=[a.model for a in self.humanoid_assets], gender_betas=self.humanoid_betas.cpu(),
                                          limb_weights=None, random_sample=True)
            return
        self._motion_lib = MotionLibSynthetic(self.humanoid_assets[0].model, self._key_body_ids.cpu().numpy(),
                                              self.device, num_motions=int(self.cfg["env"].get("num_motions", 64)),
                                              seed=int(self.cfg["env"].get("motion_seed", 0)))
        return

    def _build_lowest_point_tables(self):
        """Per-env ground-contact candidate points (body frame) and radii, to place reset poses on the ground."""
        from ...model import GEOM_CAPSULE, GEOM_SPHERE
        pool = {}
        pts, rad = [], []
        for a in self.humanoid_assets:
            if id(a) not in pool:
                m, p, r, bodies = a.model, [], [], []
                for b in range(m.num_bodies):
                    if m.geom_type[b] == GEOM_SPHERE:
                        cand = [m.geom_a[b]]
                    elif m.geom_type[b] == GEOM_CAPSULE:
                        cand = [m.geom_a[b], m.geom_b[b]]
                    else:
                        cand = [m.geom_a[b] + m.geom_b[b] * np.array([sx, sy, sz]) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
                    p += cand
                    r += [m.geom_r[b]] * len(cand)
                    bodies += [b] * len(cand)
                pool[id(a)] = (np.asarray(p, np.float32), np.asarray(r, np.float32), bodies)
            pts.append(pool[id(a)][0])
            rad.append(pool[id(a)][1])
        self._cand_body = torch.tensor(pool[id(self.humanoid_assets[0])][2], dtype=torch.long, device=self.device)
        self._cand_points = torch.from_numpy(np.stack(pts)).to(self.device)
        self._cand_radius = torch.from_numpy(np.stack(rad)).to(self.device)

    def _lowest_point(self, env_ids):
        pos = self._rigid_body_pos[env_ids][:, self._cand_body]
        rot = self._rigid_body_rot[env_ids][:, self._cand_body]
        wp = quat_apply(rot, self._cand_points[env_ids]) + pos
        return (wp[..., 2] - self._cand_radius[env_ids]).min(dim=-1).values

    # ------------------------------------------------------------------ step
    def _post_mode_step(self):
        return super()._post_mode_step() | L.POST_AMP_SHIFT | L.POST_AMP_ROW

    def post_physics_step(self):                                  # humanoid_amp.py:139-157
        if self.amp_ring:
            self._amp_head = (self._amp_head + self._num_amp_obs_steps - 1) % self._num_amp_obs_steps      # the step's "shift"
            self._amp_ring_to_bufs()
        super().post_physics_step()
        # (ring: the physical buffer is not the reference's layout -- who reads the AMP observations asks amp_obs_logical())
        self.extras["amp_obs"] = None if self.amp_ring else self._amp_obs_buf.view(-1, self.get_num_amp_obs())
        return

    # The AMP history as a RING (opt-in, for a loop that does not read every step's AMP observations as one tensor): the reference
    # shifts rows 0..13 of every env to 1..14 each step (humanoid_amp.py:585-594) -- 23 KB of the ~38 KB an env-step moves.  With the
    # ring the row the reference calls k lives in physical row (head + k) % 15 of `_amp_obs_buf`, a step moves `head` back by one and
    # writes only the newest row (EmlocoTaskBufs.amp_ring); `amp_obs_logical()` is the reference's tensor, built on demand.
    amp_ring = False
    _amp_head = 0

    def enable_amp_ring(self, on=True):
        on = bool(on)
        if on == self.amp_ring:
            return
        if on and (torch.device(self.device).type != "cuda" or not getattr(self, "_fused_reset", False)):
            raise RuntimeError("amp ring: needs the fused device step / reset kernels (a CUDA device, fused_reset)")
        if hasattr(self, "wait_obs"):
            self.wait_obs()                          # a deferred observation / AMP pass of the last step runs in the layout it was deferred in
        if not on:                                   # back to the reference's layout, in place
            self._amp_obs_buf.copy_(self.amp_obs_logical().view_as(self._amp_obs_buf))
        self._amp_head = 0                           # (switching on: physical = logical at head 0, nothing moves)
        self.amp_ring = on
        self._amp_ring_to_bufs()
        if not on:
            self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.get_num_amp_obs())

    def _amp_ring_to_bufs(self):
        v = (1 + self._amp_head) if self.amp_ring else 0
        for name in ("_post_bufs", "_reset_bufs", "_reset_bufs_noamp"):
            b = getattr(self, name, None)
            if b is not None:
                b.amp_ring = v

    def amp_obs_logical(self):
        """(num_envs, 15 * 206) AMP observations in the reference's order (newest first), whatever the physical layout."""
        if not self.amp_ring or self._amp_head == 0:
            return self._amp_obs_buf.view(-1, self.get_num_amp_obs())
        return torch.roll(self._amp_obs_buf, shifts=-self._amp_head, dims=1).reshape(-1, self.get_num_amp_obs())

    def _amp_rows_phys(self, first, count):
        """physical rows of the logical rows first .. first + count - 1 (a LongTensor for index assignment)"""
        return (torch.arange(first, first + count, device=self.device) + (self._amp_head if self.amp_ring else 0)) % self._num_amp_obs_steps

    def get_num_amp_obs(self):
        return self._num_amp_obs_steps * self._num_amp_obs_per_step

    def fetch_amp_obs_demo(self, num_samples):                    # humanoid_amp.py:168-228
        motion_ids = self._motion_lib.sample_motions(num_samples)
        truncate = self.dt * (self._num_amp_obs_steps - 1)
        motion_times0 = self._motion_lib.sample_time(motion_ids, truncate_time=truncate) + truncate
        dt = self.dt
        motion_ids = torch.tile(motion_ids.unsqueeze(-1), [1, self._num_amp_obs_steps]).view(-1)
        steps = -dt * torch.arange(0, self._num_amp_obs_steps, device=self.device)
        motion_times = (motion_times0.unsqueeze(-1) + steps).view(-1)
        rows = self._amp_rows_from_motion(motion_ids, motion_times, self.humanoid_betas[0:1].expand(motion_ids.shape[0], -1))
        return rows.view(num_samples, self.get_num_amp_obs())

    def _amp_rows_from_motion(self, motion_ids, motion_times, betas):
        r = self._motion_lib.get_motion_state_smpl(motion_ids, motion_times)
        return self._post.amp_rows(r["root_pos"], r["root_rot"], r["root_vel"], r["root_ang_vel"], r["dof_pos"],
                                   r["dof_vel"], r["key_pos"].reshape(-1, 12), betas.contiguous())

    # ------------------------------------------------------------------ reset (humanoid_amp.py:284-563)
    def _reset_envs(self, env_ids):
        self._reset_default_env_ids = []
        self._reset_ref_env_ids = []
        if len(env_ids) > 0:
            self._state_reset_happened = True
        super()._reset_envs(env_ids)
        self._init_amp_obs(env_ids)
        return

    def _reset_actors(self, env_ids):
        if (self._state_init == HumanoidAMP.StateInit.Default):
            self._reset_default(env_ids)
        elif (self._state_init in (HumanoidAMP.StateInit.Start, HumanoidAMP.StateInit.Random)):
            self._reset_ref_state_init(env_ids)
        elif (self._state_init == HumanoidAMP.StateInit.Hybrid):
            self._reset_hybrid_state_init(env_ids)
        return

    def _reset_default(self, env_ids):
        self._humanoid_root_states[env_ids] = self._initial_humanoid_root_states[env_ids]
        self._dof_pos[env_ids] = self._initial_dof_pos[env_ids]
        self._dof_vel[env_ids] = self._initial_dof_vel[env_ids]
        self._reset_default_env_ids = env_ids
        return

    def _sample_time(self, motion_ids):
        return self._motion_lib.sample_time(motion_ids)

    def _get_state_from_motionlib(self, motion_ids, motion_times):
        r = self._motion_lib.get_motion_state_smpl(motion_ids, motion_times)
        return (r["root_pos"], r["root_rot"], r["dof_pos"], r["root_vel"], r["root_ang_vel"], r["dof_vel"], r["key_pos"],
                r["rg_pos"], r["rb_rot"], r["body_vel"], r["body_ang_vel"])

    def _sample_ref_state(self, env_ids):
        num_envs = env_ids.shape[0]
        motion_ids = self._motion_lib.sample_motions(num_envs)
        if (self._state_init in (HumanoidAMP.StateInit.Random, HumanoidAMP.StateInit.Hybrid)):
            motion_times = self._sample_time(motion_ids)
        else:
            motion_times = torch.zeros(num_envs, device=self.device)
        root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot, _, _ = \
            self._get_state_from_motionlib(motion_ids, motion_times)
        return motion_ids, motion_times, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot

    def _reset_ref_state_init(self, env_ids):
        motion_ids, motion_times, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot = \
            self._sample_ref_state(env_ids)
        self._set_env_state(env_ids=env_ids, root_pos=root_pos, root_rot=root_rot, dof_pos=dof_pos, root_vel=root_vel,
                            root_ang_vel=root_ang_vel, dof_vel=dof_vel, rigid_body_pos=rb_pos, rigid_body_rot=rb_rot)
        self._reset_ref_env_ids = env_ids
        self._reset_ref_motion_ids = motion_ids
        self._reset_ref_motion_times = motion_times
        self._motion_start_times[env_ids] = motion_times
        self._sampled_motion_ids[env_ids] = motion_ids
        return

    def _reset_hybrid_state_init(self, env_ids):
        num_envs = env_ids.shape[0]
        ref_probs = torch.full((num_envs,), self._hybrid_init_prob, device=self.device)
        ref_init_mask = torch.bernoulli(ref_probs) == 1.0
        ref_reset_ids = env_ids[ref_init_mask]
        if (len(ref_reset_ids) > 0):
            self._reset_ref_state_init(ref_reset_ids)
        default_reset_ids = env_ids[torch.logical_not(ref_init_mask)]
        if (len(default_reset_ids) > 0):
            self._reset_default(default_reset_ids)
        return

    def _set_env_state(self, env_ids, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel,
                       rigid_body_pos=None, rigid_body_rot=None):
        """humanoid_amp.py:537-563.  Body poses are not written: the indexed setters recompute them by forward
        kinematics on each env's own skeleton, after which the pose is lowered/raised so its lowest collision
        point rests `height_tolerance` above the ground (replaces the SMPL-mesh fix :321-379)."""
        self._humanoid_root_states[env_ids, 0:3] = root_pos
        self._humanoid_root_states[env_ids, 3:7] = root_rot
        self._humanoid_root_states[env_ids, 7:10] = root_vel
        self._humanoid_root_states[env_ids, 10:13] = root_ang_vel
        self._dof_pos[env_ids] = dof_pos
        self._dof_vel[env_ids] = dof_vel
        ids32 = self._humanoid_actor_ids[env_ids].contiguous()
        self.gym.set_actor_root_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self._root_states), gymtorch.unwrap_tensor(ids32), len(ids32))
        self.gym.set_dof_state_tensor_indexed(self.sim, gymtorch.unwrap_tensor(self._dof_state), gymtorch.unwrap_tensor(ids32), len(ids32))
        height_tolerance = 0.02
        ground = getattr(self, "_reset_ground_height", None)
        gz = 0.0 if ground is None else ground
        self._humanoid_root_states[env_ids, 2] -= (self._lowest_point(env_ids) - gz - height_tolerance)
        return

    def _init_amp_obs(self, env_ids):                             # humanoid_amp.py:486-535
        self._compute_amp_observations(env_ids)
        if (len(self._reset_default_env_ids) > 0):
            self._init_amp_obs_default(self._reset_default_env_ids)
        if (len(self._reset_ref_env_ids) > 0):
            self._init_amp_obs_ref(self._reset_ref_env_ids, self._reset_ref_motion_ids, self._reset_ref_motion_times)
        return

    def _init_amp_obs_default(self, env_ids):
        if self.amp_ring:
            ids = torch.as_tensor(env_ids, device=self.device).long()
            cur = self._amp_obs_buf[ids, self._amp_rows_phys(0, 1)[0]]
            self._amp_obs_buf[ids.unsqueeze(1), self._amp_rows_phys(1, self._num_amp_obs_steps - 1).unsqueeze(0)] = cur.unsqueeze(1)
            return
        self._hist_amp_obs_buf[env_ids] = self._curr_amp_obs_buf[env_ids].unsqueeze(-2)
        return

    def _init_amp_obs_ref(self, env_ids, motion_ids, motion_times):
        dt = self.dt
        n = self._num_amp_obs_steps - 1
        mids = torch.tile(motion_ids.unsqueeze(-1), [1, n]).view(-1)
        steps = -dt * (torch.arange(0, n, device=self.device) + 1)
        mtimes = (motion_times.unsqueeze(-1) + steps).view(-1)
        betas = self.humanoid_betas[env_ids].unsqueeze(1).expand(-1, n, -1).reshape(-1, 17)
        rows = self._amp_rows_from_motion(mids, mtimes, betas)
        if self.amp_ring:
            ids = torch.as_tensor(env_ids, device=self.device).long()
            self._amp_obs_buf[ids.unsqueeze(1), self._amp_rows_phys(1, n).unsqueeze(0)] = rows.view(len(env_ids), n, self._num_amp_obs_per_step)
            return
        self._hist_amp_obs_buf[env_ids] = rows.view(len(env_ids), n, self._num_amp_obs_per_step)
        return

    def _update_hist_amp_obs(self, env_ids=None):
        self._launch_post(L.POST_AMP_SHIFT, env_ids)

    def _compute_amp_observations(self, env_ids=None):
        if env_ids is not None and len(env_ids) == 0:
            return
        self._launch_post(L.POST_AMP_ROW, env_ids)
