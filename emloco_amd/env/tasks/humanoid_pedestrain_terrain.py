"""HumanoidPedestrianTerrain: the default EmLoco task (trajectory following over a height-field terrain, with the
LocoVal inputs captured at reset).

Mirror of pacer/pacer/env/tasks/humanoid_pedestrain_terrain.py (file name spelt as in the reference):
class HumanoidPedestrianTerrain :34-930, class Terrain :1135-1463.  Everything per step -- 368 self obs,
30 trajectory obs, 32x32 height obs, mirrored obs, reward, reset masks, AMP -- is one fused HIP launch.
"""
import os
import numpy as np
import torch
from scipy import ndimage

from ... import _lib as L
from ...gym import gymapi
from ...gym import torch_utils as tu
from ...gym.terrain_utils import (SubTerrain, convert_heightfield_to_trimesh, discrete_obstacles_terrain, pyramid_sloped_terrain,
                                  pyramid_stairs_terrain, random_uniform_terrain, stepping_stones_terrain)
from ...utils.draw_utils import draw_curve, draw_disk, draw_ellipse, draw_polygon
from ...utils.flags import flags
from . import humanoid_traj


class HumanoidPedestrianTerrain(humanoid_traj.HumanoidTraj):
    def __init__(self, cfg, sim_params, physics_engine, device_type, device_id, headless):
        self.real_mesh = getattr(cfg.get('args', None), "real_mesh", False)
        if self.real_mesh:
            raise NotImplementedError("real-mesh city terrain is out of scope")
        self.device = "cpu"
        if device_type == "cuda" or device_type == "GPU":
            self.device = "cuda" + ":" + str(device_id)
        self.cfg = cfg
        self.num_envs = cfg["env"]["numEnvs"]
        self.sensor_extent = cfg["env"].get("sensor_extent", 2)
        self.sensor_res = cfg["env"].get("sensor_res", 32)
        self.power_reward = cfg["env"].get("power_reward", False)
        self.power_coefficient = cfg["env"].get("power_coefficient", 0.0005)
        self.location_coefficient = cfg["env"].get("location_coefficient", 1)
        self.terrain_obs_type = cfg['env'].get("terrain_obs_type", "square")
        self.terrain_obs = cfg['env'].get("terrain_obs", False)
        self.terrain_obs_root = cfg['env'].get("terrain_obs_root", "pelvis")
        self.velocity_map = cfg["env"].get("velocity_map", False)
        if not (self.terrain_obs and self.terrain_obs_type == "square" and self.terrain_obs_root == "head" and
                self.sensor_res == 32 and self.sensor_extent == 2 and cfg['env'].get("use_center_height", False) and
                self.power_reward and self.location_coefficient == 1 and not self.velocity_map):
            raise NotImplementedError("emloco fused kernels cover pacer.yaml's terrain observation / reward settings")
        self.num_height_points = self.sensor_res * self.sensor_res
        self.num_center_height_points = 9
        self.center_height_points = self.init_center_height_points()
        self.height_meas_scale = 5
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type,
                         device_id=device_id, headless=headless)
        self.reward_raw = torch.zeros((self.num_envs, 2)).to(self.device)
        self._post_bufs = None
        n_wp = self._num_traj_samples if not flags.vru else 5
        self.waypoint_traj = torch.zeros(self.num_envs, n_wp, 3).to(self.device)     # :93-99
        self.init_pose = torch.zeros(self.num_envs, 24, 3).to(self.device)
        self.init_vel = torch.zeros(self.num_envs, 2).to(self.device)
        self._fused_reset = bool(cfg["env"].get("fused_reset", True)) and self._state_init == self.StateInit.Random \
            and not flags.vru and not flags.add_noise and not flags.fixed_path and not flags.slow
        self._reset_bufs = None
        return

    # ------------------------------------------------------------------ sizes
    def get_task_obs_size(self):
        obs_size = 0
        if (self._enable_task_obs):
            obs_size = 2 * self._num_traj_samples
            if self.terrain_obs:
                obs_size += self.num_height_points
        return obs_size

    def get_task_obs_size_detail(self):
        from collections import OrderedDict
        d = OrderedDict()
        if (self._enable_task_obs):
            d['traj'] = 2 * self._num_traj_samples
        if self.terrain_obs:
            d['heightmap'] = self.num_height_points
        return d

    def get_head_pose(self, env_ids=None):
        head_idx = self._body_names.index("Head")
        head_pose = torch.cat([self._rigid_body_pos[:, head_idx], self._rigid_body_rot[:, head_idx]], dim=1)
        return head_pose if env_ids is None else head_pose[env_ids]

    def init_center_height_points(self):                          # :633-648
        y = torch.tensor(np.linspace(-0.2, 0.2, 3), device=self.device)
        x = torch.tensor(np.linspace(-0.1, 0.1, 3), device=self.device)
        grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
        points = torch.zeros(self.num_envs, 9, 3, device=self.device)
        points[:, :, 0] = grid_x.flatten()
        points[:, :, 1] = grid_y.flatten()
        return points

    # ------------------------------------------------------------------ terrain (:817-881)
    def _create_ground_plane(self):
        self.create_training_ground()

    def create_training_ground(self):
        if flags.small_terrain:
            self.cfg["env"]["terrain"]['mapLength'] = 8
            self.cfg["env"]["terrain"]['mapWidth'] = 8
        self.terrain = Terrain(self.cfg["env"]["terrain"], num_robots=self.num_envs, device=self.device)
        tm_params = gymapi.TriangleMeshParams()
        tm_params.nb_vertices = self.terrain.vertices.shape[0]
        tm_params.nb_triangles = self.terrain.triangles.shape[0]
        tm_params.static_friction = self.cfg["env"]["terrain"]["staticFriction"]
        tm_params.dynamic_friction = self.cfg["env"]["terrain"]["dynamicFriction"]
        tm_params.restitution = self.cfg["env"]["terrain"]["restitution"]
        if not self.terrain.is_flat:      # extension: the grid behind the (slope-corrected) mesh, see gymapi.add_triangle_mesh
            tm_params.heightfield = dict(samples=self.terrain.height_field_raw, horizontal_scale=self.terrain.horizontal_scale,
                                         vertical_scale=self.terrain.vertical_scale)
        self.gym.add_triangle_mesh(self.sim, self.terrain.vertices.flatten(order='C'),
                                   self.terrain.triangles.flatten(order='C'), tm_params)
        self.height_samples = self.terrain.heightsamples.view(self.terrain.tot_rows, self.terrain.tot_cols).to(self.device)

    def get_center_heights(self, root_states, env_ids=None):      # :732-759 (host version, reset path)
        base_quat = root_states[:, 3:7]
        if env_ids is None:
            pts = self.center_height_points
        else:
            pts = self.center_height_points[env_ids.to(self.device)]
        points = tu.quat_apply_yaw(base_quat.repeat(1, self.num_center_height_points), pts) + (root_states[:, :3]).unsqueeze(1)
        heights = self.terrain.sample_height_points(points.clone(), env_ids=env_ids)
        return heights.view(root_states.shape[0], -1)

    # ------------------------------------------------------------------ fused step plumbing
    def _make_post_bufs(self):
        return self._post.make_bufs(
            n_env=self.num_envs, heightfield=self.height_samples.contiguous(), dt=self.dt,
            traj_dur=self._traj_gen.get_traj_duration(), sample_dt=self._traj_sample_timestep,
            hscale=self.terrain.horizontal_scale, vscale=self.terrain.vertical_scale, power_coef=self.power_coefficient,
            fail_dist=self._fail_dist, max_episode_length=self.max_episode_length,
            rb_state=self._rigid_body_state, dof_state=self._dof_state, dof_force=self.dof_force_tensor,
            contact_force=self._contact_forces, betas=self.humanoid_betas, traj_verts=self._traj_gen._verts,
            progress_buf=self.progress_buf, reset_buf=self.reset_buf, terminate_buf=self._terminate_buf,
            obs_buf=self.obs_buf, flip_obs_buf=self._flip_obs_buf, rew_buf=self.rew_buf, reward_raw=self.reward_raw,
            amp_obs_buf=self._amp_obs_buf)

    # ------------------------------------------------------------------ fused device reset (include/emloco_task.h)
    def _make_reset_bufs(self):
        import ctypes as C
        ml, tg, dev = self._motion_lib, self._traj_gen, self.device
        f = 0
        f |= L.RESET_RANDOM_HEADING if flags.random_heading else 0
        f |= L.RESET_INIT_HEADING if flags.init_heading else 0
        f |= L.RESET_HEADING_INVERSION if (flags.init_heading and flags.heading_inversion) else 0
        f |= L.RESET_ADJUST_ROOT_VEL if flags.adjust_root_vel else 0
        f |= L.RESET_REAL_PATH if flags.real_path else 0
        f |= L.RESET_FIXED_LOCATION if flags.fixed else 0
        real = None
        if flags.real_path:
            real = torch.from_numpy(tg.real_rows()).to(dev).contiguous()
        E = self.num_envs
        # the reset kernels write the sampled motion id / start time of a reset env straight into the task's per-env
        # bookkeeping (humanoid_amp.py:191-192 does the same by index assignment): alias, no merge pass afterwards
        assert self._sampled_motion_ids.dtype == torch.long and self._motion_start_times.dtype == torch.float32
        self._sampled_motion_ids = self._sampled_motion_ids.contiguous()
        self._motion_start_times = self._motion_start_times.contiguous()
        self._reset_motion_ids = self._sampled_motion_ids
        self._reset_motion_times = self._motion_start_times
        self._reset_ground_h = torch.zeros(E, device=dev)
        self._inverted_u8 = torch.zeros(E, dtype=torch.uint8, device=dev)
        keep = dict(real=real, vx=self.terrain.coord_x_scale.float().contiguous(), vy=self.terrain.coord_y_scale.float().contiguous(),
                    hf=self.height_samples.contiguous())
        self._reset_keep = keep
        p = lambda t: None if t is None else t.data_ptr()
        return L.ResetBufs(
            f, ml.num_motions(), 0 if real is None else int(real.shape[0]), int(keep["vx"].numel()), int(self._post.dof_subset.numel()),
            int(keep["hf"].shape[0]), int(keep["hf"].shape[1]), 50.0, 55.0, float(self.dt), 0.02,
            float(tg._dt), float(tg._dtheta_max), float(tg._speed_min), float(tg._speed_max), float(tg._accel_max),
            float(tg._sharp_turn_prob), float(tg._hybrid_init_prob), float(tg.get_traj_duration()), float(self._traj_sample_timestep),
            float(self.terrain.horizontal_scale), float(self.terrain.vertical_scale),
            p(ml.gts), p(ml.grs), p(ml.lrs), p(ml.gvs), p(ml.gavs), p(ml.dvs), p(ml._motion_lengths), p(ml._motion_dt),
            p(ml._motion_num_frames), p(ml.length_starts), p(real), p(keep["hf"]), p(keep["vx"]), p(keep["vy"]),
            p(self.humanoid_betas), p(self._post.key_bodies), p(self._post.dof_subset), p(tg._verts), p(self._inverted_u8),
            p(self.progress_buf), p(self.reset_buf), p(self._terminate_buf), p(self.waypoint_traj), p(self.init_pose), p(self.init_vel),
            p(self._amp_obs_buf), p(self._reset_motion_ids), p(self._reset_motion_times), p(self._reset_ground_h))

    def _fused_reset_envs(self, env_ids, rnd=None):
        """reset(env_ids) as three kernel launches + the indexed observation launch (replaces ~1 100 torch launches)."""
        import ctypes as C
        from ...sim import current_stream_handle
        if self._reset_bufs is None:
            self._reset_bufs = self._make_reset_bufs()
            if hasattr(self, "_amp_ring_to_bufs"):
                self._amp_ring_to_bufs()
        n = int(env_ids.numel())
        ids32 = self._humanoid_actor_ids[env_ids.to(self.device)].contiguous()
        if rnd is None:
            rnd = torch.rand((n, L.RESET_RND), device=self.device)
        rc = self._post.lib.emloco_task_reset(self.sim.native._h, C.byref(self._reset_bufs), C.c_void_p(ids32.data_ptr()), n,
                                              C.c_void_p(rnd.data_ptr()), current_stream_handle(torch.device(self.device)))
        L.check(rc, "emloco_task_reset")
        self._post.run(self._post_bufs if self._post_bufs is not None else self._ensure_post_bufs(), L.POST_OBS | L.POST_AMP_ROW, ids32)
        if flags.init_heading and flags.heading_inversion:
            self._traj_gen.inverted = self._inverted_u8.view(torch.bool)     # the kernels write 0 / 1 bytes: a view, no launch
        self.inverted = self._traj_gen.show_inverted()
        # _motion_start_times / _sampled_motion_ids: written in place by the reset kernels (aliased in _make_reset_bufs)

    def reset_done(self, rnd=None):
        """Reset every env whose reset_buf is set WITHOUT the host reading which ones: the reference's loop does
        `done_indices = dones.nonzero()` on the host each step (amp_continuous_value.py:46,74), which drains the launch
        queue; here the id list is compacted on the device (`emloco_task_compact_done`, padding entries -1 are skipped by
        the kernels) and the per-env bookkeeping uses masks.  Same result as `reset(reset_buf.nonzero())` with the same
        random rows.  Falls back to that when the fused reset path is not active."""
        import contextlib
        import ctypes as C
        from ...sim import current_stream_handle
        if not (getattr(self, "_fused_reset", False) and getattr(self, "_traj_gen", None) is not None):
            env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()
            if len(env_ids) > 0:
                self._reset_envs(env_ids)
            return
        if self._reset_bufs is None:
            self._reset_bufs = self._make_reset_bufs()
            if hasattr(self, "_amp_ring_to_bufs"):
                self._amp_ring_to_bufs()
        E = self.num_envs
        if getattr(self, "_done_ids", None) is None:
            self._done_ids = torch.full((E + 1,), -1, dtype=torch.int32, device=self.device)
        dev = torch.device(self.device)
        lib = self._post.lib
        side = None
        if self.fused_chain and dev.type == "cuda":
            # the whole chain in two launches on the caller's stream: compaction (+ flag snapshot + the next step's dispatch order),
            # then ONE launch for the reset chain of the finished envs, their AMP history, their observations AND the observation /
            # AMP pass post_physics_step deferred for the envs that did not finish (humanoid.py: fused_chain)
            if getattr(self, "_rs_skip", None) is None:
                self._rs_skip = torch.zeros(E, dtype=torch.int64, device=dev)
            st = current_stream_handle(dev)
            L.check(lib.emloco_task_compact_done_order(self.sim.native._h, C.c_void_p(self.reset_buf.data_ptr()), E,
                                                       C.c_void_p(self._done_ids.data_ptr()), C.c_void_p(self._rs_skip.data_ptr()), st),
                    "emloco_task_compact_done_order")
            live_mode, self._obs_deferred = self._obs_deferred, 0
            if rnd is None:
                if getattr(self, "_rnd_ws", None) is None:
                    self._rnd_ws = torch.empty((E, L.RESET_RND), device=self.device)
                    self._rnd_calls = 0
                    self._rnd_seed0 = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
                self._rnd_calls += 1
                seed = (self._rnd_seed0 * 0x9E3779B97F4A7C15 + self._rnd_calls) & 0xFFFFFFFFFFFFFFFF
                # the pool of pre-drawn episodes (include/emloco_task.h: EmlocoResetPool): this launch draws the first K entries of the
                # NEXT call beside its other work, and copies the ones the previous launch drew for THIS call's seed
                K = min(int(os.environ.get("EMLOCO_RESET_POOL", "256")), E)
                pool = None
                if K > 0:
                    if getattr(self, "_pool", None) is None or self._pool[0].shape[0] != K:
                        self._pool = [torch.zeros((K, L.POOL_FLOATS), device=dev) for _ in range(2)]
                        self._pool_tag = [torch.zeros(K, dtype=torch.int64, device=dev) for _ in range(2)]
                        self._pool_flip = 0
                    c, nx = self._pool_flip, self._pool_flip ^ 1
                    self._pool_flip = nx
                    next_seed = (self._rnd_seed0 * 0x9E3779B97F4A7C15 + self._rnd_calls + 1) & 0xFFFFFFFFFFFFFFFF
                    pool = L.ResetPool(K, self._pool[c].data_ptr(), self._pool_tag[c].data_ptr(), self._pool[nx].data_ptr(),
                                       self._pool_tag[nx].data_ptr(), next_seed)
            else:
                seed, pool = 0, None
            post_bufs = self._post_bufs if self._post_bufs is not None else self._ensure_post_bufs()
            L.check(lib.emloco_task_reset_obs_pooled(self.sim.native._h, C.byref(self._reset_bufs), C.byref(post_bufs), int(live_mode),
                                                     C.c_void_p(self._rs_skip.data_ptr()), C.c_void_p(self._done_ids.data_ptr()), E, C.c_uint64(seed),
                                                     None if rnd is not None else C.c_void_p(self._rnd_ws.data_ptr()),
                                                     None if rnd is None else C.c_void_p(rnd.data_ptr()),
                                                     None if pool is None else C.byref(pool), st), "emloco_task_reset_obs_pooled")
            if flags.init_heading and flags.heading_inversion:
                self._traj_gen.inverted = self._inverted_u8.view(torch.bool)
            self.inverted = self._traj_gen.show_inverted()
            return
        if self.overlap_reset and dev.type == "cuda":
            # the reset chain runs beside the step of the live envs (see overlap_reset below); the flags are snapshot by the
            # compaction because the reset kernels clear them
            if getattr(self, "_hp_stream", None) is None:
                if getattr(self, "_rs_skip", None) is None:
                    self._rs_skip = torch.zeros(E, dtype=torch.int64, device=dev)
                self._hp_stream = torch.cuda.Stream(device=dev, priority=-1)
                self._ev_rs_fork, self._ev_rs_reset, self._ev_rs_pd, self._ev_rs_big = (torch.cuda.Event() for _ in range(4))
            if self.overlap_reset_mode == "hp" and getattr(self, "_obs_stream", None) is None:
                self._make_obs_stream()
            self.wait_reset()
            main = torch.cuda.current_stream(dev)
            L.check(lib.emloco_task_compact_done_snapshot(C.c_void_p(self.reset_buf.data_ptr()), E, C.c_void_p(self._done_ids.data_ptr()),
                                                          C.c_void_p(self._rs_skip.data_ptr()), current_stream_handle(dev)),
                    "emloco_task_compact_done_snapshot")
            self._ev_rs_fork.record(main)
            side = self._obs_stream if self.overlap_reset_mode == "hp" else self._hp_stream
            side.wait_event(self._ev_rs_fork)
        else:
            L.check(lib.emloco_task_compact_done(C.c_void_p(self.reset_buf.data_ptr()), E, C.c_void_p(self._done_ids.data_ptr()),
                                                 current_stream_handle(dev)), "emloco_task_compact_done")
        # sequential order with an observation stream: the AMP history back-fill (reads only what the reset sampled) leaves the
        # caller's chain for that stream; wait_obs() covers it
        amp_aside = (side is None and self.overlap_obs and dev.type == "cuda" and getattr(self, "_obs_stream", None) is not None
                     and os.environ.get("EMLOCO_AMP_ASIDE", "1") != "0")
        bufs = self._reset_bufs
        if amp_aside:
            if getattr(self, "_reset_bufs_noamp", None) is None:
                self._reset_bufs_noamp = type(bufs).from_buffer_copy(bufs)
                self._reset_bufs_noamp.flags |= L.RESET_NO_AMP_HISTORY
                self._ev_rs_state = torch.cuda.Event()
            self._reset_bufs_noamp.amp_ring = bufs.amp_ring
            bufs = self._reset_bufs_noamp
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            st = current_stream_handle(dev)
            if rnd is None:
                # random rows made on the device for the finished envs only (row i serves the i-th finished env, ascending id),
                # from a per-call seed: torch's seed (set by run.py / set_seed) + a call counter
                if getattr(self, "_rnd_ws", None) is None:
                    self._rnd_ws = torch.empty((E, L.RESET_RND), device=self.device)
                    self._rnd_calls = 0
                    self._rnd_seed0 = int(torch.initial_seed()) & 0xFFFFFFFFFFFF
                self._rnd_calls += 1
                L.check(lib.emloco_task_reset_seeded(self.sim.native._h, C.byref(bufs), C.c_void_p(self._done_ids.data_ptr()), E,
                                                     C.c_uint64((self._rnd_seed0 * 0x9E3779B97F4A7C15 + self._rnd_calls) & 0xFFFFFFFFFFFFFFFF),
                                                     C.c_void_p(self._rnd_ws.data_ptr()), st), "emloco_task_reset_seeded")
            else:
                L.check(lib.emloco_task_reset(self.sim.native._h, C.byref(bufs), C.c_void_p(self._done_ids.data_ptr()), E,
                                              C.c_void_p(rnd.data_ptr()), st), "emloco_task_reset")
            if amp_aside:
                self._ev_rs_state.record(torch.cuda.current_stream(dev))
                self._obs_stream.wait_event(self._ev_rs_state)
                with torch.cuda.stream(self._obs_stream):
                    L.check(lib.emloco_task_reset_amp_history(C.byref(bufs), C.c_void_p(self._done_ids.data_ptr()), E,
                                                              current_stream_handle(dev)), "emloco_task_reset_amp_history")
                    self._ev_obs.record(self._obs_stream)
                self._obs_pending = True
            self._post.run(self._post_bufs if self._post_bufs is not None else self._ensure_post_bufs(), L.POST_OBS | L.POST_AMP_ROW,
                           self._done_ids[:E])
            if side is not None:
                self._ev_rs_reset.record(side)
                self._rs_pending = self._rs_unjoined = True
        if flags.init_heading and flags.heading_inversion:
            self._traj_gen.inverted = self._inverted_u8.view(torch.bool)     # the kernels write 0 / 1 bytes: a view, no launch
        self.inverted = self._traj_gen.show_inverted()      # motion ids / start times: written in place by the reset kernels

    # Opt-in (set by a rollout loop that calls wait_reset() before it reads anything reset_done() wrote -- the observations of
    # the reset envs, init_pose / init_vel, the trajectory -- between reset_done() and step()): envs are independent
    # (humanoid.py:838-841), so `reset_done(); step(a)` is issued as two chains.  The rigid-body launch of the envs that did
    # not finish (emloco_sim_step_subset with the flag snapshot) goes to a high-priority stream; the reset chain of the
    # finished ones and their observations run beside it on the (high-priority) observation stream; their step (the same
    # kernel over the compacted id list: one code object, the two launches share the instruction cache) follows on the
    # caller's stream at normal priority, so its workgroups take the wave slots the big launch leaves over at the end of its
    # first round instead of displacing workgroups of it; step() joins before its post-physics launch.  Every env sees
    # exactly the launches it would see in the sequential order: identical results (tests/test_gpu_env.py).  Four streams in
    # all with the LocoVal fit's: HIP serves a process with four hardware queues.
    overlap_reset = False
    # "side": the reset chain and the reset envs' step on one high-priority stream, the big launch on the caller's (the
    # arrangement above with the roles of the two launches swapped: the small launch competes for wave slots at the end of
    # the big launch's first round); "hp": as described above
    overlap_reset_mode = os.environ.get("EMLOCO_OVERLAP_MODE", "side")

    def wait_reset(self):
        """Make the caller's stream wait for the reset chain of the last reset_done() (no-op without overlap_reset)."""
        if getattr(self, "_rs_unjoined", False):
            torch.cuda.current_stream(self.device).wait_event(self._ev_rs_reset)
            self._rs_unjoined = False

    def _physics_step(self):
        if not getattr(self, "_rs_pending", False):
            return super()._physics_step()
        self._rs_pending = False
        if self.paused or not self.enable_viewer_sync:
            self.wait_reset()
            return
        dev = torch.device(self.device)
        main, hp, E = torch.cuda.current_stream(dev), self._hp_stream, self.num_envs
        self._ev_rs_pd.record(main)                              # the PD targets of all envs are in place
        if self.overlap_reset_mode == "hp":
            hp.wait_event(self._ev_rs_pd)
            with torch.cuda.stream(hp):
                self.gym.simulate_n_subset(self.sim, self.control_freq_inv, skip=self._rs_skip)
                self._ev_rs_big.record(hp)
            self.wait_reset()                                    # the reset envs' state and observations are in place
            self.gym.simulate_n_subset(self.sim, self.control_freq_inv, ids=self._done_ids[:E], count=False)
        else:
            self.gym.simulate_n_subset(self.sim, self.control_freq_inv, skip=self._rs_skip)
            hp.wait_event(self._ev_rs_pd)
            with torch.cuda.stream(hp):                          # behind the reset chain on its stream
                self.gym.simulate_n_subset(self.sim, self.control_freq_inv, ids=self._done_ids[:E], count=False)
                self._ev_rs_big.record(hp)
            self._rs_unjoined = False                            # the join below covers the reset chain
        main.wait_event(self._ev_rs_big)

    def _ensure_post_bufs(self):
        self._post_bufs = self._make_post_bufs()
        if hasattr(self, "_amp_ring_to_bufs"):
            self._amp_ring_to_bufs()
        return self._post_bufs

    def _reset_envs(self, env_ids):
        if getattr(self, "_fused_reset", False) and len(env_ids) > 0 and getattr(self, "_traj_gen", None) is not None:
            self._fused_reset_envs(env_ids)
        else:
            super()._reset_envs(env_ids)

    # ------------------------------------------------------------------ reset (:493-631)
    def _reset_task(self, env_ids):
        if len(env_ids) > 0:
            root_pos = self._humanoid_root_states[env_ids, 0:3]
            root_vel = self._humanoid_root_states[env_ids, 7:10]
            motion_ids = getattr(self, "_reset_ref_motion_ids", None)
            motion_times = getattr(self, "_reset_ref_motion_times", None)
            self._traj_gen.reset(env_ids, root_pos, root_vel, motion_ids, motion_times)
            self.inverted = self._traj_gen.show_inverted()
            self.waypoint_traj[env_ids] = self._fetch_traj_samples(env_ids)
            self.init_pose[env_ids] = self._rigid_body_pos[env_ids]
            self.init_vel[env_ids] = root_vel[:, :2]
        return

    def _sample_ref_state(self, env_ids, vel_min=1, vel_range=0.5):   # :526-573
        out = super()._sample_ref_state(env_ids)
        motion_ids, motion_times, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot = out
        num_envs = env_ids.shape[0]
        if flags.random_heading:
            yaw = np.pi * (2 * np.random.random([num_envs]) - 1.0)
            h = torch.from_numpy(np.stack([np.zeros(num_envs), np.zeros(num_envs), np.sin(yaw / 2), np.cos(yaw / 2)], -1)).float().to(self.device)
            h_rep = h[:, None].repeat(1, 24, 1)
            root_rot = tu.quat_mul(h, root_rot).clone()
            rb_pos = tu.quat_apply(h_rep, rb_pos - root_pos[:, None, :]).clone()
            key_pos = tu.quat_apply(h_rep[:, :4, :], (key_pos - root_pos[:, None, :])).clone()
            rb_rot = tu.quat_mul(h_rep, rb_rot).clone()
            root_ang_vel = tu.quat_apply(h, root_ang_vel).clone()
            curr_heading = tu.calc_heading_quat(root_rot)
            root_vel[:, 0] = (torch.rand([num_envs]) * vel_range + vel_min).to(self.device)
            root_vel = tu.quat_apply(curr_heading, root_vel).clone()
        return motion_ids, motion_times, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot

    def _reset_ref_state_init(self, env_ids):
        motion_ids, motion_times, root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos, rb_pos, rb_rot = \
            self._sample_ref_state(env_ids)
        new_root_xy = self.terrain.sample_valid_locations(self.num_envs, env_ids)
        if flags.fixed:
            new_root_xy[:, 0], new_root_xy[:, 1] = 50, 55          # :588
        root_pos[:, 0:2] = new_root_xy
        root_states = torch.cat([root_pos, root_rot], dim=1)
        center_height = self.get_center_heights(root_states, env_ids=env_ids).mean(dim=-1)
        self._reset_ground_height = center_height
        env_ids = env_ids.to(self.device)
        self._set_env_state(env_ids=env_ids, root_pos=root_pos, root_rot=root_rot, dof_pos=dof_pos, root_vel=root_vel,
                            root_ang_vel=root_ang_vel, dof_vel=dof_vel, rigid_body_pos=rb_pos, rigid_body_rot=rb_rot)
        self._reset_ref_env_ids = env_ids
        self._reset_ref_motion_ids = motion_ids
        self._reset_ref_motion_times = motion_times
        self._motion_start_times[env_ids] = motion_times
        self._sampled_motion_ids[env_ids] = motion_ids
        return


def poles_terrain(terrain, difficulty=1):                           # :937-993
    """Obstacle course of thin tall shapes: up to 10m disks always, then with probability p each a batch of 3m curves,
    m polygons and 5m ellipses (m = width // 80), every shape kept with probability p and raised by U[200, 500)
    vertical units; p = [0.9, 0.4, 0.5, 0.5] * (0.5 + 0.5 * difficulty).  Rasterisation: utils/draw_utils.py."""
    img = np.zeros((terrain.width, terrain.length), dtype=int)
    probs = np.array([0.9, 0.4, 0.5, 0.5]) * (0.5 * difficulty + 0.5)
    low, high = 200, 500
    m = int(terrain.width // 80)
    batches = [(10 * m, lambda: draw_disk(img_size=terrain.width, max_r=7)),
               (3 * m, lambda: draw_curve(img_size=terrain.width)),
               (1 * m, lambda: draw_polygon(img_size=terrain.width, max_sides=5)),
               (5 * m, lambda: draw_ellipse(img_size=terrain.width, max_size=5))]
    for k, (count, draw) in enumerate(batches):
        p = probs[k]
        if k > 0 and not np.random.binomial(1, p):                  # the disks have no batch-level gate
            continue
        for _ in range(count):
            if np.random.binomial(1, p):
                img += draw() * int(np.random.uniform(low, high))
    terrain.height_field_raw[0:terrain.width, 0:terrain.length] = img
    return terrain


class Terrain:
    """Height-field terrain (mirror of humanoid_pedestrain_terrain.py:1135-1463): int16 height samples at 0.1 m /
    0.005 m resolution with a 50 m border, one sub-terrain per (level, terrain) cell drawn from `terrainProportions`
    ([pyramid slope, slope + noise, stairs down, stairs up, discrete obstacles, stepping stones, poles, flat]), a
    walkable mask, and sampling helpers.  Random draws follow the reference's `np.random` call order, so a seeded map
    equals the reference's (tests/golden/terrain_layout.npz)."""

    def __init__(self, cfg, num_robots, device) -> None:
        self.type = cfg["terrainType"]
        self.device = device
        if self.type in ["none", 'plane']:
            raise NotImplementedError("terrainType must be trimesh for the terrain observations")
        self.horizontal_scale = 0.1
        self.vertical_scale = 0.005
        self.border_size = 50
        self.env_length = cfg["mapLength"]
        self.env_width = cfg["mapWidth"]
        self.proportions = [np.sum(cfg["terrainProportions"][:i + 1]) for i in range(len(cfg["terrainProportions"]))]
        self.env_rows = cfg["numLevels"]
        self.env_cols = cfg["numTerrains"]
        self.num_maps = self.env_rows * self.env_cols
        self.env_origins = np.zeros((self.env_rows, self.env_cols, 3))
        self.width_per_env_pixels = int(self.env_width / self.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / self.horizontal_scale)
        self.border = int(self.border_size / self.horizontal_scale)
        self.tot_cols = int(self.env_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(self.env_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self.walkable_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg["curriculum"]:
            self.curiculum(num_robots, num_terrains=self.env_cols, num_levels=self.env_rows)
        else:
            self.randomized_terrain()
        self.heightsamples = torch.from_numpy(self.height_field_raw).to(self.device)
        self.walkable_field = torch.from_numpy(self.walkable_field_raw).to(self.device)
        self.is_flat = not self.height_field_raw.any()
        if self.is_flat:
            # collision mesh of a flat field: two triangles (a full-resolution mesh of a plane adds nothing)
            ex, ey = (self.tot_rows - 1) * self.horizontal_scale, (self.tot_cols - 1) * self.horizontal_scale
            self.vertices = np.array([[0, 0, 0], [ex, 0, 0], [0, ey, 0], [ex, ey, 0]], dtype=np.float32)
            self.triangles = np.array([[0, 3, 1], [0, 2, 3]], dtype=np.uint32)
        else:
            self.vertices, self.triangles = convert_heightfield_to_trimesh(
                self.height_field_raw, self.horizontal_scale, self.vertical_scale, cfg["slopeTreshold"])
        self.sample_extent_x = int((self.tot_rows - self.border * 2) * self.horizontal_scale)
        self.sample_extent_y = int((self.tot_cols - self.border * 2) * self.horizontal_scale)
        coord_x, coord_y = torch.where(self.walkable_field == 0)
        cx, cy = coord_x * self.horizontal_scale, coord_y * self.horizontal_scale
        b = self.border * self.horizontal_scale
        sub = torch.logical_and(torch.logical_and(cy < cy.max() - b, cx < cx.max() - b),
                                torch.logical_and(cy > cy.min() + b, cx > cx.min() + b))
        self.coord_x_scale = cx[sub]
        self.coord_y_scale = cy[sub]
        self.num_samples = self.coord_x_scale.shape[0]

    # ------------------------------------------------------------------ sub-terrain layout (:1299-1463)
    def _place(self, i, j, choice, difficulty):
        """One (level i, terrain j) cell: pick the generator by `choice` against the cumulative proportions, scale it
        by `difficulty`, write the patch and its spawn origin (highest point of the central 2 m x 2 m)."""
        P = self.proportions
        patch = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                           vertical_scale=self.vertical_scale, horizontal_scale=self.horizontal_scale)
        slope = difficulty * 0.7
        step_height = 0.05 + 0.175 * difficulty
        x0 = self.border + i * self.length_per_env_pixels
        y0 = self.border + j * self.width_per_env_pixels
        x1, y1 = x0 + self.length_per_env_pixels, y0 + self.width_per_env_pixels
        if choice < P[0]:
            pyramid_sloped_terrain(patch, slope=-slope if choice < 0.05 else slope, platform_size=3.)
        elif choice < P[1]:
            pyramid_sloped_terrain(patch, slope=-slope if choice < 0.15 else slope, platform_size=3.)
            random_uniform_terrain(patch, min_height=-0.1, max_height=0.1, step=0.025, downsampled_scale=0.2)
        elif choice < P[3]:
            pyramid_stairs_terrain(patch, step_width=0.31, step_height=-step_height if choice < P[2] else step_height,
                                   platform_size=3.)
        elif choice < P[4]:
            discrete_obstacles_terrain(patch, 0.025 + difficulty * 0.15, 1., 2., 40, platform_size=3.)
        elif choice < P[5]:
            stepping_stones_terrain(patch, stone_size=2 - 1.8 * difficulty, stone_distance=0.1, max_height=0., platform_size=3.)
        elif choice < P[6]:
            poles_terrain(terrain=patch, difficulty=difficulty)
            self.walkable_field_raw[x0:x1, y0:y1] = (patch.height_field_raw != 0)
        # else (choice < P[7]): plain walking terrain
        self.height_field_raw[x0:x1, y0:y1] = patch.height_field_raw
        hs = self.horizontal_scale
        cx0, cx1 = int((self.env_length / 2. - 1) / hs), int((self.env_length / 2. + 1) / hs)
        cy0, cy1 = int((self.env_width / 2. - 1) / hs), int((self.env_width / 2. + 1) / hs)
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width,
                                  np.max(patch.height_field_raw[cx0:cx1, cy0:cy1]) * self.vertical_scale]

    def randomized_terrain(self):                                   # :1299-1372: type and difficulty drawn per cell
        for k in range(self.num_maps):
            i, j = np.unravel_index(k, (self.env_rows, self.env_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.uniform(0.1, 1)
            self._place(i, j, choice, difficulty)
        self.walkable_field_raw = ndimage.binary_dilation(self.walkable_field_raw, iterations=3).astype(int)

    def curiculum(self, num_robots, num_terrains, num_levels):      # :1374-1461: type by column, difficulty by level
        for j in range(num_terrains):
            for i in range(num_levels):
                self._place(i, j, j / num_terrains, i / num_levels)
        self.walkable_field_raw = ndimage.binary_dilation(self.walkable_field_raw, iterations=3).astype(int)

    def sample_valid_locations(self, max_num_envs, env_ids, group_num_people=16, sample_groups=False):   # :1196-1210
        idxes = np.random.randint(0, self.num_samples, size=env_ids.shape[0])
        return torch.stack([self.coord_x_scale[idxes], self.coord_y_scale[idxes]], dim=-1)

    def world_points_to_map(self, points):                         # :1212-1218
        points = (points / self.horizontal_scale).long()
        px = torch.clip(points[:, :, 0].view(-1), 0, self.heightsamples.shape[0] - 2)
        py = torch.clip(points[:, :, 1].view(-1), 0, self.heightsamples.shape[1] - 2)
        return px, py

    def sample_height_points(self, points, root_states=None, root_points=None, env_ids=None, **kw):   # :1282-1288
        px, py = self.world_points_to_map(points)
        heights = torch.min(self.heightsamples[px, py], self.heightsamples[px + 1, py + 1])
        return heights * self.vertical_scale
