"""TrajGenerator: the per-env target polyline (101 vertices) the humanoid follows.

Host-side mirror of pacer/pacer/env/util/traj_generator.py (class TrajGenerator :19-296): same constructor,
`reset`, `calc_pos`, `show_inverted`, `get_*` methods and the same order of random draws, so a seeded CPU run
reproduces the reference's vertices (pinned by tests/test_host_logic.py against tests/golden/traj_reset_*.npz).
`reset` runs once per episode; the per-step sampling (`calc_pos`, :278-296) is done on the device by the fused
post-physics kernel and is kept here for host callers.

Real-world paths (`flags.real_path`): `traj_data` is a list of dicts {id: {'traj': (>=101,3) array, 'pose': ...}}
in the format load_jta_traj.py:109-114 writes; pass it in (or a pickle path) instead of the hard-coded files.
"""
import random

import numpy as np
import torch


class TrajGenerator():
    def __init__(self, num_envs, episode_dur, num_verts, device, dtheta_max, speed_min, speed_max, accel_max,
                 sharp_turn_prob, motion_lib=None, hybridInitProb=0.5, flags=None, traj_data=None):
        self._device = device
        self._dt = episode_dur / (num_verts - 1)
        self._dtheta_max = dtheta_max
        self._speed_min = speed_min
        self._speed_max = speed_max
        self._accel_max = accel_max
        self._sharp_turn_prob = sharp_turn_prob
        self._motion_lib = motion_lib
        self._hybrid_init_prob = hybridInitProb
        self._flags = flags
        self.inverted = torch.zeros(num_envs, dtype=torch.bool, device=self._device)
        self._verts_flat = torch.zeros((num_envs * num_verts, 3), dtype=torch.float32, device=self._device)
        self._verts = self._verts_flat.view((num_envs, num_verts, 3))
        self.traj_data = []
        if self._flags is not None and self._flags.real_path:
            if traj_data is None:
                raise ValueError("flags.real_path needs traj_data (list of {id: {'traj': array}} dicts or pickle paths)")
            for d in traj_data:
                if isinstance(d, str):
                    import joblib
                    d = joblib.load(d)
                self.traj_data.append(d)
        self.heading = torch.zeros(num_envs, 1)

    # ------------------------------------------------------------------ random draws, in the reference's order (:63-78)
    def _draw(self, n, num_verts):
        dev = self._device
        return dict(
            r_dtheta=torch.rand([n, num_verts - 1], device=dev),
            r_dtheta_sharp=torch.rand([n, num_verts - 1], device=dev),
            bern_sharp=torch.bernoulli(self._sharp_turn_prob * torch.ones([n, num_verts - 1], device=dev)),
            r_heading=torch.rand([n], device=dev),
            r_dspeed=torch.rand([n, num_verts - 1], device=dev),
            r_speed0=torch.rand([n], device=dev),
        )

    def _polyline(self, d, init_pos, root_vel):
        """Random-walk polyline from the draws (:63-118)."""
        dtheta = 2 * d["r_dtheta"] - 1.0
        dtheta *= self._dtheta_max * self._dt
        dtheta_sharp = np.pi * (2 * d["r_dtheta_sharp"] - 1.0)
        sharp_mask = d["bern_sharp"] == 1.0
        dtheta[sharp_mask] = dtheta_sharp[sharp_mask]
        dtheta[:, 0] = np.pi * (2 * d["r_heading"] - 1.0)
        dspeed = 2 * d["r_dspeed"] - 1.0
        dspeed *= self._accel_max * self._dt
        dspeed[:, 0] = (self._speed_max - self._speed_min) * d["r_speed0"] + self._speed_min
        speed = torch.zeros_like(dspeed)
        speed[:, 0] = dspeed[:, 0]
        for i in range(1, dspeed.shape[-1]):
            speed[:, i] = torch.clip(speed[:, i - 1] + dspeed[:, i], self._speed_min, self._speed_max)
        f = self._flags
        if f.fixed_path:
            dtheta[:, :] = 0
            if len(dtheta) > 1:
                dtheta[1, 0] = -np.pi
            speed[:] = (self._speed_min + self._speed_max) / 2
        if f.slow:
            speed[:] = speed / 4
        if f.adjust_root_vel:
            root_speed = torch.norm(root_vel[:, :2], dim=-1)
            speed_ratio = root_speed / speed[:, 0]
            speed = torch.clip(speed_ratio.unsqueeze(-1) * speed, self._speed_min, self._speed_max)
        dtheta = torch.cumsum(dtheta, dim=-1)
        seg_len = speed * self._dt
        dpos = torch.stack([torch.cos(dtheta), -torch.sin(dtheta), torch.zeros_like(dtheta)], dim=-1)
        dpos *= seg_len.unsqueeze(-1)
        dpos[..., 0, 0:2] += init_pos[..., 0:2]
        return torch.cumsum(dpos, dim=-2)

    def reset(self, env_ids, init_pos, root_vel=None, motion_ids=None, motion_times=None, draws=None):
        n = len(env_ids)
        if n == 0:
            return
        f = self._flags
        num_verts = self.get_num_verts()
        d = draws if draws is not None else self._draw(n, num_verts)
        vert_pos = self._polyline(d, init_pos, root_vel)
        self._verts[env_ids, 0, 0:2] = init_pos[..., 0:2]
        self._verts[env_ids, 1:] = vert_pos

        if f.real_path:                                                             # :121-160
            real_prob = d.get("r_real", None)
            if real_prob is None:
                real_prob = torch.rand(n, device=self._device)
            real_mask = real_prob > self._hybrid_init_prob
            real_num = int(torch.sum(real_mask))
            sizes = [len(t) for t in self.traj_data]
            rids = d.get("real_rids", None)                 # explicit sample (tests replay the reference's / the device's)
            if rids is None:
                rids = random.sample(range(sum(sizes)), real_num)
            rids = [int(r) for r in rids]
            assert len(rids) == real_num
            traj = torch.zeros([real_num, num_verts, 3], device=self._device)
            for i, rid in enumerate(rids):
                k = 0
                while rid >= sizes[k]:
                    rid -= sizes[k]
                    k += 1
                src = self.traj_data[k]
                item = src[rid] if not isinstance(src, dict) or rid in src else src[list(src.keys())[rid]]
                traj[i] = torch.as_tensor(np.asarray(item["traj"]), dtype=torch.float32)[:num_verts]
            traj[..., 0:2] = traj[..., 0:2] - traj[..., 0, 0:2].unsqueeze(1)
            if f.adjust_root_vel:
                init_speed = torch.clamp(torch.norm(traj[:, 1] - traj[:, 0], dim=-1), min=self._speed_min * self._dt)
                root_speed = torch.norm(root_vel[real_mask, :2].clone(), dim=-1)
                ratio = root_speed.div(init_speed) * self._dt
                traj[..., 0:2] = ratio.view(-1, 1, 1) * traj[..., 0:2]
            traj[..., 0:2] += init_pos[real_mask, 0:2].unsqueeze(1)
            self._verts[env_ids[real_mask]] = traj

        if f.init_heading:                                                          # :176-235
            verts = self._verts[env_ids].clone()
            dinit = verts[:, 1, :2] - verts[:, 0, :2]
            root_mag = torch.sqrt(torch.sum(root_vel ** 2, dim=1))
            dinit_mag = torch.sqrt(torch.sum(dinit ** 2, dim=1))
            root_rot = torch.where(root_mag > 0, torch.atan2(root_vel[..., 1], root_vel[..., 0]), torch.zeros_like(root_vel[..., 0]))
            init_heading = torch.where(dinit_mag > 0, torch.atan2(dinit[..., 1], dinit[..., 0]), torch.zeros_like(dinit[..., 0]))
            rot_diff = init_heading - root_rot
            if f.heading_inversion:
                r_inv = d.get("r_inversion", None)
                if r_inv is None:
                    r_inv = torch.rand(n, device=self._device)
                inv = r_inv > 0.5
                self.inverted[env_ids[inv]] = True
                self.inverted[env_ids[~inv]] = False
                rot_diff[inv] = init_heading[inv] - root_rot[inv] + np.pi
            origin = verts[:, 0, 0:2].clone().unsqueeze(1).expand(-1, num_verts, 2)
            verts[:, :, 0:2] -= origin
            c, s = torch.cos(rot_diff), torch.sin(rot_diff)
            R = torch.stack([c, -s, s, c], dim=-1).view(-1, 2, 2)
            verts[:, :, 0:2] = torch.bmm(verts[:, :, 0:2].clone(), R)
            verts[:, :, 0:2] += origin
            self._verts[env_ids] = verts
        if f.add_noise:
            self._verts[env_ids] += torch.randn_like(self._verts[env_ids]) * 0.5

    def real_rows(self):
        """The real paths as one (n_real, num_verts, 3) float32 table, datasets concatenated in order (row = the id
        `random.sample(range(jta_num + jrdb_num), ...)` indexes, traj_generator.py:128-143)."""
        rows = []
        for src in self.traj_data:
            vals = src.values() if isinstance(src, dict) else src
            rows += [np.asarray(v["traj"], np.float32)[:self.get_num_verts()] for v in vals]
        return np.stack(rows) if rows else np.zeros((0, self.get_num_verts(), 3), np.float32)

    def reset_on_device(self, env_ids, init_pos, root_vel, rnd=None, real_pick=None, real_pick_key=0):
        """`reset` as one kernel launch (emloco_task_traj_reset, reset_kernels.hip): random rows `rnd` [n][RESET_RND]
        uniform in [0, 1) laid out as include/emloco_task.h describes (drawn with torch.rand when omitted), real-path rows
        from `real_pick` (int32 [n]) or the keyed permutation.  Needs the gfx950 library and CUDA tensors."""
        import ctypes as C
        from ... import _lib as L
        from ...sim import current_stream_handle
        lib = L.require_device()
        n = len(env_ids)
        if n == 0:
            return
        f = self._flags
        dev = self._verts.device
        if rnd is None:
            rnd = torch.rand((n, L.RESET_RND), device=dev)
        if getattr(self, "_real_dev", None) is None and f.real_path:
            self._real_dev = torch.from_numpy(self.real_rows()).to(dev).contiguous()
        if getattr(self, "_inverted_u8", None) is None:
            self._inverted_u8 = torch.zeros(self.get_num_envs(), dtype=torch.uint8, device=dev)
        self._inverted_u8.copy_(self.inverted.to(torch.uint8))
        b = L.ResetBufs()
        fl = (L.RESET_INIT_HEADING if f.init_heading else 0) | (L.RESET_ADJUST_ROOT_VEL if f.adjust_root_vel else 0)
        fl |= L.RESET_HEADING_INVERSION if (f.init_heading and f.heading_inversion) else 0
        fl |= L.RESET_REAL_PATH if f.real_path else 0
        if f.fixed_path or f.slow or f.add_noise or getattr(f, "pred_path", False):
            raise NotImplementedError("reset_on_device: fixed_path / slow / add_noise / pred_path stay on the host path (reset)")
        b.flags, b.n_real = fl, (int(self._real_dev.shape[0]) if f.real_path else 0)
        b.vert_dt, b.dtheta_max, b.speed_min, b.speed_max = self._dt, self._dtheta_max, self._speed_min, self._speed_max
        b.accel_max, b.sharp_prob, b.hybrid_prob = self._accel_max, self._sharp_turn_prob, self._hybrid_init_prob
        b.real_traj = self._real_dev.data_ptr() if f.real_path else None
        b.traj_verts, b.inverted = self._verts.data_ptr(), self._inverted_u8.data_ptr()
        keep = [rnd.contiguous(), env_ids.to(dev).to(torch.int32).contiguous(), init_pos[:, :3].float().contiguous(),
                root_vel[:, :3].float().contiguous()]
        if real_pick is not None:
            keep.append(real_pick.to(dev).to(torch.int32).contiguous())
            b.real_pick = keep[-1].data_ptr()
        b.real_pick_key = int(real_pick_key) & 0xFFFFFFFF
        L.check(lib.emloco_task_traj_reset(C.byref(b), keep[1].data_ptr(), n, keep[0].data_ptr(), keep[2].data_ptr(),
                                           keep[3].data_ptr(), current_stream_handle(dev)), "emloco_task_traj_reset")
        if f.init_heading and f.heading_inversion:
            self.inverted = self._inverted_u8.bool()

    def show_inverted(self):
        return self.inverted

    def get_num_verts(self):
        return self._verts.shape[1]

    def get_num_segs(self):
        return self.get_num_verts() - 1

    def get_num_envs(self):
        return self._verts.shape[0]

    def get_traj_duration(self):
        return self.get_num_verts() * self._dt      # (sic) the reference uses num_verts, not num_segs (:270-273)

    def get_traj_verts(self, traj_id):
        return self._verts[traj_id]

    def calc_pos(self, traj_ids, times):                                             # :278-296
        traj_dur = self.get_traj_duration()
        num_verts = self.get_num_verts()
        num_segs = self.get_num_segs()
        phase = torch.clip(times / traj_dur, 0.0, 1.0)
        seg_idx = phase * num_segs
        seg_id0 = torch.floor(seg_idx).long()
        seg_id1 = torch.ceil(seg_idx).long()
        lerp = (seg_idx - seg_id0).unsqueeze(-1)
        pos0 = self._verts_flat[traj_ids * num_verts + seg_id0]
        pos1 = self._verts_flat[traj_ids * num_verts + seg_id1]
        return (1.0 - lerp) * pos0 + lerp * pos1
