"""Synthetic AMASS-shaped motion library.

The reference's MotionLib (pacer/pacer/utils/motion_lib_smpl.py) loads AMASS clips from a pickle that does
not ship (.gitignore:4) and needs the licensed SMPL model.  This class keeps its query interface and its
on-device cache layout (motion_lib_smpl.py:334-341: gts/grs/lrs/gvs/gavs/dvs + length_starts) but fills the
cache with procedurally generated walking clips: 30 fps, 150-300 frames, root speed
U[0.5, 2.5] m/s, sinusoidal gait.  Query maths (`_calc_frame_blend` :596-606, `get_motion_state_smpl`
:485-563) follows the reference and is pinned by tests/golden/frame_blend.npz.
"""
import numpy as np
import torch

from ..gym import torch_utils as tu

# joint index (body index - 1) by name
_J = {n: i for i, n in enumerate(['L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe', 'Torso',
                                  'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist',
                                  'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand'])}


def _gait_pose_aa(t, speed, phase0):
    """(T,23,3) joint rotation vectors of a simple walking cycle."""
    T = t.shape[0]
    aa = np.zeros((T, 23, 3), np.float32)
    f = 0.7 + 0.45 * speed                       # stride frequency [Hz]
    ph = 2 * np.pi * f * t + phase0
    amp = 0.25 + 0.15 * speed
    aa[:, _J['L_Hip'], 1] = -amp * np.sin(ph)
    aa[:, _J['R_Hip'], 1] = amp * np.sin(ph)
    aa[:, _J['L_Knee'], 1] = 0.1 + 0.5 * amp * (1 + np.sin(ph - 1.2))
    aa[:, _J['R_Knee'], 1] = 0.1 + 0.5 * amp * (1 - np.sin(ph - 1.2))
    aa[:, _J['L_Ankle'], 1] = -0.1 * np.sin(ph + 0.6)
    aa[:, _J['R_Ankle'], 1] = 0.1 * np.sin(ph + 0.6)
    aa[:, _J['L_Shoulder'], 0] = -1.35
    aa[:, _J['R_Shoulder'], 0] = 1.35
    aa[:, _J['L_Shoulder'], 1] = 0.5 * amp * np.sin(ph)
    aa[:, _J['R_Shoulder'], 1] = -0.5 * amp * np.sin(ph)
    aa[:, _J['L_Elbow'], 2] = -0.3
    aa[:, _J['R_Elbow'], 2] = 0.3
    aa[:, _J['Torso'], 2] = 0.05 * np.sin(ph)
    return aa


class MotionLibBase:
    """The query side of the reference's MotionLib (sampling, frame blending, `get_motion_state_smpl`) over the per-frame cache a
    subclass fills: gts / grs / lrs / gvs / gavs / dvs (frames of all clips concatenated), length_starts, _motion_lengths,
    _motion_num_frames, _motion_dt, _motion_fps, _motion_weights, _motion_bodies, _motion_limb_weights, _motion_aa."""

    def load_motions(self, **kwargs):
        return None     # caches that are built once; kept for interface parity (humanoid_amp.py:269-271)

    def num_motions(self):
        return int(self._motion_lengths.shape[0])

    def get_total_length(self):
        return float(self._motion_lengths.sum())

    def sample_motions(self, n):                                                   # motion_lib_smpl.py:437-443
        return torch.multinomial(self._motion_weights, num_samples=n, replacement=True)

    def sample_time(self, motion_ids, truncate_time=None):                         # motion_lib_smpl.py:445-456
        phase = torch.rand(motion_ids.shape, device=self._device)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            motion_len = motion_len - truncate_time
        return phase * motion_len

    def get_motion_length(self, motion_ids=None):
        return self._motion_lengths if motion_ids is None else self._motion_lengths[motion_ids]

    def _calc_frame_blend(self, time, len, num_frames, dt):                        # motion_lib_smpl.py:596-606
        time = time.clone()
        phase = torch.clip(time / len, 0.0, 1.0)
        time[time < 0] = 0
        frame_idx0 = (phase * (num_frames - 1)).long()
        frame_idx1 = torch.min(frame_idx0 + 1, num_frames - 1)
        blend = (time - frame_idx0 * dt) / dt
        return frame_idx0, frame_idx1, blend

    def get_motion_state_smpl(self, motion_ids, motion_times, offset=None):        # motion_lib_smpl.py:485-563
        motion_len = self._motion_lengths[motion_ids]
        num_frames = self._motion_num_frames[motion_ids]
        dt = self._motion_dt[motion_ids]
        i0, i1, blend = self._calc_frame_blend(motion_times, motion_len, num_frames, dt)
        f0l = i0 + self.length_starts[motion_ids]
        f1l = i1 + self.length_starts[motion_ids]
        blend = blend.unsqueeze(-1)
        bexp = blend.unsqueeze(-1)
        rg_pos = (1.0 - bexp) * self.gts[f0l] + bexp * self.gts[f1l]
        if offset is not None:
            rg_pos = rg_pos + offset[..., None, :]
        body_vel = (1.0 - bexp) * self.gvs[f0l] + bexp * self.gvs[f1l]
        body_ang_vel = (1.0 - bexp) * self.gavs[f0l] + bexp * self.gavs[f1l]
        dof_vel = (1.0 - blend) * self.dvs[f0l] + blend * self.dvs[f1l]
        local_rot = tu.slerp(self.lrs[f0l], self.lrs[f1l], bexp)
        dof_pos = tu.quat_to_exp_map(local_rot[:, 1:]).reshape(local_rot.shape[0], -1)
        rb_rot = tu.slerp(self.grs[f0l], self.grs[f1l], bexp)
        return {
            "root_pos": rg_pos[..., 0, :].clone(), "root_rot": rb_rot[..., 0, :].clone(), "dof_pos": dof_pos.clone(),
            "root_vel": body_vel[..., 0, :].clone(), "root_ang_vel": body_ang_vel[..., 0, :].clone(),
            "dof_vel": dof_vel.view(dof_vel.shape[0], -1), "key_pos": rg_pos[:, self._key_body_ids],
            "motion_aa": self._motion_aa[f0l], "rg_pos": rg_pos, "rb_rot": rb_rot, "body_vel": body_vel,
            "body_ang_vel": body_ang_vel, "motion_bodies": self._motion_bodies[motion_ids],
            "motion_limb_weights": self._motion_limb_weights[motion_ids],
        }


class MotionLibSynthetic(MotionLibBase):
    def __init__(self, model, key_body_ids, device, num_motions=64, fps=30, seed=0):
        self._device = torch.device(device)
        self._key_body_ids = torch.as_tensor(key_body_ids, dtype=torch.long, device=self._device)
        self.num_bodies = model.num_bodies
        rng = np.random.default_rng(seed)
        parent = model.parent
        off = torch.tensor(model.joint_off, dtype=torch.float32)
        gts, grs, lrs, gvs, gavs, dvs, lens, nfr, starts = [], [], [], [], [], [], [], [], []
        self._motion_aa = []
        dt = 1.0 / fps
        total = 0
        for m in range(num_motions):
            T = int(rng.integers(150, 301))
            speed = rng.uniform(0.5, 2.5)
            t = np.arange(T, dtype=np.float32) * dt
            aa = torch.from_numpy(_gait_pose_aa(t, speed, rng.uniform(0, 2 * np.pi)))
            lq = torch.cat([torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(T, 1, 4), tu.exp_map_to_quat(aa.reshape(-1, 3)).view(T, 23, 4)], 1)
            root_pos = torch.zeros(T, 3)
            root_pos[:, 0] = torch.from_numpy(speed * t)
            root_pos[:, 2] = 0.92 + 0.01 * torch.sin(torch.from_numpy(4 * np.pi * (0.7 + 0.45 * speed) * t))
            gq = torch.zeros(T, self.num_bodies, 4)
            gp = torch.zeros(T, self.num_bodies, 3)
            gq[:, 0], gp[:, 0] = lq[:, 0], root_pos
            for b in range(1, self.num_bodies):
                p = int(parent[b])
                gp[:, b] = gp[:, p] + tu.quat_apply(gq[:, p], off[b].expand(T, 3))
                gq[:, b] = tu.normalize(tu.quat_mul(gq[:, p], lq[:, b]))
            vel = torch.zeros_like(gp)
            vel[:-1] = (gp[1:] - gp[:-1]) / dt
            vel[-1] = vel[-2]
            dq = tu.quat_mul(gq[1:], tu.quat_conjugate(gq[:-1]))
            ang = torch.zeros_like(gp)
            ang[:-1] = tu.quat_to_exp_map(tu.normalize(dq).reshape(-1, 4)).view(T - 1, self.num_bodies, 3) / dt
            ang[-1] = ang[-2]
            dl = tu.quat_mul(tu.quat_conjugate(lq[:-1, 1:]), lq[1:, 1:])            # motion_lib_smpl.py:44-50
            dv = torch.zeros(T, 69)
            dv[:-1] = (tu.quat_to_exp_map(tu.normalize(dl).reshape(-1, 4)).view(T - 1, 23, 3) / dt).reshape(T - 1, 69)
            dv[-1] = dv[-2]
            gts.append(gp); grs.append(gq); lrs.append(lq); gvs.append(vel); gavs.append(ang); dvs.append(dv)
            self._motion_aa.append(torch.cat([torch.zeros(T, 3), aa.reshape(T, 69)], 1))
            lens.append(dt * (T - 1)); nfr.append(T); starts.append(total)
            total += T
        dev = self._device
        self.gts, self.grs, self.lrs = torch.cat(gts).to(dev), torch.cat(grs).to(dev), torch.cat(lrs).to(dev)
        self.gvs, self.gavs, self.dvs = torch.cat(gvs).to(dev), torch.cat(gavs).to(dev), torch.cat(dvs).to(dev)
        self._motion_aa = torch.cat(self._motion_aa).to(dev)
        self._motion_lengths = torch.tensor(lens, dtype=torch.float32, device=dev)
        self._motion_num_frames = torch.tensor(nfr, dtype=torch.long, device=dev)
        self._motion_dt = torch.full((num_motions,), dt, dtype=torch.float32, device=dev)
        self._motion_fps = torch.full((num_motions,), float(fps), device=dev)
        self.length_starts = torch.tensor(starts, dtype=torch.long, device=dev)
        self._motion_weights = torch.full((num_motions,), 1.0 / num_motions, device=dev)
        self._motion_bodies = torch.zeros(num_motions, 17, device=dev)
        self._motion_limb_weights = torch.zeros(num_motions, 10, device=dev)
