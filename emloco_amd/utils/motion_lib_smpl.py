"""MotionLib over AMASS pickles -- the reference's motion data path (pacer/pacer/utils/motion_lib_smpl.py:84-131,176-363).

On-disk format (scripts/data_process/convert_amass_isaac.py:309-317, read with joblib): a dict  clip name -> {
    "pose_quat_global" (T, 24, 4) global body rotations xyzw in the humanoid's body order,
    "root_trans_offset" (T, 3) root translation, "pose_aa" (T, 72) SMPL axis-angle pose, "beta" (16,), "gender", "fps" }.
Per clip the reference builds a poselib SkeletonMotion on the env's skeleton (`load_motion_with_skeleton`): local rotations from the
global ones, global translations by forward kinematics over the skeleton's local translations, global linear velocities as
gaussian-filtered np.gradient of the translations and angular velocities as gaussian-filtered quaternion differences
(poselib skeleton3d.py:1249-1272), joint velocities from successive local rotations (:44-67); the frames of all clips are concatenated into
the cache `get_motion_state_smpl` blends from (:334-363).  The same here, restated on torch / numpy / scipy (host-side data
preparation, as in the reference), pinned by tests/golden/motion_amass.npz = the reference's own loader run on two clips in that format.

What is NOT here: `fix_height` through the SMPL mesh (:70-82: needs the licensed SMPL model).  The reset path places a sampled pose on
the ground by its lowest collision point instead (env/tasks/humanoid_amp.py `_lowest_point`), clip by clip at reset time,
so a clip's stored height is only a starting point.  `masterfoot` skeletons are not part of this path.
"""
import numpy as np
import torch

from .motion_lib_synthetic import MotionLibBase


def _qmul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                        w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)


def _qinv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _qnormalize(q):                 # poselib quat_normalize: real part made non-negative, unit length
    q = torch.where(q[..., 3:] < 0, -q, q)
    return q / q.norm(dim=-1, keepdim=True).clamp(min=1e-9)


def _qmul_norm(a, b):
    return _qnormalize(_qmul(a, b))


def _qrot(q, v):                    # poselib quat_rotate: q (v, 0) q^-1
    return _qmul(_qmul(q, torch.cat([v, torch.zeros_like(v[..., :1])], -1)), _qinv(q))[..., :3]


def _angle_axis(q):                 # poselib quat_angle_axis: angle in [0, pi] from the real part, axis = normalised imaginary part
    angle = (2 * q[..., 3] ** 2 - 1).clamp(-1, 1).arccos()
    axis = q[..., :3] / q[..., :3].norm(dim=-1, keepdim=True).clamp(min=1e-9)
    return angle, axis


def clip_cache(clip, local_translation, parents):
    """One clip on one skeleton -> the per-frame arrays of the reference's SkeletonMotion (float64): global translation (T,24,3),
    global rotation, local rotation (T,24,4), global velocity, global angular velocity (T,24,3), joint velocities (T,23,3), fps."""
    from scipy.ndimage import gaussian_filter1d
    gq = torch.as_tensor(np.asarray(clip["pose_quat_global"]), dtype=torch.float64)
    trans = torch.as_tensor(np.asarray(clip["root_trans_offset"]), dtype=torch.float64)
    fps = float(clip.get("fps", 30))
    dt = 1.0 / fps
    T, J, _ = gq.shape
    lt = torch.as_tensor(np.asarray(local_translation), dtype=torch.float64)
    lq = torch.zeros_like(gq)
    gp = torch.zeros(T, J, 3, dtype=torch.float64)
    for b in range(J):
        p = int(parents[b])
        if p < 0:
            lq[:, b], gp[:, b] = gq[:, b], trans
        else:
            lq[:, b] = _qmul_norm(_qinv(gq[:, p]), gq[:, b])
            gp[:, b] = gp[:, p] + _qrot(gq[:, p], lt[b].expand(T, 3))
    gv = torch.from_numpy(gaussian_filter1d(np.gradient(gp.numpy(), axis=0), 2, axis=0, mode="nearest") / dt)
    dq = torch.zeros_like(gq)
    dq[..., 3] = 1.0
    dq[:-1] = _qmul_norm(gq[1:], _qinv(gq[:-1]))
    ang, ax = _angle_axis(dq)
    gav = torch.from_numpy(gaussian_filter1d((ax * ang.unsqueeze(-1) / dt).numpy(), 2, axis=0, mode="nearest"))
    dl = _qmul_norm(_qinv(lq[:-1]), lq[1:])
    a2, x2 = _angle_axis(dl)
    dv = (x2 * a2.unsqueeze(-1) / dt)[:, 1:]
    dv = torch.cat([dv, dv[-1:]], 0)
    return dict(gts=gp, grs=gq, lrs=lq, gvs=gv, gavs=gav, dvs=dv, fps=fps)


class MotionLib(MotionLibBase):
    """`MotionLib(motion_file, key_body_ids, device)` + `load_motions(skeleton_trees, gender_betas, limb_weights, random_sample,
    start_idx)` (motion_lib_smpl.py:176-363): one clip per skeleton (= per env), drawn from the pickle with `_sampling_prob`, each
    built on ITS skeleton.  A skeleton is anything with `.joint_off` (24, 3) and `.parent` (24,) -- emloco_amd.model.HumanoidModel --
    or a (local_translation, parent_indices) pair."""

    def __init__(self, motion_file, key_body_ids, device, fix_height=False, masterfoot_conifg=None, min_length=-1):
        import joblib
        if masterfoot_conifg is not None:
            raise NotImplementedError("masterfoot skeletons are not part of this path")
        if fix_height:
            raise NotImplementedError("fix_height needs the licensed SMPL mesh; the reset path grounds a sampled pose by its lowest "
                                      "collision point instead (humanoid_amp.py _lowest_point): pass fix_height=False")
        self._device = torch.device(device)
        self._key_body_ids = torch.as_tensor(key_body_ids, dtype=torch.long, device=self._device)
        data = motion_file if isinstance(motion_file, dict) else joblib.load(motion_file)
        if min_length != -1:
            data = {k: v for k, v in data.items() if len(v["pose_quat_global"]) >= min_length}
        self._motion_data_list = list(data.values())
        self._motion_data_keys = np.array(list(data.keys()))
        self._num_unique_motions = len(self._motion_data_list)
        if self._num_unique_motions == 0:
            raise ValueError("the motion file holds no clip (after the min_length filter)")
        self._curr_motion_ids = None
        self._termination_history = torch.zeros(self._num_unique_motions)
        self._success_rate = torch.zeros(self._num_unique_motions)
        self._sampling_history = torch.zeros(self._num_unique_motions)
        self._sampling_prob = torch.ones(self._num_unique_motions) / self._num_unique_motions

    @staticmethod
    def _skeleton(s):
        if hasattr(s, "joint_off"):
            return np.asarray(s.joint_off, np.float64), np.asarray(s.parent)
        lt, par = s
        return np.asarray(lt, np.float64), np.asarray(par)

    def load_motions(self, skeleton_trees, gender_betas=None, limb_weights=None, random_sample=True, start_idx=0):
        n = len(skeleton_trees)
        if random_sample:
            ids = torch.multinomial(self._sampling_prob, num_samples=n, replacement=True)
        else:
            ids = torch.clip(torch.arange(n) + start_idx, 0, self._num_unique_motions - 1)
        self._curr_motion_ids = ids
        self._sampling_batch_prob = self._sampling_prob[ids] / self._sampling_prob[ids].sum()
        keys = ("gts", "grs", "lrs", "gvs", "gavs", "dvs")
        parts = {k: [] for k in keys}
        lens, fpss, nfr, aa, bodies = [], [], [], [], []
        built = {}                      # (clip, skeleton) pairs repeat when envs share a body shape: build each once
        for j, i in enumerate(ids.tolist()):
            clip = self._motion_data_list[i]
            lt, par = self._skeleton(skeleton_trees[j])
            key = (i, lt.tobytes())
            if key not in built:
                built[key] = clip_cache(clip, lt, par)
            c = built[key]
            for k in keys:
                parts[k].append(c[k].reshape(c[k].shape[0], -1, c[k].shape[-1]) if k != "dvs" else c[k].reshape(c[k].shape[0], -1))
            T = c["gts"].shape[0]
            nfr.append(T); fpss.append(c["fps"]); lens.append((T - 1) / c["fps"])
            if "beta" in clip:
                aa.append(np.asarray(clip["pose_aa"], np.float64).reshape(-1, 72))
                bodies.append(torch.as_tensor(np.asarray(gender_betas[j]), dtype=torch.float32) if gender_betas is not None else torch.zeros(17))
            else:
                aa.append(np.zeros((T, 72)))
                bodies.append(torch.zeros(17))
        dev = self._device
        for k in keys:
            setattr(self, k, torch.cat(parts[k]).float().to(dev).contiguous())
        self._motion_lengths = torch.tensor(lens, dtype=torch.float32, device=dev)
        self._motion_fps = torch.tensor(fpss, dtype=torch.float32, device=dev)
        self._motion_dt = 1.0 / self._motion_fps
        self._motion_num_frames = torch.tensor(nfr, dtype=torch.long, device=dev)
        self._motion_aa = torch.tensor(np.concatenate(aa), dtype=torch.float32, device=dev)
        self._motion_bodies = torch.stack(bodies).to(dev).float()
        self._motion_limb_weights = (torch.as_tensor(limb_weights).float().to(dev) if limb_weights is not None else torch.zeros(n, 10, device=dev))
        shifted = self._motion_num_frames.roll(1)
        shifted[0] = 0
        self.length_starts = shifted.cumsum(0)
        self._motion_weights = self._sampling_batch_prob.to(dev).float()
        self._num_motions = n
        self.motion_ids = torch.arange(n, dtype=torch.long, device=dev)
        return self

    def get_motion_files(self, motion_ids):
        return self._motion_data_keys[self._curr_motion_ids[torch.as_tensor(motion_ids).cpu()]]
