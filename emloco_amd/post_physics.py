"""Python side of the fused post-physics kernels (include/emloco_task.h).

`PostPhysics` holds the small constant tables of the SMPL-humanoid task configuration (mirror
permutation, foot mask, key bodies, AMP dof subset; reference: pacer/pacer/env/tasks/humanoid.py:289-335,
pacer/pacer/data/cfg/pacer.yaml:50-51) on the device and fills an `EmlocoTaskBufs` from the task's
tensors.  The kernels run on torch's current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .sim import current_stream_handle, dptr

BODY_NAMES = ['Pelvis', 'L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe', 'Torso',
              'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand',
              'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']
LEFT_TO_RIGHT = [0, 5, 6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 19, 20, 21, 22, 23, 14, 15, 16, 17, 18]


def dof_subset_indices(remove=("L_Hand", "R_Hand", "L_Toe", "R_Toe")):
    """humanoid.py:289-326: all joints except hands and toes, 3 dofs each."""
    names = BODY_NAMES[1:]
    return np.concatenate([np.arange(i * 3, i * 3 + 3) for i, n in enumerate(names) if n not in remove]).astype(np.int32)


class PostPhysics:
    def __init__(self, device, key_bodies=("R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"),
                 contact_bodies=("R_Ankle", "L_Ankle", "R_Toe", "L_Toe"), head_body="Head"):
        L.require_device()
        self.lib = L.load()
        self.device = torch.device(device)
        dev = self.device
        self.left_to_right = torch.tensor(LEFT_TO_RIGHT, dtype=torch.int32, device=dev)
        mask = np.zeros(L.NB, np.uint8)
        mask[[BODY_NAMES.index(n) for n in contact_bodies]] = 1
        self.contact_body_mask = torch.from_numpy(mask).to(dev)
        self.key_bodies = torch.tensor([BODY_NAMES.index(n) for n in key_bodies], dtype=torch.int32, device=dev)
        self.dof_subset = torch.from_numpy(dof_subset_indices()).to(dev)
        self.head_body = BODY_NAMES.index(head_body)

    def make_bufs(self, *, n_env, heightfield, dt, traj_dur, sample_dt, hscale, vscale, power_coef, fail_dist,
                  max_episode_length, rb_state, dof_state, dof_force, contact_force, betas, traj_verts,
                  progress_buf, reset_buf, terminate_buf, obs_buf, flip_obs_buf, rew_buf, reward_raw, amp_obs_buf):
        assert heightfield.dtype == torch.int16 and heightfield.dim() == 2
        assert progress_buf.dtype == torch.int64 and reset_buf.dtype == torch.int64 and terminate_buf.dtype == torch.int64
        p = lambda t: dptr(t).value
        self._keep = (heightfield, rb_state, dof_state, dof_force, contact_force, betas, traj_verts, progress_buf,
                      reset_buf, terminate_buf, obs_buf, flip_obs_buf, rew_buf, reward_raw, amp_obs_buf)
        return L.TaskBufs(int(n_env), int(heightfield.shape[0]), int(heightfield.shape[1]), int(self.head_body),
                          int(self.dof_subset.numel()), float(dt), float(traj_dur), float(sample_dt), float(hscale),
                          float(vscale), float(power_coef), float(fail_dist), float(max_episode_length),
                          p(rb_state), p(dof_state), p(dof_force), p(contact_force), p(betas), p(traj_verts),
                          p(heightfield), p(self.left_to_right), p(self.contact_body_mask), p(self.key_bodies),
                          p(self.dof_subset), p(progress_buf), p(reset_buf), p(terminate_buf), p(obs_buf),
                          p(flip_obs_buf), p(rew_buf), p(reward_raw), p(amp_obs_buf))

    def run(self, bufs, mode=L.POST_STEP, env_ids_i32=None, returns=None):
        """`returns` = (EmlocoLocoValStep ctypes struct, inverted bool / uint8 tensor or None): the LocoVal return bookkeeping of every
        env in the same launch (emloco_task_post_physics_returns; all envs, mode with REWARD and RESET)."""
        if returns is not None:
            step, inv = returns
            rc = self.lib.emloco_task_post_physics_returns(C.byref(bufs), int(mode), C.byref(step), dptr(inv), current_stream_handle(self.device))
            L.check(rc, "emloco_task_post_physics_returns")
            return
        n = 0 if env_ids_i32 is None else int(env_ids_i32.numel())
        if env_ids_i32 is not None and n == 0:
            return
        rc = self.lib.emloco_task_post_physics(C.byref(bufs), int(mode), dptr(env_ids_i32), n,
                                               current_stream_handle(self.device))
        L.check(rc, "emloco_task_post_physics")

    def amp_rows(self, root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos, betas):
        n = int(root_pos.shape[0])
        out = torch.empty((n, L.AMP_ROW), dtype=torch.float32, device=self.device)
        args = [t.contiguous() for t in (root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos, betas)]
        rc = self.lib.emloco_task_amp_rows(n, *[dptr(t) for t in args], dptr(self.dof_subset),
                                           int(self.dof_subset.numel()), dptr(out), current_stream_handle(self.device))
        L.check(rc, "emloco_task_amp_rows")
        return out

    def pd_targets(self, actions, offset, scale, zero_mask, out, actions_copy=None):
        rc = self.lib.emloco_task_pd_targets_copy(int(actions.shape[0]), dptr(actions), dptr(offset), dptr(scale),
                                                  dptr(zero_mask), dptr(out), dptr(actions_copy), current_stream_handle(self.device))
        L.check(rc, "emloco_task_pd_targets_copy")
        return out
