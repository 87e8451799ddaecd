"""TransMotionJRDB -- mirror of /root/reference/social-transmotion/model_jrdb.py:12-143,282-310.

Same parameters as TransMotionJTA (it subclasses it, unused 3-D box / 2-D pose encoders included, so reference
checkpoints load), different token layout: per person 26 tokens = trajectory + 2-D box + 24 3-D pose joints, giving a
local sequence of S = 21 + 9 + 9*24 = 246 tokens.  Random masks are drawn in the reference's order (trajectory, frame,
2-D box modality, 3-D pose modality, joints).  The transformer stack and heads are the shared `_transform`.
"""
import numpy as np
import torch

from .model_jta import TransMotionJTA


class TransMotionJRDB(TransMotionJTA):
    def forward(self, tgt, padding_mask, random_masking=False, limit_obs=0, frame_masking=False, noisy_traj=False):
        dev = self.device
        B, in_F, NJ, K = tgt.shape
        Fr, J = self.obs_and_pred, self.token_num
        out_F = Fr - in_F
        N = NJ // J
        i_idx = np.append(np.arange(0, in_F), np.repeat([in_F - 1], out_F))
        tgt = tgt[:, i_idx].reshape(B, Fr, N, J, K).to(dev)
        mr_traj = 0.2 if random_masking else 0
        mr_joints = 0.2 if random_masking else 0
        mr_mod = 0.3 if random_masking else 0
        mr_frame = 0.2 if frame_masking else 0
        R = lambda *s: torch.rand(s).float().to(dev)            # CPU draws moved to the device, as the reference does
        tgt_traj = tgt[:, :, :, 0, :2] * (R(B, Fr, N) > mr_traj).unsqueeze(3)
        frame_mask = (R(B, in_F) > mr_frame).unsqueeze(2).unsqueeze(3)
        tgt_traj = torch.cat([tgt_traj[:, :in_F] * frame_mask, tgt_traj[:, in_F:]], dim=1)
        sel_2dbb = (R(B, 1, N, 1) > mr_mod)
        sel_3dpose = (R(B, 1, N, 1) > mr_mod).unsqueeze(4)
        tgt_vis = tgt[:, :, :, 1:]
        tgt_2dbb = tgt_vis[:, :, :, 0, :4] * sel_2dbb
        tgt_3dpose = tgt_vis[:, :, :, 1:, :3] * sel_3dpose
        tgt_3dpose = tgt_3dpose * (R(B, Fr, N, self.joints_3dpose) > mr_joints).unsqueeze(4)
        if limit_obs != 0:
            lm = torch.ones((B, Fr, N), device=dev)
            lm[:, :(9 - limit_obs)] = 0
            tgt_traj, tgt_2dbb = tgt_traj * lm.unsqueeze(3), tgt_2dbb * lm.unsqueeze(3)
            tgt_3dpose = tgt_3dpose * lm[..., None, None]
        t = self._embed_traj(tgt_traj, Fr, N)                                                        # (B,21,N,d)
        bb2 = self._embed(tgt_2dbb[:, :9], self.fc_in_2dbb, self.bb2d_encoder, 9)                    # (B,9,N,d)
        p3 = tgt_3dpose[:, :9].transpose(2, 3).reshape(B, -1, N, 3)
        p3 = self._embed(p3, self.fc_in_3dpose, self.pose3d_encoder, p3.shape[1])                    # (B,216,N,d)
        seq = torch.cat((t, bb2, p3), dim=1)                                                         # (B,246,N,d)
        return self._transform(seq, padding_mask, B, N, Fr)


def create_model(config, logger=None):
    """model_jrdb.py:282-310."""
    m = config["MODEL"]
    if m.get("type", "transmotion") != "transmotion":
        raise ValueError(f"Model type '{m['type']}' not found")
    return TransMotionJRDB(tok_dim=m["seq_len"], nhid=m["dim_hidden"], nhead=m["num_heads"], nmode=m.get("num_modes", 1),
                           dim_feedfwd=m["dim_feedforward"], nlayers_local=m["num_layers_local"], nlayers_global=m["num_layers_global"],
                           output_scale=m["output_scale"], obs_and_pred=config["TRAIN"]["input_track_size"] + config["TRAIN"]["output_track_size"],
                           num_tokens=m["token_num"], device=config["DEVICE"], multi_modal=config.get("MULTI_MODAL", False)
                           ).float().to(config["DEVICE"])
