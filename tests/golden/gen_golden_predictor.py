"""Golden vectors for the predictor path: the reference's TransMotionJTA + metrics + batch_process_coords run on CPU.

    python tests/golden/gen_golden.py predictor

A reduced configuration (d=32, 4 heads, ff=64, 2 local + 1 global layers) keeps the fixture small; the
architecture code path is the one the full model uses.  Weights are stored in the fixture (state_dict), so the
test loads them into this repo's mirror and compares logits, the EmLoco training loss and gradients.
"""
import numpy as np
import torch

import _ref_shim
from gen_golden import save


def gen_predictor():
    _ref_shim.install_predictor()
    import model_jta as M
    from dataset_jta import batch_process_coords
    from utils.metrics import MSE_LOSS, MSE_LOSS_MULTI
    from learning.value_pose_net import ValuePoseNet

    g = torch.Generator().manual_seed(7)
    B, N, J = 3, 3, 49
    cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}}
    joints = torch.randn(B, N, 21, J, 4, generator=g) * 0.5
    joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.3, dim=2)
    masks = torch.ones(B, N, 21, J)
    padding_mask = torch.zeros(B, N, dtype=torch.bool)
    padding_mask[1, 2] = True            # one padded person (fully masked local sequence)
    padding_mask[2, 1:] = True
    raw_joints = joints.clone()
    in_joints, in_masks, out_joints, out_masks, pm = batch_process_coords(joints.clone(), masks, padding_mask, cfg, training=True)
    save("predictor_batch", joints=raw_joints, padding_mask=padding_mask.float(), in_joints=in_joints, out_joints=out_joints, pm=pm)

    for multi in (False, True):
        torch.manual_seed(11)
        model = M.TransMotionJTA(tok_dim=453, nhid=32, nhead=4, dim_feedfwd=64, nlayers_local=2, nlayers_global=1, nmode=4,
                                 output_scale=1, obs_and_pred=21, num_tokens=J, device="cpu", multi_modal=multi).float()
        # non-trivial biases / norms so every gradient path is exercised
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.eval()                      # dropout off; masks off -> deterministic
        pred = model(in_joints.clone(), pm.clone())
        loss_fn = MSE_LOSS_MULTI if multi else MSE_LOSS
        out_gt = out_joints[:, :, 0:1, :2]
        gt_full = torch.cat([in_joints[:, :, 0:1, :2], out_gt], dim=1)
        mse = loss_fn(pred[:, 9:], out_joints)
        # EmLoco loss wiring (train_jta.py:288-308): pred_traj = [0] + 12 predicted steps of mode 0
        torch.manual_seed(5)
        vnet = ValuePoseNet(use_pose=True, use_vel=True)
        pose = torch.randn(B, 24, 3, generator=g) * 0.3
        vel = (in_joints[:, 8, 0, :2] - in_joints[:, 7, 0, :2]) * 2.5
        traj0 = pred[:, 9:, 0, :2]
        pred_traj = torch.cat([torch.zeros(B, 1, 2), traj0], dim=1)
        value, vloss = vnet.calc_embodied_motion_loss(pred_traj, pose.clone(), vel.clone())
        loss = mse + 1.0 * vloss
        loss.backward()
        grads = {n_: p.grad.clone() for n_, p in model.named_parameters() if p.grad is not None}
        pick = ["fc_in_traj.weight", "local_former.layers.0.self_attn.in_proj_weight", "local_former.layers.0.self_attn.in_proj_bias",
                "local_former.layers.1.linear1.weight", "local_former.layers.1.norm2.weight", "global_former.layers.0.self_attn.out_proj.weight",
                "global_former.layers.0.linear2.bias", "pose3d_encoder.learned_encoding.weight", "fc_in_3dpose.weight"]
        pick += ["predict_head.0.weight"] if multi else ["fc_out_traj.weight"]
        tag = "multi" if multi else "single"
        save(f"predictor_{tag}", in_joints=in_joints, pm=pm, out_joints=out_joints, pred=pred, mse=mse, pose=pose, vel=vel,
             value=value, vloss=vloss, loss=loss,
             **{"sd__" + k.replace(".", "__"): v for k, v in sd.items()},
             **{"vn__" + k.replace(".", "__"): v for k, v in vnet.state_dict().items()},
             **{"grad__" + k.replace(".", "__"): grads[k] for k in pick})
