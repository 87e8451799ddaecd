"""Golden vectors for the JRDB predictor variant: the reference's TransMotionJRDB (model_jrdb.py) and
dataset_jrdb.batch_process_coords run on CPU at reduced width.

    python tests/golden/gen_golden_jrdb.py        ->  tests/golden/predictor_jrdb.npz

torchvision / matplotlib (imported by dataset_jrdb.py for an augmentation lambda and plots) are absent and mocked by the
import shim; the evaluation-mode path generated here does not touch them.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim as shim  # noqa: E402

shim._MOCK_ROOTS.extend(["matplotlib", "torchvision", "pyemd"])
shim.install_predictor()
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import model_jrdb as M
    from dataset_jrdb import batch_process_coords
    from utils.metrics import MSE_LOSS_MULTI
    g = torch.Generator().manual_seed(17)
    B, N, J = 3, 3, 26
    cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}, "DATA": {"train_datasets": ["jrdb_all_visual_cues"]}}
    joints = torch.randn(B, N, 21, J, 4, generator=g) * 0.5
    joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.3, dim=2)
    masks = torch.ones(B, N, 21, J)
    padding_mask = torch.zeros(B, N, dtype=torch.bool)
    padding_mask[1, 2] = True
    padding_mask[2, 1:] = True
    raw = joints.clone()
    out = {}
    for sel in ("traj+all", "traj+2dbox", "traj+3dpose", "traj"):
        ij, im, oj, om, pm = batch_process_coords(joints.clone(), masks, padding_mask, cfg, modality_selection=sel, training=False)
        out[f"in_joints.{sel}"] = ij.numpy()
        out[f"out_joints.{sel}"] = oj.numpy()
    in_joints, _, out_joints, _, pm = batch_process_coords(joints.clone(), masks, padding_mask, cfg, training=False)
    torch.manual_seed(23)
    model = M.TransMotionJRDB(tok_dim=246, nhid=32, nhead=4, dim_feedfwd=64, nlayers_local=2, nlayers_global=1, nmode=4,
                              output_scale=1, obs_and_pred=21, num_tokens=J, device="cpu", multi_modal=True).float()
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.eval()
    pred = model(in_joints.clone(), pm.clone())
    pred_lim = model(in_joints.clone(), pm.clone(), limit_obs=3)
    loss = MSE_LOSS_MULTI(pred[:, 9:], out_joints)
    loss.backward()
    grads = {n_: p.grad.clone() for n_, p in model.named_parameters() if p.grad is not None}
    pick = ["fc_in_traj.weight", "fc_in_2dbb.weight", "fc_in_3dpose.bias", "local_former.layers.0.self_attn.in_proj_weight",
            "local_former.layers.1.linear2.weight", "global_former.layers.0.norm1.weight", "predict_head.0.weight",
            "pose3d_encoder.learned_encoding.weight", "bb2d_encoder.learned_encoding.weight"]
    out.update({"joints": raw.numpy(), "padding_mask": padding_mask.float().numpy(), "pm": pm.numpy(), "pred": pred.detach().numpy(),
                "pred_limit_obs3": pred_lim.detach().numpy(), "loss": np.float32(loss.item()),
                "torch_version": np.array(torch.__version__)})
    out.update({"sd__" + k.replace(".", "__"): v.numpy() for k, v in sd.items()})
    out.update({"grad__" + k.replace(".", "__"): grads[k].numpy() for k in pick})
    path = os.path.join(HERE, "predictor_jrdb.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", loss.item())


if __name__ == "__main__":
    main()
