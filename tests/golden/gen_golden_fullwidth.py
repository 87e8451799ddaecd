"""Full-width predictor goldens: the reference's TransMotionJTA (S = 453) and TransMotionJRDB (S = 246) at the production width
(d = 128, 4 heads of 32, ff = 1024) with 1 local + 1 global layer, B = 2 scenes x N = 2 people, run on CPU.

    python tests/golden/gen_golden_fullwidth.py     ->  tests/golden/predictor_fullwidth_{jta,jrdb}.npz
    python tests/golden/gen_golden_fullwidth.py jta_deep jta_deep_mm
                                                    ->  tests/golden/predictor_fulldepth_{jta,jta_mm}.npz
    python tests/golden/gen_golden_fullwidth.py jta_bool
                                                    ->  tests/golden/predictor_boolmask_jta.npz

`jta_bool` hands the model the BOOL padding mask `collate_batch` produces (dataset_jta.py:23) instead of the float copy
`batch_process_coords` returns (:84): torch then masks padded persons' keys with -inf instead of biasing them by +1.  B = 3 scenes
x N = 3 people (four padded), 2 local + 2 global layers.  Pins the masked semantics, under which this repo may skip padded
persons in the local former altogether.

The `_deep` kinds are the SHIPPED model (social-transmotion/configs/jta_all_visual_cues.yaml:20-33): 6 local + 3 global layers;
`jta_deep` single-mode with the EmLoco loss (configs[3]), `jta_deep_mm` with the 20 prediction heads and MSE_LOSS_MULTI (configs[4]).

Head dimension 32 is the one this repo's fused attention kernels serve, so these fixtures put them (and the d = 128 GEMM
tiles) inside a comparison with reference output (the d = 32 fixtures take the composed attention path).  The weights come from
tests/fullwidth_weights.py (a formula evaluated on both sides); the fixtures hold inputs, logits, losses and gradients
(small tensors whole, large ones as strided samples).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_shim as shim  # noqa: E402

shim._MOCK_ROOTS.extend(["matplotlib", "torchvision", "pyemd"])
shim.install_predictor()
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fullwidth_weights import make_state_dict, sample  # noqa: E402

PICK_WHOLE = ["fc_in_traj.weight", "fc_in_traj.bias", "local_former.layers.0.self_attn.in_proj_bias", "local_former.layers.0.norm1.weight",
              "local_former.layers.0.norm2.bias", "local_former.layers.0.linear2.bias", "global_former.layers.0.self_attn.out_proj.bias",
              "global_former.layers.0.norm2.weight"]
PICK_SAMPLE = ["local_former.layers.0.self_attn.in_proj_weight", "local_former.layers.0.self_attn.out_proj.weight",
               "local_former.layers.0.linear1.weight", "local_former.layers.0.linear2.weight", "global_former.layers.0.self_attn.in_proj_weight",
               "global_former.layers.0.linear1.weight", "pose3d_encoder.learned_encoding.weight", "fc_in_3dpose.weight"]


def run(kind):
    torch.set_num_threads(8)
    deep = kind.startswith("jta_deep")
    mm = kind.endswith("_mm")
    nl, ng, nmode = (6, 3, 20) if deep else (1, 1, 4)
    boolmask = kind == "jta_bool"
    if boolmask:
        nl, ng = 2, 2
    fname = {"jta_deep": "predictor_fulldepth_jta", "jta_deep_mm": "predictor_fulldepth_jta_mm",
             "jta_bool": "predictor_boolmask_jta"}.get(kind, f"predictor_fullwidth_{kind}")
    g = torch.Generator().manual_seed({"jta": 31, "jrdb": 37, "jta_deep": 41, "jta_deep_mm": 43, "jta_bool": 47}[kind])
    B, N = (3, 3) if boolmask else (2, 2)
    if deep or boolmask:
        kind = "jta"
    if kind == "jta":
        import model_jta as M
        from dataset_jta import batch_process_coords
        if mm:
            from utils.metrics import MSE_LOSS_MULTI as LOSS
        else:
            from utils.metrics import MSE_LOSS as LOSS
        J = 49
        model = M.TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=nl, nlayers_global=ng, nmode=nmode, output_scale=1,
                                 obs_and_pred=21, num_tokens=J, device="cpu", multi_modal=mm).float()
        cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}}
        head = "predict_head.19.weight" if mm else "fc_out_traj.weight"
    else:
        import model_jrdb as M
        from dataset_jrdb import batch_process_coords
        from utils.metrics import MSE_LOSS_MULTI as LOSS
        J = 26
        model = M.TransMotionJRDB(tok_dim=246, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=1, nlayers_global=1, nmode=4, output_scale=1,
                                  obs_and_pred=21, num_tokens=J, device="cpu", multi_modal=True).float()
        cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}, "DATA": {"train_datasets": ["jrdb_all_visual_cues"]}}
        head = "predict_head.0.weight"
    joints = torch.randn(B, N, 21, J, 4, generator=g) * 0.5
    joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.3, dim=2)
    masks = torch.ones(B, N, 21, J)
    padding_mask = torch.zeros(B, N, dtype=torch.bool)
    padding_mask[1, 1] = True                                         # one padded person
    if boolmask:
        padding_mask[1, 2] = True
        padding_mask[2, 1:] = True
        joints[padding_mask] = 0                                      # pad_sequence fills padded persons with zeros (dataset_jta.py:20)
    in_joints, _, out_joints, _, pm = batch_process_coords(joints.clone(), masks, padding_mask, cfg, training=(kind == "jta"))   # JRDB training mode applies a random (torchvision) rotation
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = make_state_dict(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    if boolmask:
        pm = padding_mask
    pred = model(in_joints.clone(), pm.clone())
    loss = LOSS(pred[:, 9:], out_joints)
    extra = {}
    if kind == "jta" and not mm:                                      # EmLoco loss wiring (train_jta.py:288-308)
        from learning.value_pose_net import ValuePoseNet
        torch.manual_seed(5)
        vnet = ValuePoseNet(use_pose=True, use_vel=True)
        pose = torch.randn(B, 24, 3, generator=g) * 0.3
        vel = (in_joints[:, 8, 0, :2] - in_joints[:, 7, 0, :2]) * 2.5
        pred_traj = torch.cat([torch.zeros(B, 1, 2), pred[:, 9:, 0, :2]], dim=1)
        value, vloss = vnet.calc_embodied_motion_loss(pred_traj, pose.clone(), vel.clone())
        extra = dict(pose=pose, vel=vel, value=value.detach(), mse=loss.detach().clone(),
                     **{"vn__" + k.replace(".", "__"): v for k, v in vnet.state_dict().items()})
        loss = loss + 1.0 * vloss
    loss.backward()
    grads = {n_: p.grad.clone() for n_, p in model.named_parameters() if p.grad is not None}
    out = dict(in_joints=in_joints, pm=pm, out_joints=out_joints, pred=pred.detach(), loss=loss.detach(),
               weight_seed=np.array(1234), n_params=np.array(sum(int(np.prod(s)) for s in shapes.values())),
               weight_checksum=np.array(float(sum(np.abs(v).sum(dtype=np.float64) for v in sd.values()))),
               keys=np.array("\n".join(f"{k} {' '.join(map(str, shapes[k]))}" for k in sorted(shapes))), **extra)
    whole, samp = list(PICK_WHOLE), list(PICK_SAMPLE)
    if boolmask:
        whole += ["local_former.layers.1.norm2.weight", "global_former.layers.1.self_attn.out_proj.bias", "double_id_encoder.person_encoding.weight"]
        samp += ["local_former.layers.1.linear1.weight", "global_former.layers.1.linear2.weight"]
    if deep:                                                          # the last layers too: nine stacked post-norm layers
        whole += ["local_former.layers.5.norm2.weight", "local_former.layers.5.linear2.bias", "global_former.layers.2.norm2.weight",
                  "global_former.layers.2.self_attn.out_proj.bias", "local_former.layers.3.self_attn.in_proj_bias"]
        samp += ["local_former.layers.5.linear1.weight", "local_former.layers.3.self_attn.in_proj_weight", "global_former.layers.2.linear2.weight",
                 "local_former.layers.2.linear2.weight"]
    for k in whole + [head]:
        out["grad__" + k.replace(".", "__")] = grads[k]
    for k in samp:
        out["gsample__" + k.replace(".", "__")] = sample(grads[k].numpy())
    res = {}
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        res[k] = np.asarray(v)
    res["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, fname + ".npz"), **res)
    print("wrote", kind, "params", int(res["n_params"]), "pred", res["pred"].shape, "loss", float(res["loss"]))


if __name__ == "__main__":
    for kind in sys.argv[1:] or ["jta", "jrdb"]:
        run(kind)
