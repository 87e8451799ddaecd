"""Golden vectors for the multi-modal evaluation / LocoVal filter, produced by running
the reference's own `evaluate_ade_fde` (social-transmotion/evaluate_jta.py:140-500) in this container.

    python tests/golden/gen_golden_eval.py        ->  tests/golden/eval_filter.npz

The reference function returns nothing: its results are the numbers it logs.  A recording logger captures them.
Model and LocoVal weights are the ones of the committed `predictor_multi` fixture (reduced-width reference model,
4 modes); matplotlib / progress / tqdm output is mocked by the import shim (plots are out of scope).  The random
baseline uses python's `random.randint`; the seed's draws are stored so the product can replay them.
"""
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim as shim  # noqa: E402

shim._MOCK_ROOTS.extend(["matplotlib", "pyemd"])
shim.install_predictor()
import numpy as np  # noqa: E402
import torch  # noqa: E402


class RecordingLogger:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))


def main():
    import evaluate_jta as EV
    import model_jta as M
    from learning.value_pose_net import ValuePoseNet
    EV.tqdm.tqdm = lambda it, **kw: it                      # no progress bar
    import types
    # On the reference's target device (cuda) `coords.to(DEVICE)` inside batch_process_coords copies the batch, so the
    # `primary_init_pose` view taken before it keeps the raw pose.  On cpu `.to` aliases and the in-place normalisation
    # would zero that view; clone to reproduce the device-copy semantics the reference runs with.
    _bpc = EV.batch_process_coords
    EV.batch_process_coords = lambda coords, *a, **k: _bpc(coords.clone(), *a, **k)
    EV.args = types.SimpleNamespace(valueloss=True)         # the function reads the script's module-level `args`
    g0 = np.load(os.path.join(HERE, "predictor_multi.npz"))
    J = 49
    model = M.TransMotionJTA(tok_dim=453, nhid=32, nhead=4, dim_feedfwd=64, nlayers_local=2, nlayers_global=1, nmode=4,
                             output_scale=1, obs_and_pred=21, num_tokens=J, device="cpu", multi_modal=True).float()
    model.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(g0[k]) for k in g0.files if k.startswith("sd__")})
    vnet = ValuePoseNet(use_pose=True, use_vel=True)
    vnet.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(g0[k]) for k in g0.files if k.startswith("vn__")})
    vnet.eval()
    g = torch.Generator().manual_seed(21)
    batches = []
    for B, N in ((5, 3), (4, 2)):
        joints = torch.randn(B, N, 21, J, 4, generator=g) * 0.5
        joints[:, :, :, 0, :2] = torch.cumsum(torch.randn(B, N, 21, 2, generator=g) * 0.6, dim=2)
        masks = torch.ones(B, N, 21, J)
        pm = torch.zeros(B, N, dtype=torch.bool)
        pm[0, N - 1] = True
        batches.append((joints, masks, pm))
    out = {}
    for thr in (0.5, 0.52):
        cfg = {"DEVICE": "cpu", "TRAIN": {"input_track_size": 9, "output_track_size": 12}, "NOISY_TRAJ": 0,
               "MODEL": {"value_threshold": thr, "valuenet_checkpoint": ""}}
        random.seed(1234)
        log = RecordingLogger()
        EV.evaluate_ade_fde(model, vnet, "test", "traj+all", [(j.clone(), m.clone(), p.clone()) for j, m, p in batches], 5, cfg, log,
                            "golden", return_all=True, visualize=False, limit_obs=0)
        text = "\n".join(log.lines)
        tag = f"thr{int(thr * 100)}"

        def num(label):
            m = re.search(r"(?:^|\n)\s*" + re.escape(label) + r":\s*([-0-9.eE+naninf]+)", text)
            assert m, (label, text)
            return float(m.group(1))

        for key, label in (("ade", "ADE"), ("fde", "FDE"), ("min_ade", "Min ADE"), ("min_fde", "Min FDE"), ("worst_ade", "Worst ADE"),
                           ("worst_fde", "Worst FDE"), ("iye", "IYE"), ("ade_value", "ADE with Value sampling"),
                           ("fde_value", "FDE with Value sampling"), ("ade_random", "ADE with Random sampling"),
                           ("fde_random", "FDE with Random sampling"), ("minade_value", "Min ADE with Value sampling"),
                           ("minfde_value", "Min FDE with Value sampling"), ("ade_rejected", "ADE of rejected samples"),
                           ("fde_rejected", "FDE of rejected samples"), ("chi_velocity", "Velocity"), ("chi_acceleration", "Acceleration"),
                           ("chi_ang_velocity", "Angular velocity"), ("chi_ang_acceleration", "Angular acceleration"),
                           ("samples", "Total samples"), ("value_mean", "Value"), ("value_gt_mean", "Value GT"),
                           ("value_loss_mean", "Value Loss"), ("value_loss_gt_mean", "Value Loss GT")):
            out[f"{tag}.{key}"] = np.float64(num(label))
        m = re.search(r"DES: \[([^\]]+)\]", text)
        out[f"{tag}.des"] = np.array([float(x) for x in m.group(1).split()])
        out[f"{tag}.log"] = np.array(text)
    random.seed(1234)
    out["random_ids"] = np.array([random.randint(0, 3) for _ in range(sum(b[0].shape[0] for b in batches))])
    for i, (j, m, p) in enumerate(batches):
        out[f"batch{i}.joints"] = j.numpy()
        out[f"batch{i}.masks"] = m.numpy()
        out[f"batch{i}.padding_mask"] = p.numpy()
    out["torch_version"] = np.array(torch.__version__)
    path = os.path.join(HERE, "eval_filter.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    print(out["thr50.log"])


if __name__ == "__main__":
    main()
