"""Golden for the AMASS-pickle motion path: the reference's own loader (pacer/pacer/utils/motion_lib_smpl.py:84-131
`load_motion_with_skeleton`: poselib SkeletonState.from_rotation_and_root_translation -> SkeletonMotion.from_skeleton_state ->
compute_motion_dof_vels) run on two synthetic clips in the on-disk format convert_amass_isaac.py writes (`pose_quat_global` (T,24,4),
`root_trans_offset` (T,3), `pose_aa` (T,72), `beta`, `gender`, `fps`), with the skeleton tree of the shipped smpl_humanoid.xml and
fix_height = False (the height fix needs the licensed SMPL mesh).

    python tests/golden/gen_golden_motion.py   ->  tests/golden/motion_amass.npz

The fixture holds the clips (inputs) and the per-frame cache the reference builds from them: global translation / rotation, local
rotation, global linear / angular velocity (np.gradient resp. quaternion differences, both gaussian-filtered, skeleton3d.py:1249-1272),
joint velocities (motion_lib_smpl.py:44-67), the skeleton's local translations.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim as shim  # noqa: E402

shim.install_pacer()
import numpy as np  # noqa: E402
import torch  # noqa: E402

from poselib.poselib.skeleton.skeleton3d import SkeletonTree  # noqa: E402
from utils import motion_lib_smpl as ml  # noqa: E402


def make_clip(T, fps, seed):
    """smooth random joint rotations on the humanoid's tree -> global rotations by FK of the rotations alone"""
    g = torch.Generator().manual_seed(seed)
    parents = [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]
    t = torch.arange(T, dtype=torch.float64) / fps
    amp = torch.rand(24, 3, generator=g, dtype=torch.float64) * 0.6
    freq = torch.rand(24, 3, generator=g, dtype=torch.float64) * 1.5 + 0.3
    ph = torch.rand(24, 3, generator=g, dtype=torch.float64) * 6.28
    aa = amp[None] * torch.sin(2 * np.pi * freq[None] * t[:, None, None] + ph[None])          # (T,24,3) rotation vectors
    aa[:, 0] *= 0.3
    ang = aa.norm(dim=-1, keepdim=True).clamp(min=1e-12)
    lq = torch.cat([aa / ang * torch.sin(ang / 2), torch.cos(ang / 2)], -1)                     # xyzw

    def qmul(a, b):
        x1, y1, z1, w1 = a.unbind(-1); x2, y2, z2, w2 = b.unbind(-1)
        return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                            w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)
    gq = torch.zeros(T, 24, 4, dtype=torch.float64)
    for b in range(24):
        gq[:, b] = lq[:, b] if parents[b] < 0 else qmul(gq[:, parents[b]], lq[:, b])
    trans = torch.stack([1.2 * t + 0.1 * torch.sin(3 * t), 0.4 * torch.sin(1.1 * t), 0.92 + 0.02 * torch.sin(7 * t)], -1)
    return {"pose_quat_global": gq.numpy().astype(np.float64), "root_trans_offset": trans.float(), "pose_aa": aa.reshape(T, 72).numpy(),
            "beta": np.linspace(-1, 1, 16).astype(np.float64) * (seed % 3), "gender": "neutral", "fps": fps}


if __name__ == "__main__":
    tree = SkeletonTree.from_mjcf(os.path.join(shim.REF, "pacer/pacer/data/assets/mjcf/smpl_humanoid.xml"))
    clips = [make_clip(45, 30, 1), make_clip(38, 60, 2)]
    gb = [torch.zeros(17), torch.cat([torch.zeros(1), torch.linspace(-1, 1, 16)])]
    res = ml.load_motion_with_skeleton(np.arange(2), clips, [tree, tree], gb, False, None, None, None, 0)
    out = {"node_names": np.array("\n".join(tree.node_names)), "parent_indices": tree.parent_indices.numpy(),
           "local_translation": tree.local_translation.numpy(), "torch_version": np.array(torch.__version__)}
    for i in range(2):
        f, mo = res[i]
        out[f"c{i}_pose_quat_global"] = clips[i]["pose_quat_global"]
        out[f"c{i}_root_trans_offset"] = clips[i]["root_trans_offset"].numpy()
        out[f"c{i}_pose_aa"] = clips[i]["pose_aa"]
        out[f"c{i}_beta"] = clips[i]["beta"]
        out[f"c{i}_fps"] = np.array(clips[i]["fps"])
        out[f"c{i}_gts"] = mo.global_translation.numpy()
        out[f"c{i}_grs"] = mo.global_rotation.numpy()
        out[f"c{i}_lrs"] = mo.local_rotation.numpy()
        out[f"c{i}_gvs"] = mo.global_velocity.numpy()
        out[f"c{i}_gavs"] = mo.global_angular_velocity.numpy()
        out[f"c{i}_dvs"] = mo.dof_vels.numpy()
    np.savez_compressed(os.path.join(HERE, "motion_amass.npz"), **out)
    print("wrote motion_amass.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})
