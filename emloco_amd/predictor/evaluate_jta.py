"""Multi-modal evaluation with the LocoVal filter, vectorised on the device.

Mirror of /root/reference/social-transmotion/evaluate_jta.py:22-35 (`inference`) and :140-500 (`evaluate_ade_fde`), and
of the JRDB variant evaluate_jrdb.py:60-330 (pose tokens `2:` with x negated, threshold 0.8 hard-coded at :206).
The reference walks samples and modes in Python and calls LocoVal with batch 1, 2 x modes times per sample; here one
batch is: one model forward, one batched distance computation, ONE LocoVal launch over B x 2M trajectories, masked
reductions for the filter -- no per-sample host work.

Two arithmetic quirks of the reference loop are reproduced by default so that its logged numbers are matched on the
same inputs (each can be switched off):
  * `reference_gt_shift`: inside the mode loop `gt_xy` gets the origin prepended after mode 0 (:291), so modes 1..M-1
    are scored against the ground truth shifted by one frame (ADE / DES; FDE uses `[-1]` and is unaffected);
  * `reference_inplace_pose`: `ValuePoseNet._rotate_normalization` rotates the caller's pose in place
    (value_pose_net.py:98), so within one sample the pose handed to call c is already rotated by the heading angles
    of all earlier calls (order: pred mode 0, gt, pred mode 1, gt, ...).
Plots / pickles of the reference are out of scope.
"""
import math

import numpy as np
import torch

from .dataset_jrdb import batch_process_coords as batch_process_coords_jrdb
from .train_jta import batch_process_coords

DELTA_T = 0.4      # 2.5 fps (utils/metrics.py:69)


def inference(model, config, input_joints, padding_mask, out_len=14, limit_obs=False):
    """evaluate_jta.py:22-35."""
    model.eval()
    with torch.no_grad():
        if torch.isnan(input_joints).any():
            input_joints = torch.where(torch.isnan(input_joints), torch.zeros_like(input_joints), input_joints)
        if config.get("NOISY_TRAJ", 0):
            input_joints[:, :, 0, :2] = input_joints[:, :, 0, :2] + torch.randn_like(input_joints[:, :, 0, :2]) * config["NOISY_TRAJ"]
        pred_joints = model(input_joints, padding_mask, limit_obs=limit_obs)
    return pred_joints[:, -out_len:]


def calculate_initial_yaw_error(group_A, group_B):
    """utils/metrics.py:49-67."""
    norm_A = torch.norm(group_A, dim=1, keepdim=True)
    norm_B = torch.norm(group_B, dim=1, keepdim=True)
    nA = torch.where(norm_A > 0, group_A / norm_A, group_A)
    nB = torch.where(norm_B > 0, group_B / norm_B, group_B)
    return torch.acos((nA * nB).sum(1).clamp(-1, 1))


def motion_primitives(xy):
    """utils/metrics.py:69-110 for a stack of trajectories xy (..., T, 2): speed, |accel|, |heading rate|, |heading accel|."""
    d = xy[..., 1:, :] - xy[..., :-1, :]
    vel = torch.linalg.norm(d / DELTA_T, dim=-1)
    acc = ((vel[..., 1:] - vel[..., :-1]) / DELTA_T).abs()
    ang = (torch.atan2(d[..., 1], d[..., 0]) / DELTA_T).abs()
    ang_acc = ((ang[..., 1:] - ang[..., :-1]) / DELTA_T).abs()
    return {"velocity": vel, "acceleration": acc, "ang_velocity": ang, "ang_acceleration": ang_acc}


def _chi_from_counts(gt_counts, pred_counts):
    gt_dens, pred_dens = gt_counts / max(gt_counts.sum(), 1), pred_counts / max(pred_counts.sum(), 1)
    s = gt_dens + pred_dens
    nz = s != 0
    return float((((gt_dens - pred_dens) ** 2)[nz] / s[nz]).sum())


def calculate_chi_distance(gt_primitive, pred_primitive, num_bins=20, reduce=None):
    """utils/metrics.py:112-145: chi-square distance of the 20-bin histograms over the joint value range.
    `density=True` histograms times the bin width are count fractions, so ranks only need to agree on the range
    (`reduce("min"/"max", x)`) and add their counts (`reduce("sum", x)`) to get the single-process result."""
    out = {}
    for key in gt_primitive:
        gt_values, pred_values = np.asarray(gt_primitive[key], np.float64), np.asarray(pred_primitive[key], np.float64)
        lo, hi = min(gt_values.min(), pred_values.min()), max(gt_values.max(), pred_values.max())
        if reduce is not None:
            lo, hi = float(reduce("min", np.array([lo]))[0]), float(reduce("max", np.array([hi]))[0])
        bins = np.linspace(lo, hi, num_bins + 1)
        gt_counts = np.histogram(gt_values, bins=bins)[0].astype(np.float64)
        pred_counts = np.histogram(pred_values, bins=bins)[0].astype(np.float64)
        if reduce is not None:
            gt_counts, pred_counts = reduce("sum", gt_counts), reduce("sum", pred_counts)
        out[key] = _chi_from_counts(gt_counts, pred_counts)
    return out


def _dist_reduce(op, arr):
    """all-reduce a float64 numpy array over the default process group (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).clone()
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
    return t.cpu().numpy()


def _heading(traj):
    """angle of _rotate_normalization (value_pose_net.py:76-85) for traj (..., 13, 2)."""
    x, y = traj[..., 1, 0], traj[..., 1, 1]
    x = torch.where(x.abs() < 1e-10, torch.full_like(x, 1e-10), x)
    return torch.atan2(y, x)


def _rotate_xy(v, ang):
    """bmm(v, R(ang)) with R = [[c, -s], [s, c]]: (x, y) -> (x c + y s, -x s + y c)."""
    c, s = torch.cos(ang), torch.sin(ang)        # ang has the shape of v[..., 0]
    x, y = v[..., 0], v[..., 1]
    return torch.stack([x * c + y * s, -x * s + y * c], dim=-1)


class EvalAccumulator:
    """Running sums of evaluate_ade_fde; `update` consumes one batch on the device, `summary` gives the logged numbers."""

    def __init__(self, filter_threshold=0.7, reference_gt_shift=True, reference_inplace_pose=True):
        self.thr = float(filter_threshold)
        self.gt_shift, self.inplace_pose = reference_gt_shift, reference_inplace_pose
        self.s = {k: 0.0 for k in ("ade", "fde", "ade_min", "fde_min", "ade_max", "fde_max", "iye", "ade_value", "fde_value",
                                   "ade_random", "fde_random", "minade_value", "minfde_value", "ade_filtered", "fde_filtered",
                                   "value", "value_gt", "value_loss", "value_loss_gt")}
        self.des = np.zeros(12)
        self.n = {"sample": 0, "value_sampling": 0, "filtered": 0, "random": 0, "values": 0}
        self.gt_prim = {k: [] for k in ("velocity", "acceleration", "ang_velocity", "ang_acceleration")}
        self.pred_prim = {k: [] for k in self.gt_prim}
        self.num_mode = 1
        self.values, self.ades, self.fdes = [], [], []

    @torch.no_grad()
    def update(self, in_joints, out_joints, pred_joints, primary_init_pose, valuenet=None, random_ids=None, dataset="jta"):
        dev = pred_joints.device
        B = out_joints.shape[0]
        pred = pred_joints.reshape(B, 12, -1, 2).float()
        M = pred.shape[2]
        self.num_mode = M
        gt = out_joints[:, :, 0, :2].float()                                      # (B,12,2)
        self.s["iye"] += float(calculate_initial_yaw_error(gt[:, 0], pred[:, 0, 0]).sum())
        gt_m = gt[:, :, None, :].expand(B, 12, M, 2)
        if self.gt_shift and M > 1:
            shifted = torch.cat([torch.zeros(B, 1, 2, device=dev), gt[:, :11]], 1)
            gt_m = torch.cat([gt[:, :, None, :], shifted[:, :, None, :].expand(B, 12, M - 1, 2)], 2)
        des = torch.linalg.norm(gt_m - pred, dim=-1).double()                     # (B,12,M)
        ade = des.mean(1)                                                         # (B,M)
        fde = torch.linalg.norm(gt[:, -1, None, :] - pred[:, -1], dim=-1).double()
        self.s["ade"] += float(ade.mean(1).sum())
        self.s["fde"] += float(fde.mean(1).sum())
        self.s["ade_min"] += float(ade.min(1)[0].sum())
        self.s["fde_min"] += float(fde.min(1)[0].sum())
        self.s["ade_max"] += float(ade.max(1)[0].sum())
        self.s["fde_max"] += float(fde.max(1)[0].sum())
        self.des += des.mean(2).sum(0).cpu().numpy()
        self.n["sample"] += B
        for k, v in motion_primitives(gt).items():
            self.gt_prim[k].append(v.reshape(-1))
        for k, v in motion_primitives(pred.permute(0, 2, 1, 3)).items():
            self.pred_prim[k].append(v.reshape(-1))
        if valuenet is None:
            return
        # ---- LocoVal over every (sample, mode) prediction and the ground truth, one launch
        init_pose = primary_init_pose.to(dev).float().clone()
        if dataset == "jta":
            init_pose[..., 2] = -init_pose[..., 2]                                # evaluate_jta.py:221
        else:
            init_pose[..., 0] = -init_pose[..., 0]                                # evaluate_jrdb.py:108
        init_vel = ((in_joints[:, 8, 0, :2] - in_joints[:, 7, 0, :2]) * 2.5).to(dev).float()
        zero = torch.zeros(B, 1, 2, device=dev)
        gt_traj = torch.cat([zero, gt], 1)                                        # (B,13,2)
        pred_traj = torch.cat([zero[:, :, None, :].expand(B, 1, M, 2), pred], 1).permute(0, 2, 1, 3)   # (B,M,13,2)
        # call order of the reference loop: pred 0, gt, pred 1, gt, ...
        calls = torch.stack([pred_traj, gt_traj[:, None].expand(B, M, 13, 2)], 2).reshape(B, 2 * M, 13, 2)
        valid = ~(torch.isnan(init_pose).reshape(B, -1).any(1)[:, None] | torch.isnan(pred_traj).reshape(B, M, -1).any(2))   # (B,M)
        pose_c = init_pose[:, None].expand(B, 2 * M, 24, 3).clone()
        if self.inplace_pose:
            theta = _heading(calls) * valid.repeat_interleave(2, 1)               # skipped calls do not rotate the pose
            before = torch.cumsum(theta, 1) - theta                               # rotation already applied when call c starts
            pose_c[..., :2] = _rotate_xy(pose_c[..., :2], before[:, :, None].expand(B, 2 * M, 24))
            hidden = before.new_zeros(B, 2 * M, dtype=torch.bool)
            hidden[:, 1:] = (torch.cumsum(valid.repeat_interleave(2, 1).int(), 1)[:, :-1] > 0)
            for j in (4, 8, 9, 10, 11):                                           # joints zeroed by the first executed call
                pose_c[:, :, j] = torch.where(hidden[:, :, None], torch.zeros_like(pose_c[:, :, j]), pose_c[:, :, j])
        flat = lambda t: t.reshape(B * 2 * M, *t.shape[2:]).contiguous()
        safe_calls = torch.nan_to_num(flat(calls))
        safe_pose = torch.nan_to_num(flat(pose_c))
        vel_c = torch.nan_to_num(flat(init_vel[:, None].expand(B, 2 * M, 2)))
        was_inplace = getattr(valuenet, "inplace_pose", False)
        valuenet.inplace_pose = False
        values = valuenet(safe_calls, safe_pose, vel_c).reshape(B, M, 2).double()
        valuenet.inplace_pose = was_inplace
        v_pred, v_gt = values[..., 0], values[..., 1]
        nv = int(valid.sum())
        self.n["values"] += nv
        self.s["value"] += float((v_pred * valid).sum())
        self.s["value_gt"] += float((v_gt * valid).sum())
        self.s["value_loss"] += float((((v_pred - 1) ** 2) * valid).sum())
        self.s["value_loss_gt"] += float((((v_gt - 1) ** 2) * valid).sum())
        if M <= 1:
            return
        self.values.append(v_pred[valid].cpu()); self.ades.append(ade[valid].cpu()); self.fdes.append(fde[valid].cpu())
        has = valid.any(1)                                                        # samples with at least one scored mode
        big = 1e30
        if random_ids is None:
            random_ids = torch.randint(0, M, (B,), device=dev)
        random_ids = torch.as_tensor(random_ids, device=dev).long()
        self.s["ade_random"] += float((ade.gather(1, random_ids[:, None])[:, 0] * has).sum())
        self.s["fde_random"] += float((fde.gather(1, random_ids[:, None])[:, 0] * has).sum())
        self.n["random"] += int(has.sum())
        kept = valid & (v_pred >= self.thr)
        rejected = valid & (v_pred < self.thr)
        any_kept = kept.any(1)
        best = torch.where(valid, v_pred, torch.full_like(v_pred, -big)).argmax(1)
        ade_best, fde_best = ade.gather(1, best[:, None])[:, 0], fde.gather(1, best[:, None])[:, 0]
        fallback = has & ~any_kept                                                # nothing passes: the arg-max value mode stands in
        self.s["ade_value"] += float((ade * kept).sum() + (ade_best * fallback).sum())
        self.s["fde_value"] += float((fde * kept).sum() + (fde_best * fallback).sum())
        self.n["value_sampling"] += int(kept.sum()) + int(fallback.sum())
        min_ade_kept = torch.where(kept, ade, torch.full_like(ade, big)).min(1)[0]
        min_fde_kept = torch.where(kept, fde, torch.full_like(fde, big)).min(1)[0]
        self.s["minade_value"] += float(torch.where(any_kept, min_ade_kept, ade_best * fallback).sum())
        self.s["minfde_value"] += float(torch.where(any_kept, min_fde_kept, fde_best * fallback).sum())
        self.s["ade_filtered"] += float((ade * rejected).sum() + (ade_best * fallback).sum())      # evaluate_jta.py:339-340,356-359
        self.s["fde_filtered"] += float((fde * rejected).sum() + (fde_best * fallback).sum())
        self.n["filtered"] += int(rejected.sum())

    def summary(self, distributed=None):
        """The numbers evaluate_ade_fde logs.  With `distributed` (default: a process group with world_size > 1 exists) the
        running sums, histogram counts and per-bin statistics are all-reduced first, so every rank returns the result of
        the whole data-parallel evaluation (eval shards by batch, only scalars are exchanged)."""
        import torch.distributed as dist
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        red = _dist_reduce if distributed else None
        s, cnt, des, num_mode = dict(self.s), dict(self.n), self.des.copy(), self.num_mode
        if distributed:
            keys_s, keys_n = sorted(s), sorted(cnt)
            flat = _dist_reduce("sum", np.array([s[k] for k in keys_s] + [cnt[k] for k in keys_n] + list(des)))
            s = dict(zip(keys_s, flat[:len(keys_s)]))
            cnt = {k: int(round(v)) for k, v in zip(keys_n, flat[len(keys_s):len(keys_s) + len(keys_n)])}
            des = flat[len(keys_s) + len(keys_n):]
            num_mode = int(_dist_reduce("max", np.array([float(num_mode)]))[0])
        n = max(cnt["sample"], 1)
        out = {"samples": cnt["sample"], "ade": s["ade"] / n, "fde": s["fde"] / n, "min_ade": s["ade_min"] / n,
               "min_fde": s["fde_min"] / n, "worst_ade": s["ade_max"] / n, "worst_fde": s["fde_max"] / n,
               "iye": s["iye"] / n, "des": des / n}
        if self.gt_prim["velocity"]:
            gp = {k: torch.cat(v).cpu().numpy() for k, v in self.gt_prim.items()}
            pp = {k: torch.cat(v).cpu().numpy() for k, v in self.pred_prim.items()}
            out.update({"chi_" + k: v for k, v in calculate_chi_distance(gp, pp, reduce=red).items()})
        if cnt["values"]:
            nv = cnt["values"]
            out.update({"value_mean": s["value"] / nv, "value_gt_mean": s["value_gt"] / nv,
                        "value_loss_mean": s["value_loss"] / nv, "value_loss_gt_mean": s["value_loss_gt"] / nv})
        if num_mode > 1 and cnt["values"]:
            d = lambda a, b: a / b if b > 0 else 0.0
            out.update({"threshold": self.thr, "ade_value": d(s["ade_value"], cnt["value_sampling"]),
                        "fde_value": d(s["fde_value"], cnt["value_sampling"]),
                        "ade_random": d(s["ade_random"], cnt["sample"]), "fde_random": d(s["fde_random"], cnt["sample"]),
                        "minade_value": d(s["minade_value"], cnt["sample"]), "minfde_value": d(s["minfde_value"], cnt["sample"]),
                        "ade_rejected": d(s["ade_filtered"], cnt["filtered"]), "fde_rejected": d(s["fde_filtered"], cnt["filtered"])})
            # plausibility-score bins (evaluate_jta.py:432-449): mean ADE / FDE per 0.1-wide value bin, value histogram
            if self.values:
                v, a, f = torch.cat(self.values).numpy(), torch.cat(self.ades).numpy(), torch.cat(self.fdes).numpy()
            else:
                v = a = f = np.zeros(0)
            idx = np.digitize(v, np.arange(0, 1.05, 0.1))
            per_bin = np.stack([np.array([x[idx == i].sum() for i in range(1, 11)]) for x in (a, f, np.ones_like(v))])
            hist = np.histogram(v, bins=10, range=(0, 1))[0].astype(np.float64)
            if distributed:
                per_bin, hist = _dist_reduce("sum", per_bin), _dist_reduce("sum", hist)
            with np.errstate(invalid="ignore", divide="ignore"):
                out["ade_per_value_bin"] = np.where(per_bin[2] > 0, per_bin[0] / per_bin[2], np.nan)
                out["fde_per_value_bin"] = np.where(per_bin[2] > 0, per_bin[1] / per_bin[2], np.nan)
            out["value_hist"] = hist.astype(np.int64)
        return out


def evaluate_ade_fde(model, valuenet, split, modality_selection, dataloader, bs, config, logger=None, exp_name="", return_all=False,
                     visualize=False, limit_obs=False, dataset="jta", random_ids=None, shard=None, **acc_kw):
    """Same arguments as the reference; additionally returns the summary dict (the reference only logs it).
    Data-parallel (configs[4]): with a process group of W ranks, rank r evaluates batches r, r+W, ... (`shard=(r, W)`,
    default from torch.distributed) and the summary is all-reduced -- no tensor data crosses ranks."""
    import torch.distributed as dist
    if shard is None:
        shard = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    out_F = config["TRAIN"]["output_track_size"]
    thr = 0.8 if dataset == "jrdb" else config["MODEL"]["value_threshold"]
    acc = EvalAccumulator(thr, **acc_kw)
    off = 0
    for bi, batch in enumerate(dataloader):
        joints, masks, padding_mask = batch[0], batch[1], batch[2]
        if bi % shard[1] != shard[0]:
            off += joints.shape[0]
            continue
        padding_mask = padding_mask.to(config["DEVICE"])
        primary_init_pose = joints[:, 0, 8, 3:27, :3] if dataset == "jta" else joints[:, 0, 8, 2:, :3]
        bpc = batch_process_coords if dataset == "jta" else batch_process_coords_jrdb
        in_joints, in_masks, out_joints, out_masks, padding_mask = bpc(joints, masks, padding_mask, config, modality_selection)
        pred_joints = inference(model, config, in_joints, padding_mask, out_len=out_F, limit_obs=limit_obs)
        B = out_joints.shape[0]
        ids = None if random_ids is None else random_ids[off:off + B]
        off += B
        acc.update(in_joints, out_joints, pred_joints, primary_init_pose, valuenet, ids, dataset)
    res = acc.summary()
    if logger is not None:
        logger.info(f"Total samples: {res['samples']}")
        for label, key in (("ADE", "ade"), ("FDE", "fde"), ("Min ADE", "min_ade"), ("Min FDE", "min_fde"), ("Worst ADE", "worst_ade"),
                           ("Worst FDE", "worst_fde"), ("IYE", "iye")):
            logger.info(f"{label}: {res[key]:.5f}")
        logger.info(f"DES: {np.round(res['des'], 5)}")
        if "chi_velocity" in res:
            logger.info(f"Chi-square distance:\\n Velocity: {res['chi_velocity']:.5f},\\n Acceleration: {res['chi_acceleration']:.5f},\\n "
                        f"Angular velocity: {res['chi_ang_velocity']:.5f},\\n Angular acceleration: {res['chi_ang_acceleration']:.5f}")
        if "ade_value" in res:
            logger.info(f"Threadhold: {res['threshold']}")
            for label, key in (("ADE with Value sampling", "ade_value"), ("FDE with Value sampling", "fde_value"),
                               ("ADE with Random sampling", "ade_random"), ("FDE with Random sampling", "fde_random"),
                               ("Min ADE with Value sampling", "minade_value"), ("Min FDE with Value sampling", "minfde_value"),
                               ("ADE of rejected samples", "ade_rejected"), ("FDE of rejected samples", "fde_rejected")):
                logger.info(f"{label}: {res[key]:.5f}")
    return res


# ---------------------------------------------------------------------------------------------- entry point (evaluate_jta.py:509-625)
def build_arg_parser():
    """The flags of the reference's `python evaluate_jta.py` (evaluate_jta.py:511-527), plus --data_root / --out_root / --dataset."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--exp_name", type=str, help="checkpoint path")
    p.add_argument("--split", type=str, default="test", help="Split to use. one of [train, test, valid]")
    p.add_argument("--metric", type=str, default="ade_fde", help="Evaluation metric")
    p.add_argument("--modality", type=str, default="traj+all", help="modality combination, e.g. 'traj', 'traj+3dpose', 'traj+all'")
    p.add_argument("--vis", action="store_true", help="Visualize the predictions (out of scope here: ignored)")
    p.add_argument("--limit_obs", type=int, default=0, help="Limit the number of observations")
    p.add_argument("--valueloss", action="store_true", help="Use value loss")
    p.add_argument("--all_frames", action="store_true", help="Evaluate all observation frames")
    p.add_argument("--noisy_traj", type=float, default=0, help="Add noise to the trajectory to mimic real data")
    p.add_argument("--multi_modal", action="store_true", help="Use multi-modal model")
    p.add_argument("--last_epoch", action="store_true", help="Use last epoch checkpoint")
    p.add_argument("--no_pose", action="store_true", help="No pose in valuenet")
    p.add_argument("--no_vel", action="store_true", help="No velocity in valuenet")
    p.add_argument("--epoch", type=str, default=0, help="Epoch to evaluate")
    p.add_argument("--filter_threshold", type=float, default=0.7, help="Threshold for filtering samples")
    p.add_argument("--data_root", type=str, default="data", help="root of <dataset>/preprocess_smpl/<split>/part_*.pkl")
    p.add_argument("--out_root", type=str, default="experiments", help="root of the experiment directories")
    p.add_argument("--dataset", type=str, default="jta", choices=["jta", "jrdb"], help="evaluate_jta.py | evaluate_jrdb.py")
    return p


def find_checkpoint(args):
    """evaluate_jta.py:537-553: which file of the experiment's checkpoint directory the flags select."""
    import os
    d = os.path.join(args.out_root, args.dataset.upper(), args.exp_name, "checkpoints")
    if args.last_epoch:
        names = ["checkpoint.pth.tar"]
    elif args.epoch != 0:
        names = [f"best_val_checkpoint_{args.epoch}epoch.pth.tar", f"checkpoint_{args.epoch}epoch.pth.tar", f"best_val_{args.epoch}.pth.tar"]
    else:
        names = ["best_val_checkpoint.pth.tar"]
    for n in names:
        if os.path.exists(os.path.join(d, n)):
            return os.path.join(d, n)
    raise FileNotFoundError(f"Checkpoint not found: {names} under {d}")


def run(args, logger=None):
    """evaluate_jta.py:555-625: load the checkpoint and its config, the LocoVal network, the split, and evaluate."""
    import os
    from torch.utils.data import DataLoader
    from ..learning.value_pose_net import ValuePoseNet
    from .train_jta import load_checkpoint, load_config
    dev = f"cuda:{torch.cuda.current_device()}"
    ckpt_name = find_checkpoint(args)
    if logger is not None:
        logger.info(f"Loading checkpoint from {ckpt_name}")
    config = torch.load(ckpt_name, map_location="cpu")["config"]
    new_cfg = load_config(f"configs/{args.dataset}_all_visual_cues.yaml", exp_name=args.exp_name + "_eval", dataset_name=args.dataset.upper(),
                          out_root=args.out_root)
    config["DEVICE"], config["NOISY_TRAJ"], config["MULTI_MODAL"] = dev, args.noisy_traj, args.multi_modal
    config["MODEL"]["value_threshold"] = args.filter_threshold
    valuenet = None
    if args.valueloss:
        valuenet = ValuePoseNet(use_pose=not args.no_pose, use_vel=not args.no_vel)
        vck = config["MODEL"].get("valuenet_checkpoint") or new_cfg["MODEL"].get("valuenet_checkpoint", "")
        if vck:
            valuenet.load_state_dict(torch.load(vck, map_location="cpu"))
        elif logger is not None:
            logger.info("No checkpoint provided for valuenet. Using random weights.")
        valuenet = valuenet.to(dev).eval()
    if args.dataset == "jta":
        from .dataset_jta import collate_batch, create_dataset
        from .model_jta import create_model
    else:
        from .dataset_jrdb import collate_batch, create_dataset
        from .model_jrdb import create_model
    model = create_model(config, logger)
    load_checkpoint(model, ckpt_name, strict=True)
    in_F, out_F = config["TRAIN"]["input_track_size"], config["TRAIN"]["output_track_size"]
    dataset = create_dataset(config["DATA"]["train_datasets"][0], logger, split=args.split, track_size=in_F + out_F, track_cutoff=in_F,
                             preprocessed=config["DATA"]["preprocessed"], root=args.data_root)
    bs = new_cfg["TRAIN"]["batch_size"] * 10
    loader = DataLoader(dataset, batch_size=bs, num_workers=0, shuffle=False, collate_fn=collate_batch)
    out = {}
    for obs_i in ([1, 2, 3, 4, 5, 6, 7, 8, 0] if args.all_frames else [args.limit_obs]):
        if logger is not None:
            logger.info(f"Evaluating with {9 if obs_i == 0 else obs_i} frames")
        out[obs_i] = evaluate_ade_fde(model, valuenet, args.split, args.modality, loader, bs, config, logger, args.exp_name, return_all=True,
                                      limit_obs=obs_i, dataset=args.dataset)
    return out


if __name__ == "__main__":
    import random
    from .. import _lib, configure_runtime
    from ..dist import init_from_env
    configure_runtime()                                  # entry point: 16 hardware queues, ahead of the first GPU call
    from .train_jta import create_logger
    _lib.require_device()
    a = build_arg_parser().parse_args()
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    _rank, local_rank, _world = init_from_env("nccl")
    torch.cuda.set_device(local_rank)
    import os
    logdir = os.path.join(a.out_root, a.dataset.upper(), a.exp_name, "eval_logs")
    os.makedirs(logdir, exist_ok=True)
    run(a, create_logger(logdir))
