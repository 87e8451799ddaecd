"""One `train_jta.py` iteration: batch normalisation, predictor forward, EmLoco loss, backward, clip, Adam.

Mirror of social-transmotion/train_jta.py (compute_loss :98-127, nan_handler :143-165, train loop body :245-320),
dataset_jta.py (batch_process_coords :27-86) and utils/metrics.py (MSE_LOSS / MSE_LOSS_MULTI :4-26).  Host control
flow is the reference's; the arithmetic of the model and of the LocoVal loss runs in the HIP kernels behind
TransMotionJTA / ValuePoseNet.  For data-parallel training `step` all-reduces one flat gradient bucket (RCCL).
"""
import os

import torch

from ..dist import FlatGradBucket, all_reduce_, all_reduce_mean_scalar, barrier, broadcast_parameters, rank, world_size


def MSE_LOSS(output, target, mask=None):
    pred_xy = output[:, :, 0, :2]
    gt_xy = target[:, :, 0, :2]
    norm = torch.norm(pred_xy - gt_xy, p=2, dim=-1)
    return torch.mean(torch.mean(norm, dim=-1)) * 100


def MSE_LOSS_MULTI(output, target, mask=None):
    pred_xys = output[:, :, :, :2]
    gt_xys = target[:, :, 0, :2].unsqueeze(2).repeat(1, 1, pred_xys.size(2), 1)
    norm = torch.norm(pred_xys - gt_xys, p=2, dim=-1)
    mean_K = torch.mean(norm, dim=1)
    return torch.mean(torch.min(mean_K, dim=1)[0]) * 100


def batch_process_coords(coords, masks, padding_mask, config, modality_selection='traj+all', training=False, multiperson=True):
    joints = coords.to(config["DEVICE"]).clone()
    masks = masks.to(config["DEVICE"])
    in_F = config["TRAIN"]["input_track_size"]
    joints[:, :, :, 0] = joints[:, :, :, 0] - joints[:, 0:1, (in_F - 1):in_F, 0]          # primary pelvis at t = in_F-1
    joints[:, :, :, 1:3] = joints[:, :, :, 1:3] - joints[:, :, (in_F - 1):in_F, 1:3]
    joints[:, :, :, 3:27] = joints[:, :, :, 3:27] - joints[:, :, (in_F - 1):in_F, 3:27]
    joints[:, :, :, 27:] = joints[:, :, :, 27:] - joints[:, :, (in_F - 1):in_F, 27:]
    B, N, F, J, K = joints.shape
    if not training:
        sel = {'traj+all': [], 'traj': [slice(1, None)], 'traj+2dbox': [slice(1, 2), slice(3, None)],
               'traj+3dpose': [slice(1, 3), slice(27, None)], 'traj+2dpose': [slice(1, 27)],
               'traj+3dpose+3dbox': [slice(2, 3), slice(27, None)], 'traj+2dpose+3dpose': [slice(1, 3)]}
        if modality_selection not in sel:
            raise ValueError('modality error')
        for s in sel[modality_selection]:
            joints[:, :, :, s] = 0
    joints = joints.transpose(1, 2).reshape(B, F, N * J, K)
    masks = masks.transpose(1, 2).reshape(B, F, N * J)
    out_F = config["TRAIN"]["output_track_size"]
    return (joints[:, :in_F].float(), masks[:, :in_F].float(), joints[:, in_F:in_F + out_F].float(),
            masks[:, in_F:in_F + out_F].float(), padding_mask.float())


def nan_handler(pred_traj, init_pose, init_vel):
    if len(pred_traj.shape) == 3:
        nan_traj = torch.isnan(pred_traj).any(dim=1).any(dim=1)
    else:
        nan_traj = torch.isnan(pred_traj).any(dim=1).any(dim=1).any(dim=1)
    nan_mask = nan_traj | torch.isnan(init_pose).any(dim=1).any(dim=1) | torch.isnan(init_vel).any(dim=1)
    if nan_mask.any():
        pred_traj, init_pose, init_vel = pred_traj[~nan_mask], init_pose[~nan_mask], init_vel[~nan_mask]
    zero = torch.all(init_pose == 0, dim=1).all(dim=1)
    if zero.any():
        pred_traj, init_pose, init_vel = pred_traj[~zero], init_pose[~zero], init_vel[~zero]
    return pred_traj, init_pose, init_vel


def compute_loss(model, config, in_joints, out_joints, in_masks, out_masks, padding_mask, mode='val', limit_obs=False,
                 random_masking=None):
    in_F = in_joints.shape[1]
    if random_masking is None:
        random_masking = mode == 'train'
    # train_jta.py:102-103 replaces NaN inputs by zero when there are any; done unconditionally (same result, no host read)
    in_joints = torch.where(torch.isnan(in_joints), torch.zeros_like(in_joints), in_joints)
    if config.get("NOISY_TRAJ"):      # train_jta.py:115-117: gaussian noise on the primary track, inputs and targets (in place, as there)
        in_joints[:, :, 0, :2] = in_joints[:, :, 0, :2] + torch.randn_like(in_joints[:, :, 0, :2]) * config["NOISY_TRAJ"] ** 2
        out_joints[:, :, 0, :2] = out_joints[:, :, 0, :2] + torch.randn_like(out_joints[:, :, 0, :2]) * config["NOISY_TRAJ"] ** 2
    pred = model(in_joints, padding_mask, random_masking, limit_obs=limit_obs, frame_masking=config.get('USE_FRAME_MASK', False))
    loss_fn = MSE_LOSS_MULTI if config.get("MULTI_MODAL", False) else MSE_LOSS
    return loss_fn(pred[:, in_F:], out_joints, out_masks), pred


def emloco_loss(config, valuenet, pred_joints, primary_init_pose, primary_init_vel, in_F):
    """train_jta.py:288-308."""
    dev = pred_joints.device
    w = config["TRAIN"].get("valuenet_weight", 1.0)
    if config.get("MULTI_MODAL", False):
        pred_trajs = pred_joints[:, in_F:]
        pred_trajs = torch.cat([torch.zeros(pred_trajs.size(0), 1, pred_trajs.size(2), 2, device=dev), pred_trajs], dim=1)
        pred_trajs, pose, vel = nan_handler(pred_trajs, primary_init_pose, primary_init_vel)
        total = 0
        for i in range(pred_trajs.size(2)):
            _, vl = valuenet.calc_embodied_motion_loss(pred_trajs[:, :, i].contiguous(), pose, vel)
            total = total + vl
        total = total * w
        return total / pred_trajs.size(2) if pred_trajs.size(2) else 0
    pred_traj = pred_joints[:, in_F:].squeeze(2)
    pred_traj = torch.cat([torch.zeros(pred_traj.size(0), 1, 2, device=dev), pred_traj], dim=1)
    pred_traj, pose, vel = nan_handler(pred_traj, primary_init_pose, primary_init_vel)
    _, vl = valuenet.calc_embodied_motion_loss(pred_traj, pose, vel)
    return vl * w


def emloco_loss_masked(config, valuenet, pred_joints, primary_init_pose, primary_init_vel, in_F):
    """The same loss as `emloco_loss` (train_jta.py:143-165,288-308) without data-dependent shapes: instead of dropping the
    rows `nan_handler` drops (NaN trajectory / pose / velocity, all-zero pose) they get weight 0 in the mean -- identical
    value and gradients, no host read.  Returns (sum over kept rows of the per-row loss, averaged over modes and scaled by
    valuenet_weight; number of kept rows): the caller divides, so that data-parallel ranks can divide by the global count.
    A batch with no kept row contributes 0, where the reference's NaN mean is skipped (:308); so does a batch in which LocoVal
    returns NaN for a kept row."""
    dev = pred_joints.device
    w = config["TRAIN"].get("valuenet_weight", 1.0)
    multi = config.get("MULTI_MODAL", False)
    pred = pred_joints[:, in_F:]                                                    # (B, 12, M, 2)
    B, _, M, _ = pred.shape
    pred = torch.cat([torch.zeros(B, 1, M, 2, device=dev), pred], dim=1)           # (B, 13, M, 2)
    if not multi:
        pred = pred[:, :, :1]
        M = 1
    bad = torch.isnan(pred).flatten(1).any(1) | torch.isnan(primary_init_pose).flatten(1).any(1) | torch.isnan(primary_init_vel).any(1)
    bad = bad | (primary_init_pose == 0).flatten(1).all(1)
    keep = (~bad).float()
    kb = ~bad
    pred = torch.where(kb[:, None, None, None], pred, torch.zeros_like(pred))
    pose = torch.where(kb[:, None, None], primary_init_pose, torch.zeros_like(primary_init_pose))
    vel = torch.where(kb[:, None], primary_init_vel, torch.zeros_like(primary_init_vel))
    total = 0
    any_nan = torch.zeros((), dtype=torch.bool, device=dev)
    for i in range(M):                # the LocoVal call rotates `pose` in place, cumulatively over the modes (value_pose_net.py:97)
        value = valuenet(pred[:, :, i].contiguous(), pose, vel).reshape(-1)
        # a NaN value of a kept row makes the reference's (scalar) value loss NaN, and the reference then leaves the term out of
        # the step (train_jta.py:308); here the NaN is taken out of the graph and the whole term gets weight 0 -- no host read
        nan_row = torch.isnan(value)
        any_nan = any_nan | (nan_row & kb).any()
        value = torch.where(nan_row, torch.ones_like(value), value)
        total = total + (keep * (value - 1.0) ** 2).sum()
    return total * (~any_nan).float() * (w / M), keep.sum()


class EmLocoTrainer:
    """Adam(lr) + clip_grad_norm_(max_grad_norm) (train_jta.py:317-318,411) around the loss above."""

    def __init__(self, model, valuenet, config, data_parallel=False):
        self.model, self.valuenet, self.config = model, valuenet, config
        if valuenet is not None:
            valuenet.eval()
            for p in valuenet.parameters():
                p.requires_grad_(False)          # the LocoVal weights are frozen while the predictor trains (train_jta.py:197-204)
        if data_parallel:
            broadcast_parameters(model, valuenet)      # replicas start from rank 0's weights (nn.DataParallel has one copy)
        # clip_grad_norm_ + Adam as three launches on flat buffers (fused_adam.py; a torch.optim.Adam in every other respect: same
        # state dict, same param_groups).  EMLOCO_FLAT_ADAM=0, or parameters that are not on a GPU: torch's own.
        params = [p for p in model.parameters() if p.requires_grad]
        self.flat_adam = bool(params) and params[0].device.type == "cuda" and os.environ.get("EMLOCO_FLAT_ADAM", "1") != "0"
        if self.flat_adam:
            from .fused_adam import FlatClipAdam
            self.optimizer = FlatClipAdam(params, lr=config["TRAIN"]["lr"])
            self.bucket = self.optimizer.bucket                # (also what the data-parallel all-reduce travels in)
        else:
            self.bucket = FlatGradBucket(model.parameters()) if data_parallel else None
            self.optimizer = torch.optim.Adam(model.parameters(), lr=config["TRAIN"]["lr"])
        self.data_parallel = bool(data_parallel)
        # the backward pass leaves its gradients on the parameters and ONE launch gathers them into the bucket (dist.FlatGradBucket.release /
        # gather) instead of one accumulation launch per parameter (132 per step); EMLOCO_GATHER_GRADS=0: autograd's accumulation
        self._gather_grads = os.environ.get("EMLOCO_GATHER_GRADS", "1") != "0"

    # what differs between train_jta.py and train_jrdb.py inside the loop body
    process_coords = staticmethod(batch_process_coords)
    value_loss_with_multi_modal = True        # train_jta.py:308 adds the value term in both branches

    def primary_state(self, joints, in_joints):
        """LocoVal's pose / velocity input of the primary agent (train_jta.py:264-274): raw SMPL joints at the last observed
        frame with z flipped, velocity from the last two observed positions at 2.5 fps."""
        pose = joints[:, 0, 8, 3:27, :3].clone().to(self.config["DEVICE"])
        pose[..., 2] *= -1
        vel = ((in_joints[:, 8, 0, :2] - in_joints[:, 7, 0, :2]) * 2.5).clone()
        return pose, vel

    def step(self, joints, masks, padding_mask, modality_selection='traj+all', random_masking=True):
        """One iteration of train_jta.py:245-320 on this rank's batch.  Data parallel: the ranks' batches are the equal
        slices of one global batch; the MSE term is a mean over the batch (each rank adds mean / world), the EmLoco term a
        mean over the rows nan_handler keeps (each rank adds its kept rows' sum / the GLOBAL kept count, one scalar
        all-reduce before the backward pass), the gradient bucket is sum-reduced -- the update equals the single-process
        step on the concatenated batch.  No host synchronisation inside the step."""
        cfg = self.config
        self.model.train()
        if self.bucket is not None:
            self.bucket.zero()
            if self._gather_grads:
                self.bucket.release()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        W = world_size() if self.data_parallel else 1
        in_joints, in_masks, out_joints, out_masks, pm = self.process_coords(joints, masks, padding_mask, cfg, modality_selection, training=True)
        pose, vel = self.primary_state(joints, in_joints)
        # MASK_PADDED_PERSONS (extension, off = the reference): hand the model collate_batch's BOOL mask instead of the float copy
        # batch_process_coords returns -- padded persons are then masked (-inf) instead of biased (+1), and the model skips them
        # in the local former (model_jta.py `_transform`); a different model function from the reference's training loop
        pm_model = padding_mask.bool() if cfg.get("MASK_PADDED_PERSONS") else pm.to(cfg["DEVICE"])
        mse, pred = compute_loss(self.model, cfg, in_joints, out_joints, in_masks, out_masks, pm_model, mode='train',
                                 random_masking=random_masking)
        if cfg.get("VAL_LOSS_ONLY"):
            mse = mse * 0                     # train_jta.py:282-283: the value term alone drives the step
        loss = mse / W
        if self.valuenet is not None and (self.value_loss_with_multi_modal or not cfg.get("MULTI_MODAL", False)):
            vsum, cnt = emloco_loss_masked(cfg, self.valuenet, pred, pose, vel, in_joints.shape[1])
            if W > 1:
                cnt = all_reduce_(cnt.detach().clone())
            loss = loss + vsum / cnt.clamp(min=1.0)
        loss.backward()
        if self.bucket is not None and self._gather_grads:
            self.bucket.gather()
        if self.data_parallel:
            self.bucket.all_reduce(average=False)
        if self.flat_adam:
            self.optimizer.step(max_grad_norm=cfg["TRAIN"]["max_grad_norm"])
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), cfg["TRAIN"]["max_grad_norm"])
            self.optimizer.step()
        return loss.detach() * W, mse.detach()


# ---------------------------------------------------------------------------------------------- training loop pieces
def adjust_learning_rate(optimizer, epoch, config):
    """train_jta.py:129-141: lr * decay^epoch, times 0.1 after 80 % of the epochs when TRAIN.lr_drop (applied only then,
    as in the reference)."""
    lr = config['TRAIN']['lr'] * (config['TRAIN'].get('lr_decay', 1) ** epoch)
    if config['TRAIN'].get('lr_drop'):
        lr = lr * (0.1 ** (epoch // (config['TRAIN']['epochs'] * 4. / 5.)))
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr
    return lr


def save_checkpoint(model, optimizer, epoch, config, filename, logger=None):
    """train_jta.py:166-175 layout: {'model', 'optimizer', 'epoch', 'config'}; keys carry the DataParallel 'module.' prefix
    the reference's checkpoints have, so either side loads the other's files."""
    import os
    sd = model.state_dict()
    if not any(k.startswith("module.") for k in sd):
        sd = {"module." + k: v for k, v in sd.items()}
    path = os.path.join(config['OUTPUT']['ckpt_dir'], filename)
    if rank() == 0:               # data parallel: the replicas are identical, one writer (concurrent torch.save to one path corrupts it)
        if logger is not None:
            logger.info(f'Saving checkpoint to {path}.')
        torch.save({'model': sd, 'optimizer': optimizer.state_dict(), 'epoch': epoch, 'config': config}, path)
    barrier()                     # nobody resumes from / evaluates a file that is still being written
    return path


def load_checkpoint(model, path, optimizer=None, strict=False):
    ck = torch.load(path, map_location="cpu")
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ck["model"].items()}
    model.load_state_dict(sd, strict=strict)
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return ck.get("epoch", 0)


def evaluate_loss(model, dataloader, config, modality_selection='traj+all', limit_obs=False):
    """train_jta.py:78-96: mean validation loss (x100 ADE) over a loader."""
    model.eval()
    tot, n = 0.0, 0
    in_F = config['TRAIN']['input_track_size']
    with torch.no_grad():
        for joints, masks, padding_mask in dataloader:
            i, im, o, om, pm = batch_process_coords(joints, masks, padding_mask, config, modality_selection)
            pred = model(torch.nan_to_num(i), pm, False, limit_obs=limit_obs)
            loss = (MSE_LOSS_MULTI if config.get("MULTI_MODAL") else MSE_LOSS)(pred[:, in_F:], o, om)
            tot += loss.item() * len(joints)
            n += len(joints)
    return tot / max(n, 1)


def train_epoch(trainer, dataloader, epoch, modality_selection='traj+all', max_steps=None):
    """One epoch of train_jta.py:225-352 over a DataLoader with the EmLoco loss (EmLocoTrainer.step per batch)."""
    adjust_learning_rate(trainer.optimizer, epoch, trainer.config)
    sampler = getattr(dataloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):          # DistributedSampler: a new permutation (and per-rank shard) every epoch, as shuffle=True gives
        sampler.set_epoch(epoch)
    tot, n = 0.0, 0
    for step, (joints, masks, padding_mask) in enumerate(dataloader):
        loss, _mse = trainer.step(joints, masks, padding_mask, modality_selection)
        tot += float(loss) * len(joints)
        n += len(joints)
        if max_steps is not None and step + 1 >= max_steps:
            break
    return tot / max(n, 1)


# ---------------------------------------------------------------------------------------------- entry point (train_jta.py:354-506)
def load_config(path, exp_name="default", dataset_name="default", out_root="experiments"):
    """utils/utils.py:49-63: read the yaml, add the OUTPUT directories of the experiment and leave a copy of the config beside
    the checkpoints.  A relative `path` that does not exist is looked up among the configs that ship with this package."""
    import os
    import yaml
    if not os.path.exists(path):
        shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), path)
        if os.path.exists(shipped):
            path = shipped
    with open(path, "rt") as f:
        config = yaml.safe_load(f)
    base = os.path.join(out_root, dataset_name, exp_name or "default")
    config.setdefault("OUTPUT", {})
    for key, sub in (("log_dir", "logs"), ("ckpt_dir", "checkpoints"), ("runs_dir", "runs")):
        config["OUTPUT"][key] = os.path.join(base, sub)
        os.makedirs(config["OUTPUT"][key], exist_ok=True)
    if rank() == 0:               # one writer per experiment directory
        with open(os.path.join(config["OUTPUT"]["ckpt_dir"], "config.yaml"), "w") as f:
            yaml.safe_dump(config, f)
    return config


def create_logger(log_dir):
    import logging
    import os
    logger = logging.getLogger("emloco.train_jta")
    logger.setLevel(logging.INFO)
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
        for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(log_dir, "log.txt"))):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger


def prepare(config, logger, data_root="data"):
    """train_jta.py:180-223: the frozen LocoVal network (when the value loss is on) and the train / validation loaders."""
    from torch.utils.data import ConcatDataset, DataLoader
    from ..learning.value_pose_net import ValuePoseNet
    from .dataset_jta import collate_batch, create_dataset, get_datasets
    valuenet = None
    if config.get("USE_VALUELOSS"):
        valuenet = ValuePoseNet(use_pose=config.get("USE_POSE", True), use_vel=config.get("USE_VELOCITY", True)).to(config["DEVICE"])
        ck = config["MODEL"].get("valuenet_checkpoint", "")
        if ck:
            logger.info(f"Loading checkpoint from {ck}")
            valuenet.load_state_dict(torch.load(ck, map_location="cpu"))
        else:
            logger.info("No checkpoint provided for valuenet. Using random weights.")
    in_F, out_F = config["TRAIN"]["input_track_size"], config["TRAIN"]["output_track_size"]
    train = ConcatDataset(get_datasets(config["DATA"]["train_datasets"], config, logger, root=data_root))
    val = create_dataset(config["DATA"]["train_datasets"][0], logger, split="valid", track_size=in_F + out_F, track_cutoff=in_F,
                         preprocessed=config["DATA"]["preprocessed"], root=data_root)
    kw = dict(batch_size=config["TRAIN"]["batch_size"], num_workers=config["TRAIN"].get("num_workers", 0), collate_fn=collate_batch)
    if world_size() > 1:          # data parallel: every rank draws its own slice of each epoch's permutation (nn.DataParallel scatters a batch)
        from torch.utils.data.distributed import DistributedSampler
        return valuenet, DataLoader(train, sampler=DistributedSampler(train, shuffle=True, drop_last=True), **kw), DataLoader(val, shuffle=False, **kw)
    return valuenet, DataLoader(train, shuffle=True, **kw), DataLoader(val, shuffle=False, **kw)


def main(config, logger, valuenet, dataloader_train, dataloader_val, limit_obs=0):
    """train_jta.py:354-444: create the model (optionally resume), train epoch by epoch with the EmLoco loss, keep the best
    validation checkpoint (`best_val_checkpoint.pth.tar`, `best_val_checkpoint_<epoch>epoch.pth.tar`) and `checkpoint.pth.tar`
    every fifth epoch.  Returns (best validation ADE, its epoch)."""
    import os
    from .model_jta import create_model
    model = create_model(config, logger)
    if config.get("RESUME", -1) != -1:
        ck = config["MODEL"].get("checkpoint", "")
        if not ck:
            for name in ("checkpoint.pth.tar", "best_val_checkpoint.pth.tar"):
                if os.path.exists(os.path.join(config["OUTPUT"]["ckpt_dir"], name)):
                    ck = os.path.join(config["OUTPUT"]["ckpt_dir"], name)
                    break
            if not ck:
                raise ValueError("No checkpoint found.")
        logger.info(f"Loading checkpoint from {ck}")
        load_checkpoint(model, ck)
    else:
        logger.info("Training from scratch.")
    trainer = EmLocoTrainer(model, valuenet, config, data_parallel=world_size() > 1)
    logger.info(f"Model has {sum(p.numel() for p in model.parameters() if p.requires_grad)} parameters.")
    if valuenet is not None:
        logger.info(f'Using Value Loss weight: {float(config["TRAIN"]["valuenet_weight"]):.3f}')
    modality = config.get("MODALITY", "traj+all")
    min_val, best_epoch = 1e6, -1
    for epoch in range(config.get("RESUME", -1) + 1, config["TRAIN"]["epochs"]):
        tr = train_epoch(trainer, dataloader_train, epoch, modality, max_steps=1 if config.get("dry_run") else None)
        # data parallel: every rank evaluates its own shuffled pass and the ranks' numbers differ in the last digits; the decision
        # below leads into a collective (save_checkpoint's barrier), so it is taken on the ranks' MEAN, which is the same everywhere
        val_ade = all_reduce_mean_scalar(evaluate_loss(model, dataloader_val, config, modality, limit_obs=limit_obs) / 100)
        logger.info(f"Epoch {epoch} | Train Loss: {tr:.3f} | Val ADE: {val_ade:.3f}")
        if val_ade < min_val:
            min_val, best_epoch = val_ade, epoch
            logger.info(f"Best ADE: {val_ade}")
            save_checkpoint(model, trainer.optimizer, epoch, config, "best_val_checkpoint.pth.tar", logger)
            save_checkpoint(model, trainer.optimizer, epoch, config, f"best_val_checkpoint_{epoch}epoch.pth.tar", logger)
        if epoch % 5 == 0:
            save_checkpoint(model, trainer.optimizer, epoch, config, "checkpoint.pth.tar", logger)
        if config.get("dry_run"):
            break
    return min_val, best_epoch


def build_arg_parser():
    """The flags of the reference's `python train_jta.py` (train_jta.py:446-463), plus --data_root / --out_root for where the
    preprocessed splits and the experiment directories live."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--exp_name", type=str, default="", help="Experiment name. Otherwise will use timestamp")
    p.add_argument("--cfg", type=str, default="configs/jta_all_visual_cues.yaml", help="Config name. Otherwise will use default config")
    p.add_argument("--dry-run", action="store_true", help="Run just one iteration")
    p.add_argument("--valueloss_w", type=float, default=0, help="Use value loss")
    p.add_argument("--resume", type=int, default=-1, help="Resume training from a checkpoint")
    p.add_argument("--not_pose", action="store_true", help="Not using pose input for value function")
    p.add_argument("--not_vel", action="store_true", help="Not using velocity input for value function")
    p.add_argument("--limit_obs", type=int, default=0, help="Limit the number of observations")
    p.add_argument("--frame_mask", type=bool, default=True, help="Use frame masking")
    p.add_argument("--value_path", type=str, default="", help="Path to the value network checkpoint")
    p.add_argument("--value_dir", type=str, default="", help="Directory to the value network checkpoint")
    p.add_argument("--noisy_traj", type=float, default=0, help="Add noise to the trajectory to mimic real data")
    p.add_argument("--multi_modal", action="store_true", help="Use multimodal model")
    p.add_argument("--valueloss_only", action="store_true", help="Train with the value loss only")
    p.add_argument("--modality", type=str, default="traj+all", help="modality combination, e.g. 'traj', 'traj+3dpose', 'traj+all'")
    p.add_argument("--data_root", type=str, default="data", help="root of <dataset>/preprocess_smpl/<split>/part_*.pkl")
    p.add_argument("--out_root", type=str, default="experiments", help="root of the experiment directories")
    return p


def config_from_args(args):
    """train_jta.py:465-488: the command line folded into the config dict."""
    import os
    cfg = load_config(args.cfg, exp_name=args.exp_name, dataset_name="JTA", out_root=args.out_root)
    cfg["dry_run"], cfg["RESUME"] = args.dry_run, args.resume
    cfg["USE_VALUELOSS"] = args.valueloss_w > 0
    cfg["USE_POSE"], cfg["USE_VELOCITY"] = not args.not_pose, not args.not_vel
    cfg["TRAIN"]["valuenet_weight"] = args.valueloss_w
    cfg["USE_FRAME_MASK"], cfg["hypara_tune"] = args.frame_mask, False
    cfg["NOISY_TRAJ"], cfg["MULTI_MODAL"] = args.noisy_traj, args.multi_modal
    cfg["VAL_LOSS_ONLY"], cfg["MODALITY"] = args.valueloss_only, args.modality
    ck = args.value_path or cfg["MODEL"].get("valuenet_checkpoint", "")
    cfg["MODEL"]["valuenet_checkpoint"] = os.path.join(args.value_dir, ck) if ck else ""
    return cfg


if __name__ == "__main__":
    import random
    import numpy as np
    from .. import _lib, configure_runtime
    from ..dist import init_from_env
    configure_runtime()                                  # entry point: 16 hardware queues, ahead of the first GPU call
    _lib.require_device()                                # the predictor runs on the HIP library: no CPU path
    args = build_arg_parser().parse_args()
    _rank, local_rank, _world = init_from_env("nccl")
    torch.cuda.set_device(local_rank)
    cfg = config_from_args(args)
    cfg["DEVICE"] = f"cuda:{local_rank}"
    random.seed(cfg["SEED"]); torch.manual_seed(cfg["SEED"]); np.random.seed(cfg["SEED"])
    logger = create_logger(cfg["OUTPUT"]["log_dir"])
    logger.info("Initializing with config:")
    logger.info(cfg)
    valuenet, dl_train, dl_val = prepare(cfg, logger, data_root=args.data_root)
    best, epoch = main(cfg, logger, valuenet, dl_train, dl_val, limit_obs=args.limit_obs)
    logger.info(f"Best validation loss: {best:.3f} at epoch {epoch}")
    logger.info("All done.")
