"""`train_jrdb.py`: TransMotionJRDB training with the EmLoco loss on JRDB-format batches.

Mirror of social-transmotion/train_jrdb.py (evaluate_loss :26-99, loaders :101-113, prepare :115-144, train :146-295,
main :297-350, command line :353-421).  What differs from the JTA loop and is kept: the batch comes from
dataset_jrdb.batch_process_coords (x flip, box rescale, the random-yaw augmentation); LocoVal's pose is the NORMALISED
tokens 2:26 of the last observed frame, no z flip (:186); with --multi_modal the value term is computed for the log but
never added to the loss (:206-231: the addition sits in the single-mode branch); validation runs on the "val" split with
three times the batch size, shuffled; periodic checkpoints are `checkpoint_<epoch>epoch.pth.tar`; the initial validation
loss is evaluated unless --resume 0 (`1e6 if not cfg["RESUME"]`, :333).  The step itself is EmLocoTrainer.step
(train_jta.py): model, LocoVal and optimiser arithmetic on the HIP kernels, one flat gradient all-reduce when data parallel.
"""
import os

import torch

from ..dist import all_reduce_mean_scalar, world_size
from .dataset_jrdb import batch_process_coords, collate_batch, create_dataset, get_datasets
from .train_jta import (MSE_LOSS, MSE_LOSS_MULTI, EmLocoTrainer, adjust_learning_rate, create_logger, load_checkpoint,  # noqa: F401
                        load_config, save_checkpoint)


class JrdbTrainer(EmLocoTrainer):
    process_coords = staticmethod(batch_process_coords)
    value_loss_with_multi_modal = False       # train_jrdb.py:206-231

    def primary_state(self, joints, in_joints):
        pose = in_joints[:, 8, 2:26, :3].clone().to(self.config["DEVICE"])                   # train_jrdb.py:186
        vel = ((in_joints[:, 8, 0, :2] - in_joints[:, 7, 0, :2]) * 2.5).clone()
        return pose, vel


def evaluate_loss(model, dataloader, valuenet, config, limit_obs=False, modality_selection='traj+all'):
    """train_jrdb.py:26-99: mean validation MSE loss (x100 ADE) over a loader (the value term there only feeds the progress bar)."""
    model.eval()
    tot, n = 0.0, 0
    in_F = config['TRAIN']['input_track_size']
    with torch.no_grad():
        for joints, masks, padding_mask, _idxs in dataloader:
            i, im, o, om, pm = batch_process_coords(joints, masks, padding_mask, config, modality_selection)
            pred = model(torch.nan_to_num(i), pm.to(config["DEVICE"]), False, limit_obs=limit_obs)
            loss = (MSE_LOSS_MULTI if config.get("MULTI_MODAL") else MSE_LOSS)(pred[:, in_F:], o, om)
            tot += loss.item() * len(joints)
            n += len(joints)
    return tot / max(n, 1)


def train_epoch(trainer, dataloader, epoch, modality_selection='traj+all', max_steps=None):
    """train_jrdb.py:146-262 over a DataLoader of (joints, masks, padding_mask, idxs)."""
    if trainer.config["TRAIN"].get("optimizer", "adam") == "adam":
        adjust_learning_rate(trainer.optimizer, epoch, trainer.config)
    sampler = getattr(dataloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):
        sampler.set_epoch(epoch)
    tot, n = 0.0, 0
    for step, (joints, masks, padding_mask, _idxs) in enumerate(dataloader):
        loss, _mse = trainer.step(joints, masks, padding_mask, modality_selection)
        tot += float(loss) * len(joints)
        n += len(joints)
        if max_steps is not None and step + 1 >= max_steps:
            break
    return tot / max(n, 1)


def prepare(config, logger, data_root="data"):
    """train_jrdb.py:115-144."""
    from torch.utils.data import ConcatDataset, DataLoader
    from ..learning.value_pose_net import ValuePoseNet
    valuenet = None
    if config.get("USE_VALUELOSS"):
        valuenet = ValuePoseNet(use_pose=config.get("USE_POSE", True), use_vel=config.get("USE_VELOCITY", True)).to(config["DEVICE"])
        ck = config["MODEL"].get("valuenet_checkpoint", "")
        if ck:
            logger.info(f"Loading checkpoint from {ck}")
            valuenet.load_state_dict(torch.load(ck, map_location="cpu"))
        else:                                  # (the reference asserts a checkpoint; none ships here)
            logger.info("No checkpoint provided for valuenet. Using random weights.")
    in_F, out_F = config["TRAIN"]["input_track_size"], config["TRAIN"]["output_track_size"]
    train = ConcatDataset(get_datasets(config["DATA"]["train_datasets"], config, logger, root=data_root))
    logger.info(f"Training on a total of {len(train)} annotations.")
    val = create_dataset(config["DATA"]["train_datasets"][0], logger, split="val", track_size=in_F + out_F, track_cutoff=in_F,
                         preprocessed=config["DATA"]["preprocessed"], root=data_root)
    kw = dict(num_workers=config["TRAIN"].get("num_workers", 0), collate_fn=collate_batch)
    bs = config["TRAIN"]["batch_size"]
    dl_val = DataLoader(val, batch_size=bs * 3, shuffle=True, **kw)
    if world_size() > 1:
        from torch.utils.data.distributed import DistributedSampler
        return valuenet, DataLoader(train, batch_size=bs, sampler=DistributedSampler(train, shuffle=True, drop_last=True), **kw), dl_val
    return valuenet, DataLoader(train, batch_size=bs, shuffle=True, **kw), dl_val


def main(config, logger, valuenet, dataloader_train, dataloader_val, limit_obs=0):
    """train_jrdb.py:297-350.  Returns (best validation ADE, its epoch)."""
    from .model_jrdb import create_model
    model = create_model(config, logger)
    if config.get("RESUME", -1) != -1:
        ck = config["MODEL"].get("checkpoint", "")
        if not ck:
            logger.info("Using the latest checkpoint.")
            for name in (f"checkpoint_{config['RESUME']}epoch.pth.tar", "best_val_checkpoint.pth.tar"):
                if os.path.exists(os.path.join(config["OUTPUT"]["ckpt_dir"], name)):
                    ck = os.path.join(config["OUTPUT"]["ckpt_dir"], name)
                    break
            if not ck:
                logger.info("No checkpoint found.")
                raise ValueError("No checkpoint found.")
        logger.info(f"Loading checkpoint from {ck}")
        load_checkpoint(model, ck)
    else:
        logger.info("Training from scratch.")
    trainer = JrdbTrainer(model, valuenet, config, data_parallel=world_size() > 1)
    logger.info(f"Model has {sum(p.numel() for p in model.parameters() if p.requires_grad)} parameters.")
    modality = config.get("MODALITY", "traj+all")
    min_val = 1e6 if not config.get("RESUME", -1) else all_reduce_mean_scalar(evaluate_loss(model, dataloader_val, valuenet, config) / 100)
    logger.info(f"Initial validation loss: {min_val:.3f}")
    if valuenet is not None:
        logger.info(f'Using Value Loss weight: {float(config["TRAIN"]["valuenet_weight"]):.3f}')
    best_epoch = -1
    for epoch in range(config.get("RESUME", -1) + 1, config["TRAIN"]["epochs"]):
        tr = train_epoch(trainer, dataloader_train, epoch, modality, max_steps=1 if config.get("dry_run") else None)
        # (the ranks' mean: the comparison below leads into save_checkpoint's barrier and must come out the same on every rank)
        val_ade = all_reduce_mean_scalar(evaluate_loss(model, dataloader_val, valuenet, config, limit_obs=False, modality_selection=modality) / 100)
        logger.info(f"Epoch {epoch} | Train Loss: {tr:.3f} | Val ADE: {val_ade:.3f}")
        if val_ade < min_val:
            min_val, best_epoch = val_ade, epoch
            logger.info(f"Best ADE: {val_ade}")
            save_checkpoint(model, trainer.optimizer, epoch, config, "best_val_checkpoint.pth.tar", logger)
            save_checkpoint(model, trainer.optimizer, epoch, config, f"best_val_checkpoint_{epoch}epoch.pth.tar", logger)
        if epoch % 5 == 0:
            save_checkpoint(model, trainer.optimizer, epoch, config, f"checkpoint_{epoch}epoch.pth.tar", logger)
        if config.get("dry_run"):
            break
    return min_val, best_epoch


def build_arg_parser():
    """The flags of the reference's `python train_jrdb.py` (train_jrdb.py:353-370; --use_hypara_best needs its optuna study
    and is rejected), plus --data_root / --out_root."""
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("--exp_name", type=str, default="", help="Experiment name. Otherwise will use timestamp")
    p.add_argument("--cfg", type=str, default="configs/jrdb_all_visual_cues.yaml", help="Config name. Otherwise will use default config")
    p.add_argument("--dry-run", action="store_true", help="Run just one iteration")
    p.add_argument("--valueloss_w", type=float, default=0, help="Weight for value loss")
    p.add_argument("--resume", type=int, default=-1, help="Resume training from a checkpoint")
    p.add_argument("--not_pose", action="store_true", help="Not using pose input for value function")
    p.add_argument("--not_vel", action="store_true", help="Not using velocity input for value function")
    p.add_argument("--limit_obs", type=int, default=0, help="Limit the number of past observations")
    p.add_argument("--frame_mask", type=bool, default=True, help="Use frame masking")
    p.add_argument("--value_path", type=str, default="", help="Path to the value network checkpoint")
    p.add_argument("--value_dir", type=str, default="", help="Directory to the value network checkpoint")
    p.add_argument("--noisy_traj", action="store_true", help="Add noise to the trajectory to mimic real data")
    p.add_argument("--use_hypara_best", action="store_true", help="Use the best hyperparameters (optuna study: not supported)")
    p.add_argument("--multi_modal", action="store_true", help="Use multimodal model")
    p.add_argument("--valueloss_only", action="store_true", help="Train with the value loss only")
    p.add_argument("--modality", type=str, default="traj+all", help="available modality combination from['traj','traj+all', 'traj+2dbox','traj+3dpose']")
    p.add_argument("--data_root", type=str, default="data", help="root of <dataset>/preprocess_smpl_filtered_v4/<split>/part_*.pkl")
    p.add_argument("--out_root", type=str, default="experiments", help="root of the experiment directories")
    return p


def config_from_args(args):
    """train_jrdb.py:372-397."""
    if args.use_hypara_best:
        raise NotImplementedError("--use_hypara_best reads an optuna study (hyper_tuning_jrdb.py): pass --valueloss_w instead")
    cfg = load_config(args.cfg, exp_name=args.exp_name, dataset_name="JRDB", out_root=args.out_root)
    cfg["dry_run"], cfg["RESUME"] = args.dry_run, args.resume
    cfg["USE_VALUELOSS"] = args.valueloss_w > 0
    cfg["USE_POSE"], cfg["USE_VELOCITY"] = not args.not_pose, not args.not_vel
    cfg["USE_FRAME_MASK"], cfg["NOISY_TRAJ"] = args.frame_mask, args.noisy_traj
    cfg["MULTI_MODAL"], cfg["VAL_LOSS_ONLY"], cfg["MODALITY"] = args.multi_modal, args.valueloss_only, args.modality
    cfg["TRAIN"]["valuenet_weight"] = args.valueloss_w
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
