"""JRDB batch preparation -- mirror of /root/reference/social-transmotion/dataset_jrdb.py:14-118.

`collate_batch` pads scenes to the largest person count (padding_mask True = padded person); `batch_process_coords`
centres trajectories on the primary person's position at the last observed frame, centres and rescales (x0.25) the 2-D
boxes, flips the pose x axis, applies the evaluation modality selection or the training-time augmentation
(`jrdb_2dbox`: pose dropped; `jrdb_all_visual_cues`: one random yaw per scene applied to trajectory and pose), and
reshapes (B,N,F,J,K) -> (B,F,N*J,K).  The input batch is not modified (the reference normalises in place on cpu and on a
device copy on cuda; evaluate_jrdb.py:88 reads the raw pose afterwards, which only works with the copy).
"""
import torch
from torch.nn.utils.rnn import pad_sequence


def collate_batch(batch):
    joints_list, masks_list, idxs_list, num_people_list = [], [], [], []
    for joints, masks, idxs in batch:
        joints_list.append(joints)
        masks_list.append(masks)
        idxs_list.append(idxs)
        num_people_list.append(torch.zeros(joints.shape[0]))
    joints = pad_sequence(joints_list, batch_first=True)
    masks = pad_sequence(masks_list, batch_first=True)
    padding_mask = pad_sequence(num_people_list, batch_first=True, padding_value=1).bool()
    return joints, masks, padding_mask, idxs_list


def random_yaw_rotate(x, angles):
    """getRandomRotatePoseTransform (dataset_jrdb.py:91-111): rotate (B,N,F,J,3) about z by one angle per scene."""
    B = x.shape[0]
    rot = torch.zeros(B, 3, 3, device=x.device)
    rot[:, 0, 0] = torch.cos(angles)
    rot[:, 0, 1] = -torch.sin(angles)
    rot[:, 1, 0] = torch.sin(angles)
    rot[:, 1, 1] = torch.cos(angles)
    rot[:, 2, 2] = 1
    return torch.bmm(x.reshape(B, -1, 3).float(), rot).reshape(x.shape)


def batch_process_coords(coords, masks, padding_mask, config, modality_selection='traj+all', training=False, multiperson=True):
    joints = coords.to(config["DEVICE"]).clone()
    masks = masks.to(config["DEVICE"])
    in_F = config["TRAIN"]["input_track_size"]
    joints[:, :, :, 0] = joints[:, :, :, 0] - joints[:, 0:1, (in_F - 1):in_F, 0]
    joints[:, :, :, 1] = joints[:, :, :, 1] - joints[:, :, (in_F - 1):in_F, 1]
    joints[:, :, :, 1] *= 0.25
    joints[:, :, :, 2:, 0] *= -1
    B, N, F, J, K = joints.shape
    if not training:
        if modality_selection == 'traj':
            joints[:, :, :, 1:] = 0
        elif modality_selection == 'traj+2dbox':
            joints[:, :, :, 2:] = 0
        elif modality_selection == 'traj+3dpose':
            joints[:, :, :, 1] = 0
        elif modality_selection != 'traj+all':
            raise ValueError('modality error')
    elif 'jrdb_2dbox' in config['DATA']['train_datasets']:
        joints[:, :, :, 2:] = 0
    elif 'jrdb_all_visual_cues' in config['DATA']['train_datasets']:
        angles = torch.deg2rad(torch.rand(len(joints)) * 360).to(joints.device)
        joints[:, :, :, 0, :3] = random_yaw_rotate(joints[:, :, :, 0, :3].unsqueeze(3), angles).squeeze(3)
        joints[:, :, :, 2:, :3] = random_yaw_rotate(joints[:, :, :, 2:, :3], angles)
    joints = joints.transpose(1, 2).reshape(B, F, N * J, K)
    masks = masks.transpose(1, 2).reshape(B, F, N * J)
    out_F = config["TRAIN"]["output_track_size"]
    return (joints[:, :in_F].float(), masks[:, :in_F].float(), joints[:, in_F:in_F + out_F].float(),
            masks[:, in_F:in_F + out_F].float(), padding_mask.float())
