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


# ---------------------------------------------------------------------------------------------- datasets (dataset_jrdb.py:129-260)
PREPROCESSED_DIR = "preprocess_smpl_filtered_v4"       # dataset_jrdb.py:146: the split directory the reference reads


class MultiPersonTrajPoseDataset(torch.utils.data.Dataset):
    """`<root>/<name>/preprocess_smpl_filtered_v4/<split>/*.pkl`, each a pickled list of scenes; a scene is a list of people; a
    person is a tuple (joints (21, 26, 4), mask (21, 26)[, ids]).  Items carry the scene index (dataset_jrdb.py:203-210: the
    evaluation looks up scene names through it).  The raw-data branch (`preprocessed: false`) is out of scope."""

    def __init__(self, name, split="train", track_size=21, track_cutoff=9, segmented=True, add_flips=False, frequency=1,
                 preprocessed=False, root="data"):
        import os
        import pickle
        self.name, self.split, self.track_size, self.track_cutoff, self.frequency = name, split, track_size, track_cutoff, frequency
        if not preprocessed:
            raise NotImplementedError("only preprocessed splits are supported (DATA.preprocessed: true, as the shipped configs set)")
        self.datalist = []
        d = os.path.join(root, self.name, PREPROCESSED_DIR, self.split)
        for file in sorted(os.listdir(d)):
            with open(os.path.join(d, file), "rb") as f:
                self.datalist += pickle.load(f)

    def __len__(self):
        return len(self.datalist)

    def show_meta_info(self, idx):
        return [s[2] for s in self.datalist[idx]]

    def __getitem__(self, idx):
        scene = self.datalist[idx]
        return torch.stack([torch.as_tensor(s[0]) for s in scene]), torch.stack([torch.as_tensor(s[1]) for s in scene]), idx


def create_dataset(dataset_name, logger=None, **args):
    if logger is not None:
        logger.info("Loading dataset " + dataset_name)
    if dataset_name == "jta_all_visual_cues":
        raise NotImplementedError("This is a code for JRDB dataset, not JTA dataset.")
    if dataset_name in ("jrdb_2dbox", "jrdb_all_visual_cues"):
        return MultiPersonTrajPoseDataset(dataset_name, frequency=1, **args)
    raise ValueError(f"Dataset with name '{dataset_name}' not found.")


def get_datasets(datasets_list, config, logger=None, root="data"):
    in_F, out_F = config["TRAIN"]["input_track_size"], config["TRAIN"]["output_track_size"]
    return [create_dataset(n, logger, split="train", track_size=in_F + out_F, track_cutoff=in_F,
                           preprocessed=config["DATA"]["preprocessed"], root=root) for n in datasets_list]


def write_synthetic_split(root, split, n_scenes, max_people=8, seed=0, name="jrdb_all_visual_cues"):
    """Scenes in the on-disk format above with SURVEY 8d's statistics (26 tokens per person: trajectory, 2-D box, 24 joints)."""
    import os
    import pickle
    import tempfile
    from .dataset_jta import write_synthetic_split as write_jta
    with tempfile.TemporaryDirectory() as tmp:
        src = write_jta(tmp, split, n_scenes, max_people=max_people, seed=seed, name=name, tokens=26)
        d = os.path.join(root, name, PREPROCESSED_DIR, split)
        os.makedirs(d, exist_ok=True)
        for f in sorted(os.listdir(src)):
            with open(os.path.join(src, f), "rb") as fi:
                scenes = pickle.load(fi)
            for si, people in enumerate(scenes):            # pose joints around the pelvis in tokens 2:26, ids as the reference stores them
                for pi, (j, m) in enumerate(people):
                    j[:, 2:26, :3] = j[:, 0:1, :3] + torch.randn(21, 24, 3, generator=torch.Generator().manual_seed(seed * 7919 + si * 31 + pi)) * 0.3
                    people[pi] = (j, m, (f"scene_{si}", list(range(21))))
            with open(os.path.join(d, f), "wb") as fo:
                pickle.dump(scenes, fo)
    return d
