"""JTA data pipeline -- mirror of /root/reference/social-transmotion/dataset_jta.py:1-26,88-190.

On-disk format consumed (dataset_jta.py:97-103): `<root>/<name>/preprocess_smpl/<split>/*.pkl`, each a pickled list of
scenes; a scene is a list of people; a person is a tuple (joints (21, 49, 4) float tensor, mask (21, 49) tensor).
`collate_batch` pads scenes to the largest person count (padding_mask True = padded person).  The raw-data branch
(`preprocessed: false`, utils/data.py loaders over the original JTA dumps) is out of scope; `write_synthetic_split`
produces files of the same format for tests and benches.  `batch_process_coords` lives in train_jta.py.
"""
import os
import pickle

import torch
from torch.nn.utils.rnn import pad_sequence


def collate_batch(batch):
    """dataset_jta.py:11-25."""
    joints_list, masks_list, num_people_list = [], [], []
    for joints, masks in batch:
        joints_list.append(joints)
        masks_list.append(masks)
        num_people_list.append(torch.zeros(joints.shape[0]))
    joints = pad_sequence(joints_list, batch_first=True)
    masks = pad_sequence(masks_list, batch_first=True)
    padding_mask = pad_sequence(num_people_list, batch_first=True, padding_value=1).bool()
    return joints, masks, padding_mask


class MultiPersonTrajPoseDataset(torch.utils.data.Dataset):
    def __init__(self, name, split="train", track_size=21, track_cutoff=9, segmented=True, add_flips=False, frequency=1,
                 preprocessed=False, root="data"):
        self.name, self.split, self.track_size, self.track_cutoff, self.frequency = name, split, track_size, track_cutoff, frequency
        if not preprocessed:
            raise NotImplementedError("only preprocessed splits are supported (DATA.preprocessed: true, as the shipped configs set)")
        self.datalist = []
        d = os.path.join(root, self.name, "preprocess_smpl", self.split)
        for file in sorted(os.listdir(d)):
            with open(os.path.join(d, file), "rb") as f:
                self.datalist += pickle.load(f)

    def __len__(self):
        return len(self.datalist)

    def __getitem__(self, idx):
        scene = self.datalist[idx]
        return torch.stack([torch.as_tensor(s[0]) for s in scene]), torch.stack([torch.as_tensor(s[1]) for s in scene])


class JtaAllVisualCuesDataset(MultiPersonTrajPoseDataset):
    def __init__(self, **args):
        super().__init__("jta_all_visual_cues", frequency=1, **args)


def create_dataset(dataset_name, logger=None, **args):
    if logger is not None:
        logger.info("Loading dataset " + dataset_name)
    if dataset_name == "jta_all_visual_cues":
        return JtaAllVisualCuesDataset(**args)
    if dataset_name == "jrdb_all_visual_cues":                     # dataset_jrdb.py: the same on-disk layout, 26 tokens per person
        return MultiPersonTrajPoseDataset("jrdb_all_visual_cues", frequency=1, **args)
    raise ValueError(f"Dataset with name '{dataset_name}' not found.")


def get_datasets(datasets_list, config, logger=None, root="data"):
    in_F, out_F = config["TRAIN"]["input_track_size"], config["TRAIN"]["output_track_size"]
    return [create_dataset(n, logger, split="train", track_size=in_F + out_F, track_cutoff=in_F,
                           preprocessed=config["DATA"]["preprocessed"], root=root) for n in datasets_list]


def write_synthetic_split(root, split, n_scenes, max_people=8, seed=0, name="jta_all_visual_cues", part_size=5000, tokens=49):
    """Scenes of the on-disk format with the statistics of SURVEY.md section 8d (2.5 fps walkers, 24 joints around the pelvis)."""
    g = torch.Generator().manual_seed(seed)
    d = os.path.join(root, name, "preprocess_smpl", split)
    os.makedirs(d, exist_ok=True)
    scenes = []
    for _ in range(n_scenes):
        n = int(torch.randint(1, max_people + 1, (1,), generator=g))
        people = []
        for _p in range(n):
            j = torch.randn(21, tokens, 4, generator=g)
            heading = (torch.rand(1, generator=g) * 2 - 1) * 3.14159 + torch.cumsum(torch.randn(21, generator=g) * 0.05, 0)
            step = torch.stack([torch.cos(heading), torch.sin(heading)], -1) * (torch.rand(1, generator=g) * 1.5 + 0.3) * 0.4
            j[:, 0, :2] = torch.cumsum(step, 0) + torch.randn(1, 2, generator=g) * 3
            j[:, 0, 2:] = 0
            if tokens >= 27:
                j[:, 3:27, :3] = j[:, 0:1, :3] + torch.randn(21, 24, 3, generator=g) * 0.3
            people.append((j, torch.ones(21, tokens)))
        scenes.append(people)
    for part, i in enumerate(range(0, n_scenes, part_size)):
        with open(os.path.join(d, f"part_{part}.pkl"), "wb") as f:
            pickle.dump(scenes[i:i + part_size], f)
    return d
