"""ValuePoseNet: the Locomotion Value function ("LocoVal").

Mirror of pacer/pacer/learning/value_pose_net.py (class ValuePoseNet :10-159): same constructor flags, same
parameter names (`_network.fc{1,2,3}.{weight,bias}`, so reference checkpoints load), same
`forward / calc_embodied_motion_loss` signatures.  The arithmetic -- yaw normalisation, hidden-joint zeroing,
100->49->24->1 MLP, and the whole backward -- is one fused HIP kernel pair (emloco_locoval_fwd/bwd).
Only the full-input variant (use_pose and use_vel, README steps 2-3) is on the hot path.

Bug-compatibility: the reference rotates / zeroes the CALLER's init_pose tensor in place (:97,:141-144), so in the
multi-modal training loop the pose is rotated cumulatively once per mode (train_jta.py:294-296).  `inplace_pose=True`
(default) reproduces that side effect; pass False for the side-effect-free behaviour.
"""
import torch
import torch.nn as nn

from ..predictor.ops import LocoValFn


class ValuePoseNet(nn.Module):
    def __init__(self, use_pose, use_vel, hide_toe=True, hide_spine=True, normalize=True, vru=False, inplace_pose=True, **kwargs):
        super().__init__(**kwargs)
        if not (use_pose and use_vel and hide_toe and hide_spine and normalize and not vru):
            raise NotImplementedError("the fused LocoVal kernel implements the full-input network (use_pose, use_vel, normalize)")
        self.use_pose, self.use_vel, self.hide_toe, self.hide_spine, self.normalize, self.use_vru = True, True, True, True, True, False
        self.inplace_pose = inplace_pose
        self.traj_size, self.pose_size, self.vel_size = 13 * 2, 24 * 3, 2
        self._network = nn.Sequential()
        self._network.add_module('fc1', nn.Linear(100, 49))
        self._network.add_module('relu1', nn.ReLU())
        self._network.add_module('fc2', nn.Linear(49, 24))
        self._network.add_module('relu2', nn.ReLU())
        self._network.add_module('fc3', nn.Linear(24, 1))
        self._network.add_module('sigmoid', nn.Sigmoid())
        for m in self._network:
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
        self.criterion = nn.MSELoss()

    def forward(self, waypoint_traj, init_pose=None, init_vel=None):
        assert init_pose is not None and init_vel is not None, "init_pose and init_vel should be included"
        n = self._network
        pose_in = init_pose.clone() if self.inplace_pose else init_pose     # the kernel keeps the un-rotated pose for its backward
        value, x100 = LocoValFn.apply(waypoint_traj, pose_in, init_vel, n.fc1.weight, n.fc1.bias, n.fc2.weight, n.fc2.bias,
                                      n.fc3.weight, n.fc3.bias)
        if self.inplace_pose and not init_pose.requires_grad:
            with torch.no_grad():
                init_pose.copy_(x100[:, 26:98].view(-1, 24, 3))
        return value

    net_forward = forward

    def calc_embodied_motion_loss(self, pred_traj, init_pose=None, init_vel=None):
        pred_value = self.forward(pred_traj, init_pose, init_vel)
        loss = self.criterion(pred_value, torch.ones_like(pred_value))
        return pred_value, loss
