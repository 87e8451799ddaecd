"""CPU restatement of the predictor path in plain PyTorch fp32 -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Restates social-transmotion/model_jta.py:130-336 (TransMotionJTA) with stock torch.nn modules, parameter names equal
to the reference's so its state_dict loads, pinned against tests/golden/predictor_{single,multi}.npz
(tests/test_oracle_golden.py).  Used as (1) an independent fp32 reference for the HIP kernels and (2) the
`cpu_baseline` of the JTA leg of bench.py on the GPU box, where /root/reference does not exist.
"""
import numpy as np
import torch
import torch.nn as nn


class TransMotionJTAOracle(nn.Module):
    def __init__(self, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20, num_tokens=49,
                 multi_modal=False, dropout=0.1):
        super().__init__()
        self.nhid, self.J, self.multi_modal = nhid, num_tokens, multi_modal
        self.fc_in_traj = nn.Linear(2, nhid)
        if multi_modal:
            self.predict_head = nn.ModuleList([nn.Linear(nhid, 2) for _ in range(nmode)])
        else:
            self.fc_out_traj = nn.Linear(nhid, 2)
        self.double_id_encoder = nn.Module()
        self.double_id_encoder.learned_encoding = nn.Embedding(21, nhid // 2, max_norm=True)
        self.double_id_encoder.person_encoding = nn.Embedding(1000, nhid // 2, max_norm=True)
        self.id_encoder = nn.Module()
        self.id_encoder.person_encoding = nn.Embedding(1000, nhid, max_norm=True)
        for name, k, n in (("3dbb", 4, 9), ("2dbb", 4, 9), ("3dpose", 3, 216), ("2dpose", 2, 198)):
            setattr(self, "fc_in_" + name, nn.Linear(k, nhid))
            enc = nn.Module()
            enc.learned_encoding = nn.Embedding(n, nhid, max_norm=True)
            setattr(self, {"3dbb": "bb3d_encoder", "2dbb": "bb2d_encoder", "3dpose": "pose3d_encoder", "2dpose": "pose2d_encoder"}[name], enc)
        mk = lambda n: nn.TransformerEncoder(nn.TransformerEncoderLayer(nhid, nhead, dim_feedfwd, dropout, "relu"), n, enable_nested_tensor=False)
        self.local_former, self.global_former = mk(nlayers_local), mk(nlayers_global)

    def forward(self, tgt, padding_mask):
        """Deterministic forward (no stochastic masks): tgt (B,9,N*J,4), padding_mask (B,N) float or bool."""
        B, in_F, NJ, K = tgt.shape
        F, J = 21, self.J
        N = NJ // J
        idx = np.append(np.arange(in_F), np.repeat([in_F - 1], F - in_F))
        tgt = tgt[:, idx].reshape(B, F, N, J, K)
        d, half = self.nhid, self.nhid // 2
        t = self.fc_in_traj(tgt[:, :, :, 0, :2])
        t[:, :, :, 0:half * 2:2] = t[:, :, :, 0:half * 2:2] + self.double_id_encoder.learned_encoding(torch.arange(F)).unsqueeze(1).unsqueeze(0)
        t[:, :, :, 1:half * 2:2] = t[:, :, :, 1:half * 2:2] + self.double_id_encoder.person_encoding(torch.arange(N)).unsqueeze(0).unsqueeze(0)
        vis = tgt[:, :, :, 1:]
        e = lambda x, fc, enc: fc(x) + enc.learned_encoding(torch.arange(x.shape[1])).unsqueeze(1).unsqueeze(0)
        bb3 = e(vis[:, :9, :, 0, :4], self.fc_in_3dbb, self.bb3d_encoder)
        bb2 = e(vis[:, :9, :, 1, :4], self.fc_in_2dbb, self.bb2d_encoder)
        p3 = e(vis[:, :9, :, 2:26, :3].transpose(2, 3).reshape(B, -1, N, 3), self.fc_in_3dpose, self.pose3d_encoder)
        p2 = e(vis[:, :9, :, 26:, :2].transpose(2, 3).reshape(B, -1, N, 2), self.fc_in_2dpose, self.pose2d_encoder)
        seq = torch.cat([x.transpose(0, 1).reshape(x.shape[1], -1, d) for x in (t, bb3, bb2, p3, p2)], 0)      # (S, B*N, d)
        S = seq.shape[0]
        pad_local = padding_mask.reshape(-1).unsqueeze(1).repeat_interleave(S, dim=1)
        out_local = self.local_former(seq, mask=None, src_key_padding_mask=pad_local) + seq
        g = out_local[:21].reshape(21, B, N, d).permute(2, 0, 1, 3).reshape(-1, B, d)
        out_global = self.global_former(g, mask=None, src_key_padding_mask=padding_mask.repeat_interleave(F, dim=1)) + g
        prim = out_global.reshape(N, F, B, d)[0]
        if self.multi_modal:
            return torch.stack([h(prim) for h in self.predict_head], dim=2).transpose(0, 1)
        return self.fc_out_traj(prim).transpose(0, 1).reshape(B, F, 1, 2)


class LocoValOracle(nn.Module):
    """value_pose_net.py:36-159 restated (full-input network), side-effect free."""

    def __init__(self):
        super().__init__()
        self._network = nn.Sequential()
        for name, mod in (("fc1", nn.Linear(100, 49)), ("relu1", nn.ReLU()), ("fc2", nn.Linear(49, 24)), ("relu2", nn.ReLU()),
                          ("fc3", nn.Linear(24, 1)), ("sigmoid", nn.Sigmoid())):
            self._network.add_module(name, mod)

    def forward(self, traj, pose, vel):
        x = traj[:, 1, 0]
        near = x.abs() < 1e-10
        x = x * (~near) + near * 1e-10
        a = torch.atan2(traj[:, 1, 1], x)
        R = torch.stack([torch.cos(a), -torch.sin(a), torch.sin(a), torch.cos(a)], -1).view(-1, 2, 2)
        tr = torch.bmm(traj[..., :2], R)
        po = torch.cat([torch.bmm(pose[..., :2], R), pose[..., 2:]], -1).clone()
        po[:, [4, 8, 9, 10, 11]] = 0
        ve = torch.bmm(vel.unsqueeze(1), R)[:, 0]
        return self._network(torch.cat([tr.reshape(-1, 26), po.reshape(-1, 72), ve], -1))


def emloco_train_step(model, vnet, opt, in_joints, out_joints, pm, pose, vel, multi=False):
    """One train_jta iteration on the CPU: forward, EmLoco loss, backward, clip, Adam."""
    opt.zero_grad()
    pred = model(in_joints, pm)
    gt = out_joints[:, :, 0, :2]
    if multi:
        norm = torch.norm(pred[:, 9:, :, :2] - gt.unsqueeze(2), p=2, dim=-1)
        mse = torch.mean(torch.min(torch.mean(norm, dim=1), dim=1)[0]) * 100
    else:
        mse = torch.mean(torch.mean(torch.norm(pred[:, 9:, 0, :2] - gt, p=2, dim=-1), dim=-1)) * 100
    traj = torch.cat([torch.zeros(pred.shape[0], 1, 2), pred[:, 9:, 0, :2]], dim=1)
    v = vnet(traj, pose, vel)
    loss = mse + torch.mean((v - 1.0) ** 2)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    return loss.detach()
