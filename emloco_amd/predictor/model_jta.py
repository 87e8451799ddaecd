"""TransMotionJTA: the Social-Transmotion trajectory predictor on the MI355X kernels.

Mirror of social-transmotion/model_jta.py (live code :130-336): same constructor, same
`forward(tgt, padding_mask, random_masking=False, limit_obs=0, frame_masking=False)`, same order of torch.rand
draws for the stochastic masks (:205-264), same parameter names so reference checkpoints load
(`local_former.layers.{i}.self_attn.in_proj_weight`, `linear1.weight`, `predict_head.{i}.weight`, ...).
All projections, attention products, softmax and layer norms run through emloco_amd.predictor.ops.

Key padding follows torch's semantics (bool mask = -inf, float mask = additive; a fully -inf row gives zeros as in
torch >= 2.5 "safe softmax").  The reference passes a float 0/1 mask, so padded persons are biased, not removed.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _SelfAttnParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names."""

    def __init__(self, d, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class EncoderLayer(nn.Module):
    """Post-norm nn.TransformerEncoderLayer(d, nhead, ff, dropout, relu) (model_jta.py:177-185) on batch-first data."""

    def __init__(self, d, nhead, dim_ff, dropout):
        super().__init__()
        self.self_attn = _SelfAttnParams(d, nhead)
        self.linear1 = nn.Linear(d, dim_ff)
        self.linear2 = nn.Linear(dim_ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.p = dropout

    def forward(self, x, key_pad, live_rows=None, x_res=None, fork_out=False):
        """live_rows: only the first `live_rows` tokens of every sequence are read downstream (the last layer of a former):
        they attend over all keys, and the out-projection / norms / feed-forward run on those rows only -> (Bn, live_rows, d).
        Every row of a post-norm layer depends on the other rows through K and V alone, so the kept rows are unchanged.
        x_res: the same values as `x` as a second autograd edge (the previous layer's forked LayerNorm output) for the residual branch;
        fork_out: return (y, y') likewise for the next layer.  Two edges instead of one used twice: the LayerNorm backward adds the
        two gradients on load instead of autograd in a pass of its own."""
        sa = self.self_attn
        xr = x if x_res is None else x_res
        # (reduced-precision mode: q|k|v goes to the fused attention kernels as bf16 in HBM)
        qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias, out_bf16=(sa.embed_dim // sa.num_heads == 32))
        p = self.p if self.training else 0.0                 # the three nn.Dropout of the layer live in the GEMM epilogues,
        att = ops.attention(qkv, key_pad, sa.num_heads, drop_p=p, n_query=live_rows)   # MultiheadAttention's dropout on the probabilities in the attention kernels
        if live_rows is not None and live_rows < x.shape[1]:
            xr = xr[:, :live_rows].contiguous()
        a = ops.linear(att, sa.out_proj.weight, sa.out_proj.bias, drop_p=p)
        x1, x1r = ops.layer_norm(a, xr, self.norm1.weight, self.norm1.bias, self.norm1.eps, fork=True)
        # feed-forward, residual, norm2 (one forward launch in the reduced-precision mode, else feed_forward + layer_norm)
        return ops.feed_forward_norm(x1, x1r, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                     self.norm2.weight, self.norm2.bias, self.norm2.eps, drop_p=p, fork=fork_out)


class Encoder(nn.Module):
    def __init__(self, d, nhead, dim_ff, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d, nhead, dim_ff, dropout) for _ in range(num_layers)])

    def forward(self, x, key_pad, live_rows=None):
        last = len(self.layers) - 1
        xr = None
        for i, layer in enumerate(self.layers):
            if i == last:
                x = layer(x, key_pad, live_rows, x_res=xr)
            else:
                x, xr = layer(x, key_pad, None, x_res=xr, fork_out=True)
        return x


class _Enc(nn.Module):
    """learned encodings (model_jta.py:61-128): nn.Embedding(max_norm=True) renormalises the looked-up rows in place"""

    def __init__(self, n, d, name="learned_encoding"):
        super().__init__()
        setattr(self, name, nn.Embedding(n, d, max_norm=True))


class TransMotionJTA(nn.Module):
    def __init__(self, tok_dim=21, nhid=256, nhead=4, dim_feedfwd=1024, nlayers_local=2, nlayers_global=4, nmode=5, dropout=0.1,
                 activation='relu', output_scale=1, obs_and_pred=21, num_tokens=47, device='cuda:0', multi_modal=False):
        super().__init__()
        assert activation == 'relu'
        self.seq_len, self.nhid, self.output_scale, self.token_num = tok_dim, nhid, output_scale, num_tokens
        self.joints_3dpose, self.joints_2dpose, self.obs_and_pred = 24, 22, 21
        self.device, self.multi_modal, self.nmode = device, multi_modal, nmode
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
        self.fc_in_3dbb = nn.Linear(4, nhid)
        self.bb3d_encoder = _Enc(9, nhid)
        self.fc_in_2dbb = nn.Linear(4, nhid)
        self.bb2d_encoder = _Enc(9, nhid)
        self.fc_in_3dpose = nn.Linear(3, nhid)
        self.pose3d_encoder = _Enc(216, nhid)
        self.fc_in_2dpose = nn.Linear(2, nhid)
        self.pose2d_encoder = _Enc(198, nhid)
        self.local_former = Encoder(nhid, nhead, dim_feedfwd, dropout, nlayers_local)
        self.global_former = Encoder(nhid, nhead, dim_feedfwd, dropout, nlayers_global)
        self.dropout_p = dropout

    @staticmethod
    def _key_bias(padding_mask):
        """torch's key_padding_mask semantics: bool -> -inf on True; float -> added to the scores as is.  The reference
        hands over a FLOAT 0/1 mask (dataset_jta.py:86), i.e. a +1 bias on padded persons' keys; kept for parity."""
        if padding_mask.dtype == torch.bool:
            return torch.zeros(padding_mask.shape, device=padding_mask.device).masked_fill(padding_mask, float("-inf"))
        return padding_mask.float()

    def _drop(self, x):
        return F.dropout(x, self.dropout_p, self.training)

    def forward(self, tgt, padding_mask, random_masking=False, limit_obs=0, frame_masking=False):
        dev = self.device
        B, in_F, NJ, K = tgt.shape
        Fr, J = self.obs_and_pred, self.token_num
        out_F = Fr - in_F
        N = NJ // J
        i_idx = np.append(np.arange(0, in_F), np.repeat([in_F - 1], out_F))
        tgt = tgt[:, i_idx].reshape(B, Fr, N, J, K).to(dev)
        mr_traj = 0.2 if random_masking else 0
        mr_joints = 0.2 if random_masking else 0
        mr_mod = 0.3 if random_masking else 0
        mr_frame = 0.2 if frame_masking else 0
        R = lambda *s: torch.rand(s).float().to(dev)            # CPU draws moved to the device, as the reference does
        tgt_traj = tgt[:, :, :, 0, :2]
        tgt_traj = tgt_traj * (R(B, Fr, N) > mr_traj).unsqueeze(3)
        frame_mask = (R(B, in_F) > mr_frame).unsqueeze(2).unsqueeze(3)
        tgt_traj = torch.cat([tgt_traj[:, :in_F] * frame_mask, tgt_traj[:, in_F:]], dim=1)
        sel = [(R(B, 1, N, 1) > mr_mod).unsqueeze(4) for _ in range(4)]
        tgt_vis = tgt[:, :, :, 1:]
        tgt_3dbb = tgt_vis[:, :, :, 0, :4] * sel[0][:, :, :, 0]
        tgt_2dbb = tgt_vis[:, :, :, 1, :4] * sel[1][:, :, :, 0]
        tgt_3dpose = tgt_vis[:, :, :, 2:26, :3] * sel[2]
        tgt_2dpose = tgt_vis[:, :, :, 26:, :2] * sel[3]
        tgt_3dpose = tgt_3dpose * (R(B, Fr, N, self.joints_3dpose) > mr_joints).unsqueeze(4)
        tgt_2dpose = tgt_2dpose * (R(B, Fr, N, self.joints_2dpose) > mr_joints).unsqueeze(4)
        if limit_obs != 0:
            lm = torch.ones((B, Fr, N), device=dev)
            lm[:, :(9 - limit_obs)] = 0
            tgt_traj, tgt_3dbb, tgt_2dbb = tgt_traj * lm.unsqueeze(3), tgt_3dbb * lm.unsqueeze(3), tgt_2dbb * lm.unsqueeze(3)
            tgt_3dpose, tgt_2dpose = tgt_3dpose * lm[..., None, None], tgt_2dpose * lm[..., None, None]

        # input embeddings + learned encodings (model_jta.py:281-297)
        t = self._embed_traj(tgt_traj, Fr, N)                                                        # (B,F,N,d)
        bb3 = self._embed(tgt_3dbb[:, :9], self.fc_in_3dbb, self.bb3d_encoder, 9)                    # (B,9,N,d)
        bb2 = self._embed(tgt_2dbb[:, :9], self.fc_in_2dbb, self.bb2d_encoder, 9)
        p3 = tgt_3dpose[:, :9].transpose(2, 3).reshape(B, -1, N, 3)
        p3 = self._embed(p3, self.fc_in_3dpose, self.pose3d_encoder, p3.shape[1])                    # (B,216,N,d)
        p2 = tgt_2dpose[:, :9].transpose(2, 3).reshape(B, -1, N, 2)
        p2 = self._embed(p2, self.fc_in_2dpose, self.pose2d_encoder, p2.shape[1])                    # (B,198,N,d)

        seq = torch.cat((t, bb3, bb2, p3, p2), dim=1)                                                # (B,S,N,d), S = 453
        return self._transform(seq, padding_mask, B, N, Fr)

    def _embed_traj(self, tgt_traj, Fr, N):
        """fc_in_traj + LearnedTrajandIDEncoding (model_jta.py:61-78): time code on even channels, person code on odd."""
        dev = self.device
        half = self.nhid // 2
        t = ops.linear(tgt_traj, self.fc_in_traj.weight, self.fc_in_traj.bias)
        time_enc = self.double_id_encoder.learned_encoding(torch.arange(Fr, device=dev))             # (F, d/2)
        pers_enc = self.double_id_encoder.person_encoding(torch.arange(N, device=dev))               # (N, d/2)
        enc = torch.zeros(Fr, N, self.nhid, device=dev)
        enc[:, :, 0:half * 2:2] = time_enc.unsqueeze(1)
        enc[:, :, 1:half * 2:2] = pers_enc.unsqueeze(0)
        return self._drop(t + enc.unsqueeze(0))

    def _embed(self, x, fc, encmod, n):
        y = ops.linear(x, fc.weight, fc.bias)
        return self._drop(y + encmod.learned_encoding(torch.arange(n, device=self.device)).unsqueeze(1).unsqueeze(0))

    def _transform(self, seq, padding_mask, B, N, Fr):
        """local former over every person's S tokens -> global former over the N*21 trajectory tokens -> heads
        (model_jta.py:299-335; shared with TransMotionJRDB, model_jrdb.py:110-143).

        Padded persons.  What the mask means is torch's: the reference's training and evaluation loops hand over the FLOAT copy
        (dataset_jta.py:84 `padding_mask.float()`), which nn.MultiheadAttention ADDS to the scores -- padded persons' keys get
        a +1 bias, they are not removed, and their local-former outputs do reach the primary agent's rows through the global
        former; every person-sequence is computed then, as in the reference.  A BOOL mask (what collate_batch produces,
        dataset_jta.py:23) masks with -inf: a padded person's 21 tokens are never attended to in any global layer and only the
        primary agent's rows are read (:321), so nothing computed for a padded person reaches the output or any gradient.  With
        a bool mask the local former therefore runs on the live person-sequences only (`skip_padded_persons`, default on:
        gather the live rows, six layers, scatter the 21 trajectory tokens back, zeros for padded persons) -- outputs and
        gradients are those of the full computation (tests/golden/predictor_boolmask_jta.npz: the reference run with a bool mask)."""
        dev = self.device
        S = seq.shape[1]
        prune = getattr(self, "prune_dead_rows", True)
        live = None
        if padding_mask.dtype == torch.bool and prune and getattr(self, "skip_padded_persons", True):
            # the list of live persons is taken on the host: collate_batch builds the mask there, so a caller that passes it as it
            # is pays no device read-back; a device mask costs one synchronisation
            live = (~padding_mask.cpu().reshape(-1)).nonzero().squeeze(1)
            live = None if live.numel() == B * N else live.to(dev)
        pad = self._key_bias(padding_mask.to(dev))                                                   # (B, N) additive
        if live is not None:
            x = seq.permute(0, 2, 1, 3)[live // N, live % N].contiguous()                              # (L, S, d): live person-sequences
            out_live = self.local_former(x, None, 21) * self.output_scale + x[:, :21]                # no key bias: every key is live
            out_local = torch.zeros(B * N, 21, self.nhid, device=dev, dtype=out_live.dtype).index_copy(0, live, out_live)
        else:
            x = seq.permute(0, 2, 1, 3).reshape(B * N, S, self.nhid).contiguous()                    # batch-first (B*N, S, d)
            pad_local = pad.reshape(-1, 1).expand(-1, S).contiguous()
            # only the 21 trajectory tokens of every person leave the local former (model_jta.py:316) and only the primary agent's
            # rows leave the global one (:321): the last layer of each computes those rows alone (prune_dead_rows = False: all rows)
            out_local = self.local_former(x, pad_local, 21 if prune else None) * self.output_scale + (x[:, :21] if prune else x)
        # global former over the N*21 trajectory tokens of each scene: (B, N*21, d), person-major like the reference
        g = out_local[:, :21].reshape(B, N * 21, self.nhid).contiguous()
        pad_global = pad.repeat_interleave(Fr, dim=1).contiguous()                                   # (B, N*21)
        if prune:
            out_primary = self.global_former(g, pad_global, Fr) * self.output_scale + g[:, :Fr]      # (B,F,d) primary agent
        else:
            out_global = self.global_former(g, pad_global) * self.output_scale + g
            out_primary = out_global.view(B, N, Fr, self.nhid)[:, 0]
        if self.multi_modal:
            # the M prediction heads (model_jta.py:323-335: M separate nn.Linear(d, 2)) as ONE d -> 2M GEMM on the stacked weights
            W = torch.cat([h.weight for h in self.predict_head], dim=0)                              # (2M, d)
            bvec = torch.cat([h.bias for h in self.predict_head], dim=0)
            return ops.linear(out_primary, W, bvec).reshape(B, Fr, len(self.predict_head), 2)        # (B,F,M,2)
        return ops.linear(out_primary, self.fc_out_traj.weight, self.fc_out_traj.bias).reshape(B, Fr, 1, 2)


def create_model(config, logger=None):
    """model_jta.py:548-577."""
    m = config["MODEL"]
    return TransMotionJTA(tok_dim=m["seq_len"], nhid=m["dim_hidden"], nhead=m["num_heads"], dim_feedfwd=m["dim_feedforward"],
                          nlayers_local=m["num_layers_local"], nlayers_global=m["num_layers_global"], nmode=m.get("num_modes", 20),
                          output_scale=m["output_scale"], obs_and_pred=config["TRAIN"]["input_track_size"] + config["TRAIN"]["output_track_size"],
                          num_tokens=m["token_num"], device=config["DEVICE"], multi_modal=config.get("MULTI_MODAL", False)
                          ).to(config["DEVICE"]).float()
