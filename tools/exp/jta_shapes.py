"""Per-shape table of the GEMM launches of one JTA EmLoco train step (fp32): python tools/exp/jta_shapes.py"""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
import bench
from emloco_amd.predictor import ops
from emloco_amd.learning.value_pose_net import ValuePoseNet
from emloco_amd.predictor.model_jta import TransMotionJTA
from emloco_amd.predictor.train_jta import EmLocoTrainer
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = {"DEVICE": str(dev), "MULTI_MODAL": False, "USE_FRAME_MASK": False,
       "TRAIN": {"input_track_size": 9, "output_track_size": 12, "lr": 1e-4, "max_grad_norm": 1.0, "valuenet_weight": 1.0}}
model = TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20,
                       output_scale=1, obs_and_pred=21, num_tokens=49, device=str(dev)).to(dev)
tr = EmLocoTrainer(model, ValuePoseNet(True, True).to(dev), cfg)
joints, masks, pad = bench.synthetic_jta_batch(256)
for _ in range(2): tr.step(joints, masks, pad)
rec = []
orig = ops.gemm
def gemm(batch, m, n, k, A, lda, sa, ta, B, ldb, sb, tb, Cm, ldc, sc, alpha=1.0, bias=None, flags=0, ksplit=1, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); orig(batch, m, n, k, A, lda, sa, ta, B, ldb, sb, tb, Cm, ldc, sc, alpha, bias, flags, ksplit, **kw); e1.record()
    rec.append(((batch, m, n, k, ta, tb, flags, ksplit), e0, e1))
ops.gemm = gemm
tr.step(joints, masks, pad)
torch.cuda.synchronize()
tab = collections.OrderedDict()
for key, e0, e1 in rec:
    t = tab.setdefault(key, [0, 0.0]); t[0] += 1; t[1] += e0.elapsed_time(e1)
tot = 0.0
print(f"{'batch':>6} {'m':>7} {'n':>5} {'k':>7} ta tb flg ks  calls      ms  TFLOP/s")
for (b, m, n, k, ta, tb, fl, ks), (c, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    fl_ = 2.0 * b * m * n * k * c
    tot += ms
    print(f"{b:6d} {m:7d} {n:5d} {k:7d} {ta:2d} {tb:2d} {fl:3d} {ks:2d} {c:6d} {ms:7.2f} {fl_ / ms / 1e9:8.1f}")
print("total gemm ms", tot)
