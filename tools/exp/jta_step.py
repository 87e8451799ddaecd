"""The JTA EmLoco train step alone, for profiling: python tools/exp/jta_step.py [steps]   (JTA_PRECISION=bf16: the reduced-precision mode)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
import bench
from emloco_amd.learning.value_pose_net import ValuePoseNet
from emloco_amd.predictor.model_jta import TransMotionJTA
from emloco_amd.predictor.train_jta import EmLocoTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from emloco_amd.predictor import ops
ops.set_matmul_precision(os.environ.get("JTA_PRECISION", ops.DEFAULT_PRECISION))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = {"DEVICE": str(dev), "MULTI_MODAL": False, "USE_FRAME_MASK": False,
       "TRAIN": {"input_track_size": 9, "output_track_size": 12, "lr": 1e-4, "max_grad_norm": 1.0, "valuenet_weight": 1.0}}
model = TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20,
                       output_scale=1, obs_and_pred=21, num_tokens=49, device=str(dev)).to(dev)
tr = EmLocoTrainer(model, ValuePoseNet(True, True).to(dev), cfg)
joints, masks, pad = bench.synthetic_jta_batch(256)
for _ in range(2): tr.step(joints, masks, pad)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): tr.step(joints, masks, pad)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"{dt*1e3:.1f} ms / step, {256/dt:.1f} samples/s")
