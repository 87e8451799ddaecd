import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from fullwidth_weights import make_state_dict, sample
from emloco_amd.predictor.model_jta import TransMotionJTA
from emloco_amd.predictor.train_jta import MSE_LOSS_MULTI
g = np.load("/root/repo/tests/golden/predictor_fulldepth_jta_mm.npz")
dev = "cuda:0"
model = TransMotionJTA(tok_dim=453, nhid=128, nhead=4, dim_feedfwd=1024, nlayers_local=6, nlayers_global=3, nmode=20, output_scale=1,
                       obs_and_pred=21, num_tokens=49, device=dev, multi_modal=True).to(dev).float()
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = make_state_dict(shapes, seed=int(g["weight_seed"]))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.eval()
in_joints, pm, out_joints = (torch.from_numpy(g[k]).to(dev) for k in ("in_joints", "pm", "out_joints"))
pred = model(in_joints.clone(), pm.clone())
print("logits rel err", np.abs(pred.detach().cpu().numpy() - g["pred"]).max() / np.abs(g["pred"]).max())
loss = MSE_LOSS_MULTI(pred[:, 9:], out_joints)
norm = torch.norm(pred[:, 9:] - out_joints[:, :, 0, :2].unsqueeze(2), dim=-1).mean(1)
srt = norm.sort(dim=1).values
print("best / second-best mode loss per sample:", srt[:, :2].tolist())
loss.backward()
params = dict(model.named_parameters())
for k, v in g.items():
    if k.startswith("grad__"):
        a = params[k[6:].replace("__", ".")].grad.cpu().numpy()
    elif k.startswith("gsample__"):
        a = sample(params[k[9:].replace("__", ".")].grad.cpu().numpy())
    else:
        continue
    print("%-60s rel %.2e scale %.2e" % (k, np.abs(a - v).max() / max(np.abs(v).max(), 1e-30), np.abs(v).max()))
