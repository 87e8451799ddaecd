"""On the GPU box: a stack of 6 full-row EncoderLayers (the local former at its shipped size) against torch's own
nn.TransformerEncoderLayer stack in float64 AND float32 with the same weights -- how far two fp32 evaluations of the same
network are from the float64 one after nine post-norm layers (ReLU masks and LayerNorm amplify rounding differences)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from emloco_amd.predictor.model_jta import Encoder
from fullwidth_weights import make_state_dict
torch.manual_seed(0)
dev = "cuda:0"
d, H, ff, S, Bn, L = 128, 4, 1024, 453, 4, 6
ours = Encoder(d, H, ff, 0.0, L).to(dev)
shapes = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
sd = make_state_dict({"local_former." + k: s for k, s in shapes.items()}, seed=1234)
ours.load_state_dict({k: torch.from_numpy(sd["local_former." + k]) for k in shapes})
ours.eval()
x0 = torch.randn(Bn, S, d, device=dev) * 0.7
dy = torch.zeros(Bn, S, d, device=dev); dy[:, :21] = torch.randn(Bn, 21, d, device=dev)      # only 21 rows are read downstream
pad = torch.zeros(Bn, S, device=dev)
res = {}
x = x0.clone().requires_grad_(True)
y = ours(x, pad)
y.backward(dy)
res["ours"] = (y.detach().double().cpu(), x.grad.double().cpu(), {k: p.grad.double().cpu() for k, p in ours.named_parameters()})
for name, dt in (("t64", torch.float64), ("t32", torch.float32)):
    ref = torch.nn.TransformerEncoder(torch.nn.TransformerEncoderLayer(d, H, ff, dropout=0.0, activation="relu", batch_first=True), L).to(dt)
    ref.load_state_dict({k: v.detach().cpu().to(dt) for k, v in ours.state_dict().items()})
    ref.train()
    x = x0.cpu().to(dt).clone().requires_grad_(True)
    y = ref(x)
    y.backward(dy.cpu().to(dt))
    res[name] = (y.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in ref.named_parameters()})
e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
print("vs float64:   y ours %.2e torch32 %.2e | dx ours %.2e torch32 %.2e | ours vs torch32: y %.2e dx %.2e" % (
    e(res["ours"][0], res["t64"][0]), e(res["t32"][0], res["t64"][0]), e(res["ours"][1], res["t64"][1]), e(res["t32"][1], res["t64"][1]),
    e(res["ours"][0], res["t32"][0]), e(res["ours"][1], res["t32"][1])))
for k in ("layers.5.linear1.weight", "layers.4.linear1.weight", "layers.3.self_attn.in_proj_bias", "layers.0.self_attn.in_proj_weight", "layers.0.norm1.weight"):
    print("%-36s ours %.2e  torch32 %.2e  (vs float64)   ours vs torch32 %.2e" % (k, e(res["ours"][2][k], res["t64"][2][k]), e(res["t32"][2][k], res["t64"][2][k]),
                                                                                   e(res["ours"][2][k], res["t32"][2][k])))
