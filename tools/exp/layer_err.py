"""On the GPU box: one full-row EncoderLayer (d = 128, ff = 1024, S = 453) forward / backward against torch's own
nn.TransformerEncoderLayer in float64 and float32 with the same weights: which tensor carries which error."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from emloco_amd.predictor.model_jta import EncoderLayer
torch.manual_seed(0)
dev = "cuda:0"
d, H, ff, S, Bn = 128, 4, 1024, 453, 4
ours = EncoderLayer(d, H, ff, 0.0).to(dev)
with torch.no_grad():
    for p in ours.parameters():
        p.normal_(0, 0.09) if p.dim() == 2 else p.normal_(0, 0.05)
    ours.norm1.weight.add_(1.0); ours.norm2.weight.add_(1.0)
ours.eval()
x0 = torch.randn(Bn, S, d, device=dev)
dy = torch.randn(Bn, S, d, device=dev)
pad = torch.zeros(Bn, S, device=dev)
res = {}
x = x0.clone().requires_grad_(True)
y = ours(x, pad)
y.backward(dy)
res["ours"] = (y.detach().double().cpu(), x.grad.double().cpu(), {k: p.grad.double().cpu() for k, p in ours.named_parameters()})
for name, dt, dv in (("t64", torch.float64, "cpu"), ("t32", torch.float32, "cpu")):
    ref = torch.nn.TransformerEncoderLayer(d, H, ff, dropout=0.0, activation="relu", batch_first=True).to(dt)
    ref.load_state_dict({k: v.detach().cpu().to(dt) for k, v in ours.state_dict().items()})
    ref.train()                      # no dropout; train mode keeps torch off its fused inference fast path
    x = x0.cpu().to(dt).clone().requires_grad_(True)
    y = ref(x)
    y.backward(dy.cpu().to(dt))
    res[name] = (y.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in ref.named_parameters()})
e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
print("y     ours %.2e  torch32 %.2e" % (e(res["ours"][0], res["t64"][0]), e(res["t32"][0], res["t64"][0])))
print("dx    ours %.2e  torch32 %.2e" % (e(res["ours"][1], res["t64"][1]), e(res["t32"][1], res["t64"][1])))
for k in res["t64"][2]:
    print("%-28s ours %.2e  torch32 %.2e" % (k, e(res["ours"][2][k], res["t64"][2][k]), e(res["t32"][2][k], res["t64"][2][k])))
