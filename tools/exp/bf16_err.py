"""On the GPU box: error of the bf16 mode's pieces against the fp32 path (relative to the tensor's max)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
from emloco_amd.predictor.model_jta import EncoderLayer
torch.manual_seed(0)
dev = "cuda:0"
lay = EncoderLayer(128, 4, 1024, 0.0).to(dev)
with torch.no_grad():
    for p in lay.parameters():
        p.normal_(0, 0.09) if p.dim() == 2 else p.normal_(0, 0.05)
    lay.norm1.weight.add_(1.0); lay.norm2.weight.add_(1.0)
lay.eval()
x = torch.randn(4 * 453, 128, device=dev, requires_grad=True)
w = torch.randn(4 * 453, 128, device=dev)
w384 = torch.randn(4 * 453, 384, device=dev)
e = lambda a, b: (round(((a.float() - b.float()).abs().max() / b.float().abs().max()).item(), 5), 'l2', round(((a.float() - b.float()).norm() / b.float().norm()).item(), 5))
def run(fn, *gin):
    out = {}
    for mode in ("fp32", "bf16"):
        ops.set_matmul_precision(mode)
        y = fn()
        g = torch.autograd.grad((y.float() * (w384 if y.shape[-1] == 384 else w).reshape(y.shape)).sum(), gin)
        out[mode] = (y.detach(), [t.detach() for t in g])
    ops.set_matmul_precision("fp32")
    return e(out["bf16"][0], out["fp32"][0]), [e(a, b) for a, b in zip(out["bf16"][1], out["fp32"][1])]
print("feed_forward  y, dx, dW1, dW2:", run(lambda: ops.feed_forward(x, lay.linear1.weight, lay.linear1.bias, lay.linear2.weight, lay.linear2.bias), x, lay.linear1.weight, lay.linear2.weight))
sa = lay.self_attn
print("in-proj (bf16 out) y, dx, dW, db:", run(lambda: ops.linear(x, sa.in_proj_weight, sa.in_proj_bias, out_bf16=True), x, sa.in_proj_weight, sa.in_proj_bias))
x3 = x.view(4, 453, 128)
pad = torch.zeros(4, 453, device=dev)
w3 = w.view(4, 453, 128)
def att():
    qkv = ops.linear(x3, sa.in_proj_weight, sa.in_proj_bias, out_bf16=True)
    return ops.attention(qkv, pad, 4)
print("in-proj + attention y, dx:", run(att, x))
print("whole layer y, dx:", run(lambda: lay(x3, pad), x))
