"""On the GPU box: error of the fused attention forward / backward against float64 torch at the production shape."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from emloco_amd.predictor import ops
torch.manual_seed(4)
dev = "cuda:0"
for scale in (0.6, 2.0):
    Bn, S, H, d = 4, 453, 4, 128
    qkv = (torch.randn(Bn, S, 3 * d, device=dev) * scale).requires_grad_(True)
    pad = torch.zeros(Bn, S, device=dev)
    dout = torch.randn(Bn, S, d, device=dev)
    out = ops.FusedAttentionFn.apply(qkv, pad, H)
    out.backward(dout)
    q64 = qkv.detach().double().requires_grad_(True)
    qh, kh, vh = (q64[..., i * d:(i + 1) * d].view(Bn, S, H, 32).transpose(1, 2) for i in range(3))
    s = qh @ kh.transpose(-1, -2) / np.sqrt(32.0)
    ref = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(Bn, S, d)
    ref.backward(dout.double())
    q32 = qkv.detach().clone().requires_grad_(True)
    qh, kh, vh = (q32[..., i * d:(i + 1) * d].view(Bn, S, H, 32).transpose(1, 2) for i in range(3))
    s = qh @ kh.transpose(-1, -2) / np.sqrt(32.0)
    r32 = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(Bn, S, d)
    r32.backward(dout)
    e = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    g = qkv.grad
    print(f"scale {scale}: out fused {e(out.detach(), ref.detach()):.2e} torch32 {e(r32.detach(), ref.detach()):.2e} | "
          f"dq fused {e(g[..., :d], q64.grad[..., :d]):.2e} torch32 {e(q32.grad[..., :d], q64.grad[..., :d]):.2e} | "
          f"dk fused {e(g[..., d:2*d], q64.grad[..., d:2*d]):.2e} torch32 {e(q32.grad[..., d:2*d], q64.grad[..., d:2*d]):.2e} | "
          f"dv fused {e(g[..., 2*d:], q64.grad[..., 2*d:]):.2e} torch32 {e(q32.grad[..., 2*d:], q64.grad[..., 2*d:]):.2e}")
