import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
torch.manual_seed(0)
dev = "cuda:0"
e = lambda x, y: ((x.float() - y).abs().max() / y.abs().max()).item()
ops.set_matmul_precision("bf16")
for M in (1812, 927744 // 8):
    F, K = 1024, 128
    dz = torch.randn(M, F, device=dev)
    dz16 = dz.bfloat16()
    W1 = torch.randn(F, K, device=dev) * 0.1
    x = torch.randn(M, K, device=dev)
    dx = torch.empty(M, K, device=dev)
    ops.gemm(1, M, K, F, dz16, F, 0, 0, W1, K, 0, 1, dx, K, 0)
    ref = dz16.float() @ W1
    dW = torch.empty(F, K, device=dev)
    ops.gemm(1, F, K, M, dz16, F, 0, 1, x, K, 0, 1, dW, K, 0, ksplit=ops._ksplit_for(M, F * K))
    refW = dz16.float().T @ x
    h16 = torch.relu(dz).bfloat16()
    dz2 = torch.randn(M, K, device=dev)
    dW2 = torch.empty(K, F, device=dev)
    ops.gemm(1, K, F, M, dz2, K, 0, 1, h16, F, 0, 1, dW2, F, 0, ksplit=ops._ksplit_for(M, K * F))
    refW2 = dz2.T @ h16.float()
    y = torch.empty(M, K, device=dev)
    ops.gemm(1, M, K, F, h16, F, 0, 0, W1.T.contiguous(), F, 0, 0, y, K, 0)
    print(M, "dx", e(dx, ref), "dW1", e(dW, refW), "dW2", e(dW2, refW2), "fwd", e(y, h16.float() @ W1))
ops.set_matmul_precision("fp32")
