"""Split-K sweep on the frozen policy's five GEMM shapes (M = 4096 envs): python tools/exp/policy_ksplit.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
dev = torch.device("cuda:0")
M = 4096
for label, n, k in (("task 1056->512", 512, 1056), ("task 512->256", 256, 512), ("actor 624->2048", 2048, 624), ("actor 2048->1024", 1024, 2048), ("mu 1024->69", 69, 1024)):
    A, B, bias = torch.randn(M, k, device=dev), torch.randn(n, k, device=dev), torch.randn(n, device=dev)
    Cm = torch.empty(M, n, device=dev)
    line = f"{label:18s}"
    for ks in (1, 2, 3, 4, 6, 8, 16):
        if ks > k // 128: continue
        def run():
            ops.gemm(1, M, n, k, A, k, 0, 0, B, k, 0, 0, Cm, n, 0, bias=bias, flags=ops.GEMM_BIAS | ops.GEMM_RELU, ksplit=ks)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        line += f"  ks{ks}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us"
    print(line, flush=True)
