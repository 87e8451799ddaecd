"""Do the two gradient GEMMs of a linear layer (dx = dy W, dW = dy^T x: both read dy) gain from running on two streams?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emloco_amd  # noqa
import torch
from emloco_amd.predictor import ops

dev = torch.device("cuda", 0)
M = 466 * 1024
for (N, K, name) in ((1024, 128, "FFN1 (dy M x 1024)"), (128, 1024, "FFN2 (dy M x 128)"), (384, 128, "qkv (dy M x 384)")):
    dy = torch.randn(M, N, device=dev)
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    dx = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev)
    side = torch.cuda.Stream()
    ks = ops._ksplit_for(M, N * K)

    def g_dx():
        ops.gemm(1, M, K, N, dy, N, 0, 0, W, K, 0, 1, dx, K, 0)

    def g_dw():
        ops.gemm(1, N, K, M, dy, N, 0, 1, x, K, 0, 1, dW, K, 0, ksplit=ks)

    def seq():
        g_dx(); g_dw()

    def par():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g_dw()
        g_dx()
        torch.cuda.current_stream().wait_stream(side)

    for fn, label in ((seq, "sequential"), (par, "two streams"), (seq, "sequential"), (par, "two streams")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {label:12s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
