"""The large GEMM shapes of the JTA train step: this library vs hipBLASLt (torch.matmul).  EMLOCO_GEMM_BK=16|32 forces the stage depth."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
dev = torch.device("cuda:0")
M = 927744
SHAPES = [(M, 1024, 128, 0, 0, 3), (M, 1024, 128, 0, 0, 0), (M, 1024, 128, 0, 1, 0), (M, 128, 1024, 0, 0, 1), (M, 128, 1024, 0, 1, 0),
          (M, 384, 128, 0, 0, 1), (M, 128, 384, 0, 1, 0), (M, 128, 128, 0, 0, 1), (M, 128, 128, 0, 1, 0)]
SHAPES += [(1024, 128, M, 1, 1, 0), (128, 1024, M, 1, 1, 0), (384, 128, M, 1, 1, 0), (128, 128, M, 1, 1, 0)]
for m, n, k, ta, tb, fl in SHAPES:
    A = torch.randn((k, m) if ta else (m, k), device=dev); B = torch.randn((k, n) if tb else (n, k), device=dev)
    Cm = torch.empty(m, n, device=dev); bias = torch.randn(n, device=dev)
    ks = ops._ksplit_for(k, m * n) if k > 4096 else 1
    run = lambda: ops.gemm(1, m, n, k, A, m if ta else k, 0, ta, B, n if tb else k, 0, tb, Cm, n, 0, bias=bias if fl & 1 else None, flags=fl, ksplit=ks)
    run(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    Bt = B if tb else B.t()
    At = A.t() if ta else A
    torch.matmul(At, Bt); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): torch.matmul(At, Bt)
    torch.cuda.synchronize(); dtt = (time.time() - t0) / 10
    f = 2.0 * m * n * k
    print(f"m {m} n {n:5d} k {k:5d} ta {ta} tb {tb} flags {fl}: {dt*1e3:7.3f} ms {f/dt/1e12:6.1f} TFLOP/s   hipBLASLt {dtt*1e3:7.3f} ms {f/dtt/1e12:6.1f}", flush=True)
    del A, B, Cm
