"""fp32 MFMA vs the split mode (fp32 products from bf16 pieces) on the shapes of the predictor's train step and the policy:
time, TFLOP/s (fp32-equivalent) and error against float64.   python tools/exp/probe_split.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops

dev = torch.device("cuda:0")
M = 927744                                    # rows of the local encoder at batch 256 (tools/exp/jta_shapes.py)
SHAPES = [  # label, m, n, k, ta, tb, ksplit
    ("qkv proj     M x 384 x 128", M, 384, 128, 0, 0, 1),
    ("ffn up       M x 1024 x 128", M, 1024, 128, 0, 0, 1),
    ("ffn down     M x 128 x 1024", M, 128, 1024, 0, 0, 1),
    ("dX (tb)      M x 128 x 384", M, 128, 384, 0, 1, 1),
    ("dW (ta,tb)   1024 x 128 x M", 1024, 128, M, 1, 1, 64),
    ("dW (ta,tb)   128 x 1024 x M", 128, 1024, M, 1, 1, 64),
    ("policy  4096 x 512 x 1056", 4096, 512, 1056, 0, 0, 4),
    ("policy  4096 x 256 x 512", 4096, 256, 512, 0, 0, 4),
    ("policy  4096 x 2048 x 624", 4096, 2048, 624, 0, 0, 1),
    ("policy  4096 x 1024 x 2048", 4096, 1024, 2048, 0, 0, 2),
    ("policy  4096 x 69 x 1024", 4096, 69, 1024, 0, 0, 8),
    ("square  4096^3", 4096, 4096, 4096, 0, 0, 1),
]
for label, m, n, k, ta, tb, ks in SHAPES:
    A = torch.randn((k, m) if ta else (m, k), device=dev)
    B = torch.randn((k, n) if tb else (n, k), device=dev)
    Cm = torch.empty(m, n, device=dev)
    lda = m if ta else k; ldb = n if tb else k
    line = f"{label:30s} ks {ks:2d}:"
    # float64 reference on a sample of rows
    rows = torch.randint(0, m, (64,), device=dev)
    Ar = (A[:, rows].T if ta else A[rows]).double()
    ref = Ar @ (B.double() if tb else B.double().T)
    mag = Ar.abs() @ (B.double().abs() if tb else B.double().abs().T)
    for mode in ("fp32", "fp32_split"):
        ops.set_matmul_precision(mode)
        def run():
            ops.gemm(1, m, n, k, A, lda, 0, ta, B, ldb, 0, tb, Cm, n, 0, ksplit=ks)
        run(); torch.cuda.synchronize()
        err = ((Cm[rows].double() - ref).abs() / mag).max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_it = 10
        e0.record()
        for _ in range(n_it): run()
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / n_it * 1e-3
        line += f"  {mode:10s} {dt*1e3:8.3f} ms {2.0*m*n*k/dt/1e12:6.1f} TF err {err:.1e} |"
    print(line, flush=True)
