"""TFLOP/s of emloco_gemm_f32 on the shapes the predictor and the policy use (run on a GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops

dev = torch.device("cuda:0")
# (label, batch, m, n, k, ta, tb)
SHAPES = [
    ("qkv proj      58k x 384 x 128", 1, 57984, 384, 128, 0, 0),
    ("ffn up        58k x 1024 x 128", 1, 57984, 1024, 128, 0, 0),
    ("ffn down      58k x 128 x 1024", 1, 57984, 128, 1024, 0, 0),
    ("scores  512 x (453 x 453 x 32)", 512, 453, 453, 32, 0, 0),
    ("P.V     512 x (453 x 32 x 453)", 512, 453, 32, 453, 0, 1),
    ("dW      1024 x 128 x 58k (ta)", 1, 1024, 128, 57984, 1, 1),
    ("policy  4096 x 2048 x 624", 1, 4096, 2048, 624, 0, 0),
    ("policy  4096 x 1024 x 2048", 1, 4096, 1024, 2048, 0, 0),
    ("policy  4096 x 512 x 1056", 1, 4096, 512, 1056, 0, 0),
    ("square  4096^3", 1, 4096, 4096, 4096, 0, 0),
]
if len(sys.argv) > 1:
    ops.set_matmul_precision(sys.argv[1])          # "bf16": operands rounded to bf16 into the matrix cores (opt-in)
print("matmul precision:", ops.get_matmul_precision())
for label, b, m, n, k, ta, tb in SHAPES:
    A = torch.randn((b, k, m) if ta else (b, m, k), device=dev)
    B = torch.randn((b, k, n) if tb else (b, n, k), device=dev)
    Cm = torch.empty(b, m, n, device=dev)
    lda = m if ta else k; ldb = n if tb else k
    ks = ops._ksplit_for(k, m * n) if k > 4096 else 1
    def run():
        ops.gemm(b, m, n, k, A, lda, m * k, ta, B, ldb, n * k, tb, Cm, n, m * n, ksplit=ks)
    run(); torch.cuda.synchronize()
    if b * m * n * k < 2e10:
        ref = torch.matmul(A.transpose(1, 2) if ta else A, B if tb else B.transpose(1, 2))
        err = ((Cm - ref).abs().max() / ref.abs().max()).item()
    else:
        err = float("nan")
    n_it = 10
    t0 = time.time()
    for _ in range(n_it): run()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n_it
    fl = 2.0 * b * m * n * k
    t0 = time.time()
    for _ in range(n_it): torch.matmul(A.transpose(1, 2) if ta else A, B if tb else B.transpose(1, 2))
    torch.cuda.synchronize()
    dt_t = (time.time() - t0) / n_it
    print(f"{label:34s} ksplit {ks:2d}: {dt*1e3:8.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s   (hipBLASLt via torch: {fl/dt_t/1e12:6.1f})  rel err {err:.1e}", flush=True)
