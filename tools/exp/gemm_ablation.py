"""Timing of the split-mode GEMM's parts (ablated builds: tools/exp/mk_full_variant.sh gx_* -DGEMMX_*; results meaningless, timing only).
EMLOCO_LIB=variants/full_gx_NAME.so python tools/exp/gemm_ablation.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
dev = "cuda:0"
ops.set_matmul_precision("fp32_split")
M = 927744
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
out = []
for N, K in ((1024, 128), (128, 1024), (384, 128)):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
    out.append(f"N={N} K={K}: {timeit(lambda: ops.gemm(1, M, N, K, x, K, 0, 0, W, K, 0, 0, y, N, 0)):.3f}")
    del x, y
print(os.path.basename(os.environ.get("EMLOCO_LIB", "default")), " | ".join(out))
