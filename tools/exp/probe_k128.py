"""Where does the time of the K = 128 split-mode GEMMs go?  The feed-forward's first layer (927 744 x 1024 x 128) with and without its
epilogue pieces, against the same launch with a narrower output (N = 128) and longer reductions.   python tools/exp/probe_k128.py"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402
from emloco_amd.predictor import ops  # noqa: E402

dev = "cuda:0"
ops.set_matmul_precision("fp32_split")
M = 927744


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for N, K in ((1024, 128), (1024, 128), (384, 128), (128, 128), (128, 1024), (1024, 256), (1024, 512)):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    img = ops.split_image(W, N, K, K, 0)
    for name, kw in (("plain", dict()), ("bias+relu", dict(bias=b, flags=ops.GEMM_BIAS | ops.GEMM_RELU)),
                     ("bias+relu+dropout", dict(bias=b, flags=ops.GEMM_BIAS | ops.GEMM_RELU, drop_p=0.1, drop_seed=5)),
                     ("plain, weight image", dict(b_image=img)), ("plain again", dict()),
                     ("bias+relu+dropout, image", dict(bias=b, flags=ops.GEMM_BIAS | ops.GEMM_RELU, drop_p=0.1, drop_seed=5, b_image=img))):
        t = timeit(lambda: ops.gemm(1, M, N, K, x, K, 0, 0, W, K, 0, 0, y, N, 0, **kw))
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + N * K + M * N)
        print(f"M={M} N={N:5d} K={K:5d} {name:22s} {t:7.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s  {by / t / 1e9:6.2f} TB/s algorithmic")
    del x, y
