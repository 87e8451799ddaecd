"""Does the feed-forward block of the fp32 class run faster in ROW CHUNKS whose hidden slab (rows x 1024 fp32) stays in the 256 MB Infinity
Cache between the launch that writes it and the launches that read it?   python tools/exp/ffn_chunk_probe.py
forward: h = relu(x W1^T + b1) [dropout], f = h W2^T + b2;  backward: dz1 = (dz2 W2) o [h > 0] (+ column sums), dW2 = dz2^T h, dx = dz1 W1, dW1 = dz1^T x"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
dev = "cuda:0"
ops.set_matmul_precision("fp32_split")
M, K, F = 927744, 128, 1024
lib = ops._lib()
x = torch.randn(M, K, device=dev); W1 = torch.randn(F, K, device=dev) * 0.05; b1 = torch.randn(F, device=dev) * 0.1
W2 = torch.randn(K, F, device=dev) * 0.05; b2 = torch.randn(K, device=dev) * 0.1
h = torch.empty(M, F, device=dev); f = torch.empty(M, K, device=dev)
dz2 = torch.randn(M, K, device=dev); dz1 = torch.empty(M, F, device=dev); dx = torch.empty(M, K, device=dev)
dW1 = torch.zeros(F, K, device=dev); dW2 = torch.zeros(K, F, device=dev); db1 = torch.empty(F, device=dev)
ws = torch.empty(lib.emloco_gemm_relu_bwd_workspace(M, F), dtype=torch.float32, device=dev)
S2 = ops.GEMM_SPLIT2
P = ops._p

def fwd(rows):
    for r0 in range(0, M, rows):
        m = min(rows, M - r0)
        ops.gemm(1, m, F, K, x, K, 0, 0, W1, K, 0, 0, h, F, 0, bias=b1, flags=ops.GEMM_BIAS | ops.GEMM_RELU, drop_p=0.1, drop_seed=5 + r0, a_off=r0 * K, c_off=r0 * F)
        ops.gemm(1, m, K, F, h, F, 0, 0, W2, F, 0, 0, f, K, 0, bias=b2, flags=ops.GEMM_BIAS, drop_p=0.1, drop_seed=7 + r0, a_off=r0 * F, c_off=r0 * K)

def bwd(rows):
    first = True
    for r0 in range(0, M, rows):
        m = min(rows, M - r0)
        acc = 0 if first else ops.GEMM_ACC
        ops._chk(lib.emloco_gemm_relu_bwd(m, F, K, P(dz2, r0 * K), K, P(W2), F, 1, P(dz1, r0 * F), P(h, r0 * F), 1.0 / 0.9, P(db1), P(ws), ops.GEMM_SPLIT | S2, ops._st(x)), "relu_bwd")
        ops.gemm(1, K, F, m, dz2, K, 0, 1, h, F, 0, 1, dW2, F, 0, ksplit=ops._ksplit_for(m, K * F), flags=acc | S2, a_off=r0 * K, b_off=r0 * F)
        ops.gemm(1, m, K, F, dz1, F, 0, 0, W1, K, 0, 1, dx, K, 0, flags=S2, a_off=r0 * F, c_off=r0 * K)
        ops.gemm(1, F, K, m, dz1, F, 0, 1, x, K, 0, 1, dW1, K, 0, ksplit=ops._ksplit_for(m, F * K), flags=acc | S2, a_off=r0 * F, b_off=r0 * K)
        first = False

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

for rows in (M, 262144, 131072, 65536, 49152, 32768, 16384):
    tf = timeit(lambda: fwd(rows)); tb = timeit(lambda: bwd(rows))
    print(f"rows per chunk {rows:7d} (hidden slab {rows * F * 4 / 2**20:7.1f} MB): forward {tf:6.3f} ms   backward {tb:6.3f} ms")
