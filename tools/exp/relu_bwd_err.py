import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
from emloco_amd.predictor.ops import _p, _st, _chk, _lib, GEMM_BF16, GEMM_C16, GEMM_MASK16
torch.manual_seed(0)
dev = "cuda:0"
M, F, N = 1812, 1024, 128
dz2 = torch.randn(M, N, device=dev)
W2 = torch.randn(N, F, device=dev) * 0.05
h32 = torch.relu(torch.randn(M, F, device=dev))
h16 = h32.bfloat16()
lib = _lib()
def call(h, flags, dt):
    dz1 = torch.empty(M, F, dtype=dt, device=dev)
    db1 = torch.empty(F, device=dev)
    ws = torch.empty(lib.emloco_gemm_relu_bwd_workspace(M, F), device=dev)
    _chk(lib.emloco_gemm_relu_bwd(M, F, N, _p(dz2), N, _p(W2), F, 1, _p(dz1), _p(h), 1.0, _p(db1), _p(ws), flags, _st(dz2)), "x")
    return dz1.float(), db1
ref = (dz2 @ W2) * (h32 > 0)
a, da = call(h32, 0, torch.float32)
b, db = call(h32, GEMM_BF16, torch.float32)
c, dc = call(h16, GEMM_BF16 | GEMM_C16 | GEMM_MASK16, torch.bfloat16)
e = lambda x, y: ((x - y).abs().max() / y.abs().max()).item()
print("fp32 vs torch", e(a, ref), "bf16-operand vs torch", e(b, ref), "bf16-storage vs torch", e(c, ref), "db:", e(da, ref.sum(0)), e(db, ref.sum(0)), e(dc, ref.sum(0)))
bad = ((c - ref).abs() > 0.05 * ref.abs().max()).nonzero()
print("bad entries", bad.shape[0], bad[:8].tolist())
