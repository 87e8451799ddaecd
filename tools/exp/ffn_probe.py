"""Time the chained feed-forward kernels alone at the train step's size (M = 927 744 rows, F = 1024): python tools/exp/ffn_probe.py [lib.so ...]
Each library (default: the product's) is loaded with ctypes and both kernels are timed with HIP events, interleaved over the libraries."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
libs = sys.argv[1:] or [os.path.join(R, "emloco_amd", "lib", "libemloco_hip.so")]
M, F, D = int(os.environ.get("FFN_M", 927744)), 1024, 128
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(M, D, device=dev); dz2 = torch.randn(M, D, device=dev)
W1 = (torch.randn(F, D, device=dev) / D ** 0.5).to(torch.bfloat16); W2 = (torch.randn(D, F, device=dev) / F ** 0.5).to(torch.bfloat16)
W2T, W1T = W2.t().contiguous(), W1.t().contiguous()
b1 = torch.randn(F, device=dev) * 0.1; b2 = torch.randn(D, device=dev) * 0.1
h = torch.empty(M, F, dtype=torch.bfloat16, device=dev); dz1 = torch.empty_like(h)
out = torch.empty(M, D, device=dev); dx = torch.empty(M, D, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
handles = []
for path in libs:
    L = C.CDLL(path)
    L.emloco_ffn_fwd.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
    L.emloco_ffn_bwd_input.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p]
    handles.append((os.path.basename(path), L))
mb = torch.empty(M, F // 32, dtype=torch.int32, device=dev)
def fwd(L, p): assert L.emloco_ffn_fwd(M, F, P(x), P(W1), P(W2), P(b1), P(b2), P(h), P(mb), P(out), p, 11, 22, st) == 0
def bwd(L, p): assert L.emloco_ffn_bwd_input(M, F, P(dz2), P(W2T), P(W1T), P(mb), P(dz1), P(dx), p, st) == 0
res = {}
for rep in range(3):
    for name, L in handles:
        for what, fn, p in (("fwd p=0.1", fwd, 0.1), ("fwd p=0", fwd, 0.0), ("bwd", bwd, 0.1)):
            fn(L, p); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn(L, p)
            e1.record(); e1.synchronize()
            res.setdefault((name, what), []).append(e0.elapsed_time(e1) / 5)
fl = 4.0 * M * D * F
for (name, what), v in res.items():
    ms = sorted(v)[len(v) // 2]
    print(f"{name:28s} {what:10s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s   (runs: {' '.join(f'{t:.3f}' for t in v)})")
