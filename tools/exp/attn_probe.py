"""Time the bf16-in-memory attention kernels alone at the train step's size (2048 person-sequences x 453 tokens x 4 heads, p = 0.1):
python tools/exp/attn_probe.py [lib.so ...] -- forward, and backward (dQ + dK/dV launches together), HIP events, interleaved over the libraries.
EMLOCO_ATTN16_OLD=1 in the environment selects round 4's kernels inside every library."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
libs = sys.argv[1:] or [os.path.join(R, "emloco_amd", "lib", "libemloco_hip.so")]
n_seq, S, H, d = int(os.environ.get("ATTN_NSEQ", 2048)), 453, 4, 128
dev = torch.device("cuda", 0)
torch.manual_seed(0)
SPLIT = os.environ.get("ATTN_MODE", "bf16") == "split"       # the default precision class: fp32 q|k|v, tile products from bf16 pieces
qkv = (torch.randn(n_seq, S, 3 * d, device=dev) * 0.7)
if not SPLIT: qkv = qkv.to(torch.bfloat16)
kb = torch.zeros(n_seq, S, device=dev); kb[:, 400:] = 1.0
out = torch.empty(n_seq, S, d, device=dev); lse = torch.empty(n_seq * H, S, device=dev); dsum = torch.empty_like(lse)
dout = torch.randn(n_seq, S, d, device=dev); dqkv = torch.empty_like(qkv)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FL = 64 if SPLIT else (16 | 32)       # EMLOCO_ATTN_SPLIT, or EMLOCO_ATTN_BF16 | EMLOCO_ATTN_QKV_BF16MEM
handles = []
for path in libs:
    L = C.CDLL(path)
    L.emloco_attention_fwd_queries.argtypes = [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_uint32, C.c_void_p]
    L.emloco_attention_bwd_queries.argtypes = [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_uint32, C.c_void_p]
    handles.append((os.path.basename(path), L))
sc = 1.0 / 32 ** 0.5
PD = float(os.environ.get("ATTN_P", 0.1))
def fwd(L): assert L.emloco_attention_fwd_queries(n_seq, S, S, H, d, sc, P(qkv), P(kb), P(out), P(lse), FL, PD, 99, st) == 0
def bwd(L): assert L.emloco_attention_bwd_queries(n_seq, S, S, H, d, sc, P(qkv), P(kb), P(out), P(lse), P(dout), P(dqkv), P(dsum), FL, PD, 99, st) == 0
res = {}
for rep in range(3):
    for name, L in handles:
        for what, fn in (("fwd", fwd), ("bwd", bwd)):
            fn(L); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn(L)
            e1.record(); e1.synchronize()
            res.setdefault((name, what), []).append(e0.elapsed_time(e1) / 3)
for (name, what), v in res.items():
    print(f"{name:28s} {what:4s} {sorted(v)[len(v) // 2]:7.3f} ms   (runs: {' '.join(f'{t:.3f}' for t in v)})")
