"""Do the frozen policy's GEMMs and the discriminator's GEMMs run faster side by side on two streams than one after the other?
(M = 4096 envs, split-mode GEMMs.)   python tools/exp/concurrent_mlp_probe.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops
dev = torch.device("cuda:0")
M = 4096
def mlp(shapes):
    ws = [(torch.randn(n, k, device=dev), torch.randn(n, device=dev)) for k, n in shapes]
    xs = [torch.randn(M, shapes[0][0], device=dev)] + [torch.empty(M, n, device=dev) for _, n in shapes]
    def run():
        for i, (w, b) in enumerate(ws):
            k, n = w.shape[1], w.shape[0]
            tiles = ((M + 127) // 128) * ((n + 31) // 32 if n <= 32 else (n + 127) // 128)
            ks = int(max(1, min(512 // max(tiles, 1), k // 128, 16)))
            ops.gemm(1, M, n, k, xs[i], k, 0, 0, w, k, 0, 0, xs[i + 1], n, 0, bias=b, flags=ops.GEMM_BIAS | ops.GEMM_RELU, ksplit=ks)
    return run
policy = mlp([(1056, 512), (512, 256)]), mlp([(624, 2048), (2048, 1024), (1024, 72)])
disc = mlp([(3092, 1024), (1024, 512), (512, 4)])
def pol(): policy[0](); policy[1]()
s2 = torch.cuda.Stream(device=dev)
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def seq(): pol(); disc()
ev = torch.cuda.Event()
def conc():
    ev.record()
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        disc()
        e2 = torch.cuda.Event(); e2.record()
    pol()
    torch.cuda.current_stream().wait_event(e2)
print(f"policy alone {timed(pol):.3f} ms, discriminator alone {timed(disc):.3f} ms, one after the other {timed(seq):.3f} ms, side by side {timed(conc):.3f} ms")
