"""Split-mode GEMM rate on the small-M shapes of the policy / discriminator forward (M = 4096) and the PPO update (M = 2048), per tile
(128 x 128 | 64 x 64) and split-K factor: HIP events around 30 launches each (run on a GPU box)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from emloco_amd.predictor import ops

dev = torch.device("cuda:0")
lib = ops._lib()
SHAPES = [  # (label, m, n, k, ta, tb)
    ("policy task1 4096x512x1056", 4096, 512, 1056, 0, 0), ("policy task2 4096x256x512", 4096, 256, 512, 0, 0),
    ("policy act1 4096x2048x624", 4096, 2048, 624, 0, 0), ("policy act2 4096x1024x2048", 4096, 1024, 2048, 0, 0),
    ("policy mu 4096x72x1024", 4096, 72, 1024, 0, 0),
    ("disc 1 4096x1024x3092", 4096, 1024, 3092, 0, 0), ("disc 2 4096x512x1024", 4096, 512, 1024, 0, 0),
    ("ppo fwd 2048x2048x624", 2048, 2048, 624, 0, 0), ("ppo fwd 2048x1024x2048", 2048, 1024, 2048, 0, 0),
    ("ppo dx 2048x2048x1024 (tb)", 2048, 2048, 1024, 0, 1), ("ppo dW 1024x2048x2048 (ta,tb)", 1024, 2048, 2048, 1, 1),
    ("ppo dW 2048x624x2048 (ta,tb)", 2048, 624, 2048, 1, 1), ("ppo disc dW 1024x3092x2048 (ta,tb)", 1024, 3092, 2048, 1, 1),
]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, m, n, k, ta, tb in SHAPES:
    A = torch.randn((k, m) if ta else (m, k), device=dev)
    B = torch.randn((k, n) if tb else (n, k), device=dev)
    Cm = torch.empty(m, n, device=dev)
    lda = m if ta else k; ldb = n if tb else k
    best = {}
    line = []
    for small in (0, 1):
        lib.emloco_gemm_set_small_tile(small)
        for ks in (1, 2, 3, 4, 8):
            if k // ks < 128:
                continue
            run = lambda: ops.gemm(1, m, n, k, A, lda, 0, ta, B, ldb, 0, tb, Cm, n, 0, ksplit=ks)
            for _ in range(3): run()
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            line.append(f"{'s' if small else 'L'}{ks}:{us:6.1f}")
            best[(small, ks)] = us
    b = min(best, key=best.get)
    fl = 2.0 * m * n * k
    print(f"{label:36s} " + " ".join(line) + f"   best {'small' if b[0] else 'large'} ks{b[1]} {best[b]:.1f} us = {fl / best[b] / 1e6:.0f} TFLOP/s", flush=True)
lib.emloco_gemm_set_small_tile(-1)
