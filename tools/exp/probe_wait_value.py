"""Does hipStreamWaitValue32 order a side stream behind a value written by a kernel on another stream, and what does it cost?"""
import ctypes as C
import time
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
dev = torch.device("cuda", 0)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
out = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
ok = True
for t in range(1, 6):
    # main: a long kernel, then the flag write; side: wait for flag >= t, then copy the flag
    with torch.cuda.stream(side):
        rc = hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), C.c_void_p(flag.data_ptr()), t, 0, 0xFFFFFFFF)
        out.copy_(flag)
    y = big @ big
    flag.fill_(t)
    side.synchronize()
    print("rc", rc, "t", t, "side saw", int(out.item()))
    ok = ok and rc == 0 and int(out.item()) == t
print("ordering ok:", ok)
# cost on the main stream: N x (tiny kernel) with and without a side stream waiting on values
def loop(n, use):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if use:
            hip.hipStreamWaitValue32(C.c_void_p(side.cuda_stream), C.c_void_p(flag.data_ptr()), 100 + i, 0, 0xFFFFFFFF)
        flag.fill_(100 + i)
        out.add_(1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("us per iteration: plain %.2f, with a waiting side stream %.2f" % (loop(2000, False), loop(2000, True)))
