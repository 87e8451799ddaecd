"""Phase stamps of the pair that holds env 0 in the bench workload, step by step: prints the steps in which that pair took the fast path
(both envs <= 10 contacts) and those in which it took the full-size path.  Needs -DEMLOCO_SIM_PAIR=1 -DEMLOCO_SIM_PROFILE=1."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == "__main__":
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    task = env.task
    env.reset(torch.arange(E, device=dev))
    bench.stagger_episodes(env, seed=0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
    lib = task.sim.native.lib
    lib.emloco_sim_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    buf = (C.c_longlong * 256)()
    lib.emloco_sim_profile(task.sim.native._h, buf, 256)
    fast, slow = [], []
    prev8 = np.zeros(4)
    for k in range(200):
        env.reset_done(); env.step(pool[k % 64])
        torch.cuda.synchronize()
        lib.emloco_sim_profile(task.sim.native._h, buf, 256)
        t = np.array(buf[:], dtype=np.int64).reshape(16, 16)
        for s in range(4):
            if t[s, 10] <= t[s, 0]:
                continue
            seg = dict(kin=t[s, 1] - t[s, 0], drive=t[s, 2] - t[s, 1], fact=t[s, 3] - t[s, 2], down=t[s, 4] - t[s, 3], cand=t[s, 5] - t[s, 4],
                       tree=t[s, 9] - t[s, 8] if t[s, 8] > t[s, 5] else -1, integ=t[s, 10] - t[s, 9], total=t[s, 10] - t[s, 0])
            if t[s, 8] > t[s, 5] and t[s, 6] > t[s, 5]:          # fast path stamps are fresh
                seg.update(r6a=t[s, 6] - t[s, 5], m6b=t[s, 7] - t[s, 6], warm=t[s, 11] - t[s, 7], sweeps=t[s, 8] - t[s, 11], a7=t[s, 9] - t[s, 8])
                fast.append(seg)
            else:
                seg.update(contact=t[s, 9] - t[s, 5])
                slow.append(seg)
    for name, L in (("FAST", fast), ("FULL-SIZE", slow)):
        if not L:
            print(name, "none"); continue
        keys = list(L[0].keys())
        print(f"{name} path: {len(L)} substeps of the pair holding env 0; median ticks (10 ns):")
        print("  " + "  ".join(f"{k} {int(np.median([x[k] for x in L]))}" for k in keys))
