"""Segment profile of the two-envs-per-wave rigid-body kernel over ALL pairs of the bench workload (PACC accumulators).
Needs EMLOCO_HIPCC_EXTRA_SIM="-DEMLOCO_SIM_PAIR=1 -DEMLOCO_SIM_PROFILE=1":  python tools/exp/pair_profile.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    task = env.task
    task.sim.native.set_cost_order(True)
    env.reset(torch.arange(E, device=dev))
    bench.stagger_episodes(env, seed=0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
    for k in range(100):
        env.reset_done(); env.step(pool[k % 64])
    lib = task.sim.native.lib
    lib.emloco_sim_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    buf = (C.c_longlong * 256)()
    lib.emloco_sim_profile(task.sim.native._h, buf, 256)          # allocates the (zeroed) buffer
    for k in range(N):
        env.reset_done(); env.step(pool[k % 64])
    torch.cuda.synchronize()
    lib.emloco_sim_profile(task.sim.native._h, buf, 256)
    names = ["1 kinematics + 1b (joint)", "2 drive (joint)", "2b-3 inertia, factorise (joint)", "4 down pass (joint)", "5 candidates (joint)",
             "fast: 6a rows / chain y", "fast: 6b matrices -> registers", "fast: 6c setup + warm start", "fast: 6c sweeps", "fast: 7a impulses",
             "full-size path: 6a-7a, both envs", "7 tree passes (joint)", "8 integrate (joint)"]
    tt, cnt = np.array(buf[128:128 + 13], dtype=np.float64), np.array(buf[160:160 + 13], dtype=np.float64)
    n = cnt[0]
    print(f"pair-substeps profiled: {int(n)}; fast path {int(cnt[9])} ({cnt[9] / max(n, 1):.3f}), full-size path {int(cnt[10])} ({cnt[10] / max(n, 1):.3f})")
    print("segment: mean ticks (10 ns) when it runs | share of all pair-substep time")
    for i, nm in enumerate(names):
        if cnt[i] > 0:
            print(f"  {nm:36s} {tt[i] / cnt[i]:8.1f}   {tt[i] / tt.sum():6.3f}")
    print(f"  mean pair-substep: {tt.sum() / max(n, 1):.1f} ticks")
