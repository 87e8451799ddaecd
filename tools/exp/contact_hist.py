"""Histogram of ground contacts per env-substep over a soak of the headline loop.
Needs a library built with EMLOCO_HIPCC_EXTRA_SIM="-DEMLOCO_SIM_NCHIST=1":  python tools/exp/contact_hist.py [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    task = env.task
    task.sim.native.set_cost_order(True)
    env.reset(torch.arange(E, device=dev))
    bench.stagger_episodes(env, seed=0)
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
    k = [0]

    def pol(obs):
        k[0] += 1
        return pool[k[0] % 64]
    agent = LocoValRollout(env, horizon_length=32, policy=pol)
    lib = task.sim.native.lib
    lib.emloco_sim_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
    buf = (C.c_longlong * 256)()
    lib.emloco_sim_profile(task.sim.native._h, buf, 256)          # first call allocates the (zeroed) buffer
    for i in range(N):
        agent.step_once()
        if (i + 1) % 32 == 0:
            agent.end_epoch()
    torch.cuda.synchronize()
    lib.emloco_sim_profile(task.sim.native._h, buf, 256)
    h = np.array(buf[:], dtype=np.int64)
    cand, kept = h[:64], h[64:64 + 21]
    tot = int(kept.sum())
    print(f"{N} steps x {E} envs of the headline loop (N(0, 0.055^2) actions, steady-state resets): {tot} env-substeps "
          f"(= steps x envs x 4: {N * E * 4}; the id-list launches of freshly reset envs included)")
    print("contacts KEPT per env-substep (EMLOCO_MAXC = 20):  count  share  cumulative")
    c = 0.0
    for n in range(21):
        c += kept[n] / tot
        print(f"  {n:2d}  {int(kept[n]):12d}  {kept[n] / tot:8.5f}  {c:8.5f}")
    print("candidates inside contact_offset BEFORE the cut to 20 (63 = 63 or more):")
    for n in range(64):
        if cand[n]:
            print(f"  {n:2d}  {int(cand[n]):12d}  {cand[n] / tot:8.5f}")
    over16 = cand[17:].sum() / tot
    over20 = cand[21:].sum() / tot
    mean = float((np.arange(64) * cand).sum() / tot)
    print(f"mean candidates {mean:.2f}; more than 16: {over16:.5f} of the env-substeps; more than 20 (shallowest dropped): {over20:.5f}")
