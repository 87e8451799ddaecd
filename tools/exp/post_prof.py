"""Phase stamps inside post_physics_env (the last env's workgroup of the fused launch).  Needs a build with -DEMLOCO_POST_PROFILE:
EMLOCO_HIPCC_EXTRA_TASK=-DEMLOCO_POST_PROFILE python -m emloco_amd.build --force && python tools/exp/post_prof.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import bench
from emloco_amd import _lib as L
E = 4096
env = bench.make_env(E, 0); task = env.task
task.fused_chain = True
task.sim.native.set_cost_order(True)
dev = torch.device("cuda", 0)
env.reset(torch.arange(E, device=dev)); bench.stagger_episodes(env, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1)
pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
lib = L.load()
buf = (C.c_longlong * 8)()
lib.emloco_task_post_profile(buf)
names = ["load bodies + trajectory samples", "heading quaternions (2 x atan2 + sincos)", "self obs -> LDS, location obs, centre probes", "row writes + head heading",
         "height grid 32 x 32", "AMP shift (reward / flags are not in this role)", "AMP row"]
acc = []
for rep in range(20):
    for k in range(20):
        env.reset_done(); env.step(pool[k % 64])
    env.reset_done()
    lib.emloco_task_post_profile(buf)
    t = np.array(buf[:], dtype=np.int64)
    acc.append(np.diff(t))
a = np.array(acc)
print("post_physics_env of the last env inside reset_obs_kernel, ticks of 10 ns, median over", len(a), "launches with a full queue:")
for i, n in enumerate(names[:7]):
    print(f"  {n:45s} {np.median(a[:, i]):8.0f}")
print(f"  total {np.median(a.sum(1)):.0f}")
