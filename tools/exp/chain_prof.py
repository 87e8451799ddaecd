"""Phase stamps of the fused reset / observation launch (emloco_task_reset_obs) inside the env.step loop of bench.py's config.
Run on a GPU box: python tools/exp/chain_prof.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from emloco_amd import _lib as L
E = 4096
env = bench.make_env(E, 0)
task = env.task
task.fused_chain = True
task.sim.native.set_cost_order(True)
dev = torch.device("cuda", 0)
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1)
pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
lib = L.load()
buf = (C.c_longlong * 16)()
lib.emloco_task_chain_profile(buf)
names = ["row", "sample", "kinematics", "finish", "observations"]
acc = []
for k in range(120):
    env.reset_done(); env.step(pool[k % 64])
    if k >= 20:
        lib.emloco_task_chain_profile(buf)
        t = np.array(buf[:], dtype=np.int64)
        acc.append(t.copy())
a = np.array(acc)
t0 = a[:, 0]
print("reset slot 0 (ticks of 10 ns, mean over", len(a), "launches):")
for i, n in enumerate(names):
    print(f"  {n:14s} {np.mean(a[:, i + 1] - a[:, i]):8.1f}")
print(f"  chain total    {np.mean(a[:, 5] - a[:, 0]):8.1f}")
print(f"  history row 14: start +{np.mean(a[:, 6] - t0):.1f}, duration {np.mean(a[:, 7] - a[:, 6]):.1f}")
print(f"  observation workgroup of env 0: start +{np.mean(a[:, 8] - t0):.1f}, duration {np.mean(a[:, 9] - a[:, 8]):.1f}")
print(f"  observation workgroup of env {E - 1}: start +{np.mean(a[:, 10] - t0):.1f}, duration {np.mean(a[:, 11] - a[:, 10]):.1f}, end +{np.mean(a[:, 11] - t0):.1f}")
print(f"  pool draw, last entry: sample start +{np.mean(a[:, 12] - t0):.1f} end +{np.mean(a[:, 13] - t0):.1f}; trajectory start +{np.mean(a[:, 14] - t0):.1f} end +{np.mean(a[:, 15] - t0):.1f}")

# the same stamps WITHOUT a host synchronisation per step (the launch queue stays full, as in bench.py): last launch of 60
cnt = []
for rep in range(5):
    for k in range(60):
        env.reset_done(); env.step(pool[k % 64])
        cnt.append(task._done_ids[E:E + 1].clone())
    lib.emloco_task_chain_profile(buf)
    t = np.array(buf[:], dtype=np.int64); z = t[0]
    print("queued: chain phases", [int(t[i + 1] - t[i]) for i in range(5)], "chain end +%d" % (t[5] - z), "hist14 +%d..+%d" % (t[6] - z, t[7] - z),
          "last env obs +%d..+%d" % (t[10] - z, t[11] - z), "pool sample +%d..+%d traj +%d..+%d" % (t[12] - z, t[13] - z, t[14] - z, t[15] - z))
c = torch.cat(cnt).cpu().numpy()
print("finished envs per step: mean %.1f, p50 %d, p90 %d, p99 %d, max %d" % (c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max()))
