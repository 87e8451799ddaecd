"""Distribution of the per-env step durations in the bench workload (run on the GPU box): python tools/exp/cost_hist.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import bench
E = 4096
env = bench.make_env(E, 0)
task = env.task
task.sim.native.set_cost_order(True)
dev = task.device
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1234)
pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
for k in range(150):
    env.reset_done(); env.step(pool[k % 64])
torch.cuda.synchronize()
lib = task.sim.native.lib
buf = (C.c_uint * E)()
st = (C.c_ulonglong * (2 * E))()
lib.emloco_sim_cost_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.c_int]
assert lib.emloco_sim_cost_ticks(task.sim.native._h, buf, st, E) == 0      # allocates the start stamps
prev = np.array(buf[:], dtype=np.float64)
for k in range(3):
    env.reset_done(); env.step(pool[k % 64])
torch.cuda.synchronize()
mid = (C.c_uint * E)()
assert lib.emloco_sim_cost_ticks(task.sim.native._h, mid, st, E) == 0
midt = np.array(mid[:], dtype=np.float64)
env.reset_done(); env.step(pool[5])
assert lib.emloco_sim_cost_ticks(task.sim.native._h, buf, st, E) == 0
t = np.array(buf[:], dtype=np.float64)
print("per-env order key: mean %.1f  median %.1f  min %.1f  p10 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (
    t.mean(), np.median(t), t.min(), np.percentile(t, 10), np.percentile(t, 90), np.percentile(t, 99), t.max()))
h, e = np.histogram(t, bins=16)
for c, a, b in zip(h, e[:-1], e[1:]):
    print("  %6.1f - %6.1f us : %5d" % (a, b, c))

t0 = np.array(st[:E], dtype=np.float64)
dur = np.array(st[E:], dtype=np.float64) - t0
key = np.array(buf[:], dtype=np.float64)
print("correlation of the duration with the order key of the same step: %.3f" % np.corrcoef(key, dur)[0, 1])
print("correlation of an env's key with its key one step earlier: %.3f" % np.corrcoef(midt, key)[0, 1])
print("sum of durations / 2048 slots = %.1f us (packing bound of the launch)" % (dur.sum() / 100 / 2048))
rel = (t0 - t0.min()) / 100.0
end = rel + dur / 100.0
print("workgroup starts [us after the first]: p1 %.1f p25 %.1f p49 %.1f p51 %.1f p75 %.1f p99 %.1f max %.1f" % tuple(np.percentile(rel, [1, 25, 49, 51, 75, 99, 100])))
print("workgroup ends   [us after the first start]: p1 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [1, 50, 90, 99, 100])))
late = np.argsort(end)[-10:]
print("the 10 last to finish: start", np.round(rel[late], 1), "duration", np.round(dur[late] / 100.0, 1))
first = rel < 50
print("first-round workgroups: %d, their durations mean %.1f; second-round: %d, durations mean %.1f" % (first.sum(), dur[first].mean() / 100, (~first).sum(), dur[~first].mean() / 100))
