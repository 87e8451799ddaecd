"""Is the configs[2] loop bound by the host issuing launches?  Per-step host time (no synchronisation) against the synchronised step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from emloco_amd.learning.amp_policy import AMPPolicyBundle
from emloco_amd.learning.locoval_rollout import LocoValRollout

E = 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_env(E, 0)
env.task.sim.native.set_cost_order(True)
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
env.task.overlap_obs = True
bundle = AMPPolicyBundle(env.task, seed=0)
agent = LocoValRollout(env, horizon_length=32, policy=bundle.policy, disc_reward=bundle.disc_reward, overlap_reset=False)
agent.started = True
agent._sched_live = True
for _ in range(50):
    agent.step_once()
torch.cuda.synchronize()
# (a) free running
t0 = time.perf_counter()
host = []
for k in range(300):
    a = time.perf_counter()
    agent.step_once()
    host.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(f"free running: {t_all / 300 * 1e3:.3f} ms / step; host issue time {t_issue / 300 * 1e3:.3f} ms / step (median {host[150] * 1e3:.3f}, p10 {host[30] * 1e3:.3f})")
# (b) host time of a step when the GPU is idle at the start of every step (pure issue cost)
pure = []
for k in range(100):
    torch.cuda.synchronize()
    a = time.perf_counter()
    agent.step_once()
    pure.append(time.perf_counter() - a)
pure.sort()
print(f"issue cost of one step with an empty queue: median {pure[50] * 1e3:.3f} ms, p10 {pure[10] * 1e3:.3f} ms")
