"""Long-rollout health check (run on the GPU box): 4096 envs x 3000 control steps of exploration-sized random actions with
sync-free resets; reports episode statistics and checks that every state stays finite and bounded."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import bench
E = 4096
SIG = float(sys.argv[1]) if len(sys.argv) > 1 else 0.055
BURST = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
env = bench.make_env(E, 0); dev = torch.device("cuda", 0); task = env.task
task.fused_chain = os.environ.get("EMLOCO_FUSED_CHAIN", "1") != "0"      # the fused chain between two steps (pooled resets, deferred observations)
task.sim.native.set_cost_order(True)
env.reset(torch.arange(E, device=dev))
g = torch.Generator(device=dev); g.manual_seed(0)
n_done = torch.zeros((), device=dev); ep_len_sum = torch.zeros((), device=dev); term = torch.zeros((), device=dev)
vmax = torch.zeros((), device=dev); wmax = torch.zeros((), device=dev); bad = torch.zeros((), device=dev)
rew_sum = torch.zeros((), device=dev)
t0 = time.time()
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
for k in range(N):
    env.reset_done()
    sig = SIG if k % 500 < 400 else BURST            # mostly policy-sized noise, bursts of large exploration
    obs, rew, done, info = env.step(torch.randn(E, 69, device=dev, generator=g) * sig)
    d = done != 0
    n_done += d.sum(); ep_len_sum += (task.progress_buf * d).sum(); term += (task._terminate_buf != 0).sum()
    rew_sum += rew.sum()
    rb = task._rigid_body_state.view(E, 24, 13)
    vmax = torch.maximum(vmax, rb[..., 7:10].abs().max()); wmax = torch.maximum(wmax, rb[..., 10:13].abs().max())
    bad += (~torch.isfinite(obs)).sum() + (~torch.isfinite(rb)).sum()
torch.cuda.synchronize()
dt = time.time() - t0
print(f"{N} steps x {E} envs in {dt:.1f} s ({E * N / dt / 1e6:.2f} M env-steps/s incl. the statistics)")
print(f"episodes finished {int(n_done)}, mean length {float(ep_len_sum / n_done):.1f} steps, early terminations {int(term)}, mean reward/step {float(rew_sum / (E * N)):.4f}")
print(f"max |v| {float(vmax):.2f} m/s, max |w| {float(wmax):.1f} rad/s, non-finite values {int(bad)}")
assert int(bad) == 0 and float(vmax) < 150 and float(wmax) < 400
print("ok")
