"""sim_step_kernel alone on the bench workload (run on the GPU box): python tools/exp/kernel_time.py [lib.so ...]
For each library (default: the built one) and split setting: mean launch time over 60 launches after 150 rollout steps."""
import os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(R, "emloco_amd", "lib", "libemloco_hip.so")
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    shutil.copy(LIB, "/tmp/orig.so")
    for so in sys.argv[1:]:
        shutil.copy(so, LIB)
        for split in os.environ.get("SPLITS", "1,4").split(","):
            out = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, EMLOCO_SPLIT=split), capture_output=True, text=True)
            print(os.path.basename(so), "split", split, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
    shutil.copy("/tmp/orig.so", LIB)
    sys.exit(0)
sys.path.insert(0, R)
import numpy as np, torch
import bench
E = int(os.environ.get("ENVS", "4096"))
env = bench.make_env(E, 0)
task = env.task
dev = task.device
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1234)
pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
for k in range(150):
    env.reset_done(); env.step(pool[k % 64])
torch.cuda.synchronize()
sim = task.sim.native
if os.environ.get("COST_ORDER", "0") == "1":
    sim.set_cost_order(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for k in range(60):
    e0.record(); sim.step(2); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    env.reset_done(); env.step(pool[k % 64])          # keep the population in its steady state (resets included)
torch.cuda.synchronize()
ts = np.array(ts)
print("kernel_ms mean %.4f median %.4f min %.4f" % (ts.mean(), np.median(ts), ts.min()))
