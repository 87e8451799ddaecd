"""Duration of post_physics_kernel per mode subset at the bench configuration (run on the GPU box): python tools/exp/post_time.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from emloco_amd import _lib as L
E = 4096
env = bench.make_env(E, 0)
task = env.task
dev = task.device
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
g = torch.Generator(device=dev); g.manual_seed(1)
for k in range(30):
    env.reset_done(); env.step(torch.randn(E, 69, device=dev, generator=g) * 0.055)
torch.cuda.synchronize()
modes = {"all (POST_STEP)": L.POST_STEP, "advance+reward+reset": L.POST_ADVANCE | L.POST_REWARD | L.POST_RESET, "obs": L.POST_OBS,
         "amp shift": L.POST_AMP_SHIFT, "amp row": L.POST_AMP_ROW, "amp shift+row": L.POST_AMP_SHIFT | L.POST_AMP_ROW,
         "obs+amp": L.POST_OBS | L.POST_AMP_SHIFT | L.POST_AMP_ROW}
prog = task.progress_buf.clone()
for name, m in modes.items():
    ts = []
    for rep in range(12):
        task.progress_buf.copy_(prog)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); task._launch_post(m); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:24s} {np.median(ts[2:]):7.1f} us")
