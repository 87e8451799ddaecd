"""The configs[2] loop (frozen policy + AMP discriminator reward + LocoVal fit, bench.py `policy.with_discriminator_and_locoval_fit`)
for a kernel trace: python tools/exp/env_disc_loop.py [steps]   (run under rocprofv3 --kernel-trace, tools/exp/prof_env_disc.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
E = 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_env(E, 0)
env.task.sim.native.set_cost_order(True)
env.reset(torch.arange(E, device=dev))
bench.stagger_episodes(env, seed=0)
env.task.overlap_obs = True
out = bench.locoval_policy_leg(env, E, dev, steps, 20)
print({k: v for k, v in out.items() if k != "note"})
