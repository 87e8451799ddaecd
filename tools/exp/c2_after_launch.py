"""configs[2] loop: the deferred discriminator's side-stream launches issued AHEAD of the rigid-body launch (round 4/5) or BEHIND it
(round 6, the task's after_physics_launch hook); interleaved in one process.   python tools/exp/c2_after_launch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for mode in ("0", "1", "0", "1", "0", "1"):
        os.environ["EMLOCO_DISC_AFTER_LAUNCH"] = mode
        env = bench.make_env(E, 0)
        env.reset(torch.arange(E, device=dev))
        bench.stagger_episodes(env, seed=0)
        env.task.overlap_obs = True
        out = bench.locoval_policy_leg(env, E, dev, 200, 20)
        print("EMLOCO_DISC_AFTER_LAUNCH=" + mode, out["value"], out["ms_per_step"], "deferred", out["discriminator_deferred"], flush=True)
        env.task.after_physics_launch = None
        del env
        torch.cuda.empty_cache()
