"""configs[2] loop (frozen policy + discriminator reward + LocoVal fit) with the MAIN chain on a high-priority stream: do the rigid-body
launch's workgroups get the wave slots ahead of the deferred discriminator's GEMMs?   python tools/exp/c2_priority.py"""
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
    for mode in ("default stream", "high-priority main", "default stream", "high-priority main"):
        env = bench.make_env(E, 0)
        env.reset(torch.arange(E, device=dev))
        bench.stagger_episodes(env, seed=0)
        env.task.overlap_obs = True
        if mode.startswith("high"):
            hp = torch.cuda.Stream(device=dev, priority=-1)
            hp.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(hp):
                out = bench.locoval_policy_leg(env, E, dev, 200, 20)
            torch.cuda.current_stream(dev).wait_stream(hp)
        else:
            out = bench.locoval_policy_leg(env, E, dev, 200, 20)
        print(mode, out["value"], out["ms_per_step"], flush=True)
        del env
        torch.cuda.empty_cache()
