"""configs[2] loop: is it the host or the GPU?  Host time to ISSUE 200 steps (loop returns) against the time until the GPU has finished them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from emloco_amd.learning.amp_policy import AMPPolicyBundle  # noqa: E402
from emloco_amd.learning.locoval_rollout import LocoValRollout  # noqa: E402

if __name__ == "__main__":
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    env.reset(torch.arange(E, device=dev))
    bench.stagger_episodes(env, seed=0)
    env.task.overlap_obs = True
    bundle = AMPPolicyBundle(env.task, seed=0)
    agent = LocoValRollout(env, horizon_length=32, policy=bundle.policy, disc_reward=bundle.disc_reward, overlap_reset=False)
    agent.started = True
    agent._sched_live = True
    for k in range(40):
        agent.step_once()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(200):
            agent.step_once()
            if (k + 1) % 32 == 0:
                agent.end_epoch()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"issue {1e3 * (t1 - t0) / 200:.4f} ms/step, done {1e3 * (t2 - t0) / 200:.4f} ms/step, GPU backlog at loop end {1e3 * (t2 - t1):.2f} ms", flush=True)
