"""One warm-up + N timed AMPAgent.train_epoch at the bench's size (4096 envs x horizon 32, shipped yaml): the workload of bench.py's
`policy.ppo` leg, for rocprofv3 (tools/exp/prof_ppo.sh).  Prints the host-side phase times of the update."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    env.reset(torch.arange(E, device=dev))
    t0 = time.time()
    out = bench.ppo_leg(env, E, dev, epochs=epochs, warmup=1)
    print({k: v for k, v in out.items() if k not in ("note", "roofline", "metric")}, "wall", round(time.time() - t0, 1))
