"""Host-side time of one graphed PPO optimiser step by segment (no device synchronisation added): where the ~0.6 ms between the graph's
own replay time and the measured update per step go.   python tools/exp/ppo_host_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from emloco_amd.learning import amp_agent as A  # noqa: E402

acc = {}


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc.setdefault(key, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, g)


wrap(A, "amp_dropout_draw", "dropout draw (CPU generator)")
wrap(A.AMPAgent, "_graph_fill", "_graph_fill (gather + draw + uploads)")
wrap(A.AMPAgent, "_replay", "_replay (graph launch)")
wrap(A.AMPAgent, "_graph_step", "_graph_step (whole host side)")
wrap(A.ppo_heads, "gather_rows", "gather_rows")

E = 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_env(E, 0)
env.reset(torch.arange(E, device=dev))
out = bench.ppo_leg(env, E, dev, epochs=2, warmup=1)
print({k: out[k] for k in ("update_ms_per_optimizer_step", "graph_trial_ms", "graph_arms")})
for k, v in acc.items():
    v = sorted(v[len(v) // 3:])
    print(f"{k:42s} calls {len(v):5d}  median {v[len(v) // 2] * 1e3:7.3f} ms  p90 {v[len(v) * 9 // 10] * 1e3:7.3f} ms")
