"""Soak: the headline loop for N steps -- no device error word, finite state, a steady reset rate, throughput stable over the run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import numpy as np  # noqa: E402

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    E = 4096
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    env = bench.make_env(E, 0)
    task = env.task
    task.sim.native.set_cost_order(True)
    env.reset(torch.arange(E, device=dev))
    bench.stagger_episodes(env, seed=0)
    from emloco_amd.learning.locoval_rollout import LocoValRollout
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    pool = torch.randn(64, E, 69, device=dev, generator=g) * float(np.exp(-2.9))
    k = [0]

    def pol(obs):
        k[0] += 1
        return pool[k[0] % 64]
    agent = LocoValRollout(env, horizon_length=32, policy=pol, overlap_reset=os.environ.get("EMLOCO_SOAK_OVERLAP_RESET", "0") == "1")
    resets, rates = 0, []
    t0 = time.perf_counter()
    for i in range(N):
        agent.step_once()
        if (i + 1) % 32 == 0:
            agent.end_epoch()
        if (i + 1) % 2000 == 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rates.append(2000 * E / (t1 - t0))
            rs = task._root_states
            ok = bool(torch.isfinite(rs).all()) and bool(torch.isfinite(task.obs_buf).all()) and bool(torch.isfinite(task.rew_buf).all())
            task.sim.native.sync()                       # raises on a device error word
            print(f"step {i + 1}: {rates[-1] / 1e6:.3f} M env-steps/s, finite {ok}, max |root| {rs.abs().max().item():.1f}, "
                  f"fitted episodes {agent.fitted_episodes}, loss {agent.vnet_loss:.4f}", flush=True)
            assert ok
            t0 = time.perf_counter()
    print("rates M/s:", [round(r / 1e6, 3) for r in rates])
