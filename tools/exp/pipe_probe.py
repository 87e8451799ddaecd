"""The 4096 envs of bench.py's config as S independent shards on S HIP streams (double-buffered rollout): each shard runs the
reference's order (reset_done -> [policy] -> env.step) on its own stream, so one shard's latency-bound chain between two rigid-body
launches overlaps the other shards' rigid-body kernels.   python tools/exp/pipe_probe.py [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
E = 4096
dev = torch.device("cuda", 0)
for S, prio in ((1, 0), (2, 0), (2, -1), (4, 0), (4, -1)):
    n = E // S
    shards = [bench.make_env(n, 2000 + i) for i in range(S)]
    streams = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(S)]
    g = torch.Generator(device=dev); g.manual_seed(4321)
    pools = [torch.randn(64, n, 69, device=dev, generator=g) * float(np.exp(-2.9)) for _ in range(S)]
    for i, e in enumerate(shards):
        e.task.fused_chain = os.environ.get("EMLOCO_FUSED_CHAIN", "1") != "0"
        e.task.sim.native.set_cost_order(True)
        e.reset(torch.arange(n, device=dev))
        bench.stagger_episodes(e, seed=i)
    torch.cuda.synchronize()
    def it(k):
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                shards[i].reset_done()
                shards[i].step(pools[i][k % 64])
    for k in range(30): it(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): it(k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"shards {S} priority {prio}: {E * steps / dt / 1e6:.3f} M env-steps/s, {dt / steps * 1e3:.4f} ms per step of all {E} envs", flush=True)
    del shards, streams, pools
    torch.cuda.empty_cache()
