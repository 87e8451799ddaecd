"""configs[2]'s critical path with the frozen policy IN the loop, as S shards on S streams: does one shard's policy forward (matrix
pipes) run under another shard's rigid-body launch (vector pipes)?  Each shard runs the reference's order on its own stream
(reset_done -> policy(obs) -> env.step); the host issues shard after shard, so on the device the shards run out of phase.
    python tools/exp/pipe_policy_probe.py [steps]
Prints env-steps/s of all 4096 envs and the host's issue time per step (the loop is host bound if that is the step time)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch, yaml
import bench
from emloco_amd.learning.amp_network_sept_builder import AMPSeptBuilder
from emloco_amd.learning.policy_runner import FrozenPolicy
from emloco_amd.utils.running_mean_std import RunningMeanStd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
E = 4096
dev = torch.device("cuda", 0)
cfg = yaml.safe_load(open(os.path.join(R, "emloco_amd", "data", "cfg", "train", "rlg", "amp_humanoid_smpl_sept_task.yaml")))
torch.manual_seed(0)
rms = RunningMeanStd((1422,)).to(dev); rms.eval()
b = AMPSeptBuilder(); b.load(cfg["params"]["network"])
net = b.build("amp", actions_num=69, input_shape=(1422,), num_seqs=1, value_size=1, amp_input_shape=(3090,), self_obs_size=368,
              task_obs_size=1054, task_obs_size_detail={"traj": 30, "heightmap": 1024}, mean_std=rms).to(dev)
CONFIGS = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("PIPE_CONFIGS", "1:0,2:0,2:-1,4:0").split(",")]
for S, prio in CONFIGS:
    n = E // S
    shards = [bench.make_env(n, 3000 + i) for i in range(S)]
    streams = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(S)]
    pols = [FrozenPolicy(net, rms, n, dev) for _ in range(S)]
    gens = []
    for i, e in enumerate(shards):
        e.task.fused_chain = True
        e.task.overlap_obs = True
        e.task.sim.native.set_cost_order(True)
        e.reset(torch.arange(n, device=dev))
        bench.stagger_episodes(e, seed=i)
        g = torch.Generator(device=dev); g.manual_seed(7 + i); gens.append(g)
    torch.cuda.synchronize()
    def it():
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                t = shards[i].task
                shards[i].reset_done()
                if hasattr(t, "wait_obs"): t.wait_obs()
                act = pols[i].act(t.obs_buf, deterministic=False, generator=gens[i])
                shards[i].step(act)
    for k in range(40): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): it()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"shards {S} priority {prio}: {E * steps / dt / 1e6:.3f} M env-steps/s, {dt / steps * 1e3:.4f} ms per step of all {E} envs; host issue {t_issue / steps * 1e3:.4f} ms per step", flush=True)
    del shards, streams, pols
    torch.cuda.empty_cache()
