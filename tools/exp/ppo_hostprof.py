"""cProfile of the PPO update's host side (one timed train_epoch at the bench's size): where the 39 ms per optimiser step go."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import yaml
import bench
from emloco_amd.learning.amp_agent import AMPAgent

E = 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
env = bench.make_env(E, 0)
env.reset(torch.arange(E, device=dev))
cfg = yaml.safe_load(open(os.path.join(ROOT, "emloco_amd", "data", "cfg", "train", "rlg", "amp_humanoid_smpl_sept_task.yaml")))
cfg["params"]["config"]["minibatch_size"] = 2048
cfg["params"]["config"]["mini_epochs"] = 1
for a in ("fused_chain", "overlap_reset"):
    if hasattr(env.task, a):
        setattr(env.task, a, False)
agent = AMPAgent(env, cfg)
agent.train_epoch()
pr = cProfile.Profile()
pr.enable()
info = agent.train_epoch()
pr.disable()
print({k: round(v, 4) for k, v in info.items() if k in ("play_time", "update_time", "fps_step", "fps_total")})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
