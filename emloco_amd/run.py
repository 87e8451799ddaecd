"""run.py: entry point with the reference's surface (pacer/pacer/run.py): flags, create_rlgpu_env, RLGPUEnv.

    python -m emloco_amd.run --num_envs 4096 --random_heading --init_heading --heading_inversion \
        --adjust_root_vel --input_init_pose --input_init_vel --steps 200

rl_games (the PPO runner the reference plugs into, pacer/requirements.txt:26) is not vendored; this entry
runs the LocoVal rollout loop of `learning/locoval_rollout.py` (AMPValueAgent.play_steps bookkeeping) with a
frozen random-init policy, and prints the reference's `fps_step` counter (common_agent.py:187).
"""
import random
import sys
import time

import numpy as np
import torch

from .utils.config import get_args, load_cfg, parse_sim_params
from .utils.flags import flags
from .utils.parse_task import parse_task


def fill_flags(args):
    """run.py:263-331."""
    flags.debug, flags.follow, flags.fixed = args.debug, args.follow, True
    flags.divide_group = flags.no_collision_check = flags.fixed_path = False
    flags.real_path = flags.jta_path = flags.jrdb_path = False
    flags.pred_path, flags.small_terrain = args.pred_path, args.small_terrain
    flags.show_traj = flags.render = False
    flags.server_mode, flags.slow, flags.height_debug = args.server_mode, False, False
    flags.random_heading, flags.no_virtual_display = args.random_heading, args.no_virtual_display
    flags.init_heading, flags.heading_inversion = args.init_heading, args.heading_inversion
    flags.adjust_root_vel, flags.input_init_pose = args.adjust_root_vel, args.input_init_pose
    flags.add_noise, flags.vru, flags.add_proj = args.add_noise, args.vru, args.add_proj
    if args.real_path != "":
        flags.real_path = True
        flags.jta_path = "JTA" in args.real_path
        flags.jrdb_path = "JRDB" in args.real_path


def create_rlgpu_env(args, cfg, cfg_train, rank=0):
    """run.py:57-86; per-rank seed offset as run.py:65."""
    seed = cfg_train.get("seed", -1)
    if seed is not None and seed >= 0:
        seed = seed + rank
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
    sim_params = parse_sim_params(args, cfg, cfg_train)
    task, env = parse_task(args, cfg, cfg_train, sim_params)
    return env


class RLGPUEnv:
    """run.py:135-182 without the rl_games base class."""

    def __init__(self, env):
        self.env = env
        self.use_global_obs = self.env.num_states > 0
        self.full_state = {}
        self.full_state["obs"] = self.reset()

    def step(self, action):
        next_obs, reward, is_done, info = self.env.step(action)
        self.full_state["obs"] = next_obs
        return self.full_state["obs"], reward, is_done, info

    def reset(self, env_ids=None):
        self.full_state["obs"] = self.env.reset(env_ids)
        return self.full_state["obs"]

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()


def main(argv=None):
    from . import configure_runtime
    configure_runtime()                                  # entry point: 16 hardware queues, ahead of the first GPU call (emloco_amd/__init__.py)
    argv = list(sys.argv[1:] if argv is None else argv)
    steps = 100
    if "--steps" in argv:
        i = argv.index("--steps")
        steps = int(argv[i + 1])
        del argv[i:i + 2]
    policy_ckpt, use_policy = None, False
    if "--policy_checkpoint" in argv:            # rl_games-layout checkpoint of the frozen PACER policy (config 3)
        i = argv.index("--policy_checkpoint")
        policy_ckpt, use_policy = argv[i + 1], True
        del argv[i:i + 2]
    if "--policy_random_init" in argv:           # same architecture, random weights (no checkpoint ships)
        use_policy = True
        argv.remove("--policy_random_init")
    train_policy = "--train_policy" in argv        # configs[1]: PPO + AMP pretraining of the policy (pacer/run.py without --test)
    if train_policy:
        argv.remove("--train_policy")
    # one process per GPU (torch.distributed.run): the reference's --horovod launch (run.py:57-66, common_agent.py:165-180).
    # num_envs is the GLOBAL count: rank r simulates the contiguous shard [r * E / W, (r + 1) * E / W) on cuda:LOCAL_RANK with
    # seed + rank; the learners exchange gradients / statistics through emloco_amd.dist
    from .dist import init_from_env, shard_range
    rank, local_rank, world = init_from_env()
    if world > 1:
        for flag in ("--sim_device", "--rl_device"):
            if flag in argv:
                i = argv.index(flag)
                del argv[i:i + 2]
        argv += ["--sim_device", f"cuda:{local_rank}", "--rl_device", f"cuda:{local_rank}"]
        torch.cuda.set_device(local_rank)
    args = get_args(argv)
    if world > 1:
        args.num_envs = shard_range(int(args.num_envs), rank, world)[1]
    cfg, cfg_train, _ = load_cfg(args)
    fill_flags(args)
    env = RLGPUEnv(create_rlgpu_env(args, cfg, cfg_train, rank=rank))
    say = print if rank == 0 else (lambda *a, **k: None)
    if train_policy:
        import yaml
        from .learning.amp_agent import AMPAgent
        from .learning.amp_policy import DEFAULT_CFG
        agent = AMPAgent(env, yaml.safe_load(open(DEFAULT_CFG)))
        if policy_ckpt:
            agent.restore(policy_ckpt)
        n = 0
        while n < steps:
            info = agent.train_epoch()
            n += agent.horizon_length
            say(f"epoch {agent.epoch_num}: fps_step {info['fps_step']:,.0f} fps_total {info['fps_total']:,.0f} "
                  f"a_loss {info['actor_loss']:.4f} c_loss {info['critic_loss']:.4f} disc_loss {info['disc_loss']:.4f} kl {info['kl']:.5f}")
        return
    from .learning.locoval_rollout import LocoValRollout
    kw = {}
    if use_policy:
        from .learning.amp_policy import AMPPolicyBundle
        bundle = AMPPolicyBundle(env.env.task, checkpoint=policy_ckpt)
        kw = dict(policy=bundle.policy, disc_reward=bundle.disc_reward,
                  inversion_penalty_scale=float(bundle.config.get("inversion_penalty_scale", 0.3)))
    agent = LocoValRollout(env, use_pose=args.input_init_pose, use_vel=args.input_init_vel, **kw)
    t0 = time.time()
    n = 0
    while n < steps:
        agent.play_steps()
        n += agent.horizon_length
    torch.cuda.synchronize()
    dt = time.time() - t0
    say(f"fps_step: {env.env.num_envs * world * n / dt:,.0f} env-steps/s ({n} steps of {env.env.num_envs} envs x {world} ranks), "
        f"LocoVal loss {agent.vnet_loss:.4f}, {agent.fitted_episodes} episodes fitted")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
