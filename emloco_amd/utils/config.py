"""Argument parsing and config loading with the reference's flag names.

Mirror of pacer/pacer/utils/config.py (get_args :177-529, load_cfg :64-138, parse_sim_params :141-174) for the
flags that reach the rollout hot path.  rl_games / wandb / viewer options are accepted and ignored.
"""
import os

import yaml

from ..gym import gymapi, gymutil

SIM_TIMESTEP = 1.0 / 60.0
_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_CFG_ENV = os.path.join(_HERE, "data", "cfg", "pacer.yaml")


def get_args(argv=None):
    custom_parameters = [
        {"name": "--test", "action": "store_true", "default": False},
        {"name": "--debug", "action": "store_true", "default": False},
        {"name": "--headless", "action": "store_true", "default": True},
        {"name": "--task", "type": str, "default": "HumanoidPedestrianTerrain"},
        {"name": "--rl_device", "type": str, "default": "cuda:0"},
        {"name": "--seed", "type": int, "default": -1},
        {"name": "--num_envs", "type": int, "default": 0},
        {"name": "--episode_length", "type": int, "default": 0},
        {"name": "--cfg_env", "type": str, "default": DEFAULT_CFG_ENV},
        {"name": "--cfg_train", "type": str, "default": ""},
        {"name": "--motion_file", "type": str, "default": "synthetic"},
        {"name": "--network_path", "type": str, "default": "output/"},
        {"name": "--exp_name", "type": str, "default": "emloco"},
        {"name": "--epoch", "type": int, "default": 0},
        {"name": "--horovod", "action": "store_true", "default": False},
        {"name": "--small_terrain", "action": "store_true", "default": False},
        {"name": "--random_heading", "action": "store_true", "default": False},
        {"name": "--init_heading", "action": "store_true", "default": False},
        {"name": "--heading_inversion", "action": "store_true", "default": False},
        {"name": "--adjust_root_vel", "action": "store_true", "default": False},
        {"name": "--input_init_pose", "action": "store_true", "default": False},
        {"name": "--input_init_vel", "action": "store_true", "default": False},
        {"name": "--add_noise", "action": "store_true", "default": False},
        {"name": "--vru", "action": "store_true", "default": False},
        {"name": "--real_path", "type": str, "default": ""},
        {"name": "--real_mesh", "action": "store_true", "default": False},
        {"name": "--valuenet_path", "type": str, "default": ""},
        {"name": "--follow", "action": "store_true", "default": False},
        {"name": "--server_mode", "action": "store_true", "default": False},
        {"name": "--pred_path", "action": "store_true", "default": False},
        {"name": "--no_virtual_display", "action": "store_true", "default": True},
        {"name": "--show_sensors", "action": "store_true", "default": False},
        {"name": "--add_proj", "action": "store_true", "default": False},
    ]
    args = gymutil.parse_arguments(description="EmLoco rollout on MI355X", custom_parameters=custom_parameters, argv=argv)
    args.device_id = args.compute_device_id
    args.device = args.sim_device_type if args.use_gpu_pipeline else "cpu"
    return args


def load_cfg(args):
    with open(args.cfg_env, "r") as f:
        cfg = yaml.load(f, Loader=yaml.SafeLoader)
    if args.num_envs > 0:
        cfg["env"]["numEnvs"] = args.num_envs            # config.py:72-73
    if args.episode_length > 0:
        cfg["env"]["episodeLength"] = args.episode_length
    cfg["name"] = args.task
    cfg["headless"] = args.headless
    cfg["env"]["motion_file"] = args.motion_file
    cfg["args"] = args
    cfg_train = {"params": {"config": {"player": {"use_pose": args.input_init_pose, "use_vel": args.input_init_vel}}},
                 "seed": args.seed}
    return cfg, cfg_train, args.network_path


def parse_sim_params(args, cfg, cfg_train=None):
    sim_params = gymapi.SimParams()
    sim_params.dt = SIM_TIMESTEP
    sim_params.num_client_threads = args.slices
    sim_params.physx.solver_type = 1
    sim_params.physx.num_position_iterations = 4
    sim_params.physx.num_velocity_iterations = 0
    sim_params.physx.num_threads = 4
    sim_params.physx.use_gpu = args.use_gpu
    sim_params.physx.num_subscenes = args.subscenes
    sim_params.use_gpu_pipeline = args.use_gpu_pipeline
    if "sim" in cfg:
        gymutil.parse_sim_config(cfg["sim"], sim_params)
    if args.num_threads > 0:
        sim_params.physx.num_threads = args.num_threads
    return sim_params
