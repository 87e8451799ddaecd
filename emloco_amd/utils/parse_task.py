"""parse_task: build the task and its VecEnv wrapper (mirror of pacer/pacer/utils/parse_task.py:29-47)."""
import numpy as np

from ..env.tasks.humanoid_pedestrain_terrain import HumanoidPedestrianTerrain
from ..env.tasks.vec_task_wrappers import VecTaskPythonWrapper

TASKS = {"HumanoidPedestrianTerrain": HumanoidPedestrianTerrain}


def parse_task(args, cfg, cfg_train, sim_params):
    if args.task not in TASKS:
        raise ValueError(f"task {args.task!r} is not on the hot path; available: {sorted(TASKS)}")
    cfg["seed"] = cfg_train.get("seed", -1)
    cfg["env"]["seed"] = cfg["seed"]
    task = TASKS[args.task](cfg=cfg, sim_params=sim_params, physics_engine=args.physics_engine,
                            device_type=args.device, device_id=args.device_id, headless=args.headless)
    env = VecTaskPythonWrapper(task, args.rl_device, cfg_train.get("clip_observations", np.inf))
    return task, env
