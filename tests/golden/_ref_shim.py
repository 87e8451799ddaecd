"""Import shim that lets the reference's pure-torch task code import in THIS container.

Used only by tests/golden/gen_golden.py (fixture generation). It is never imported by the
product, by the tests, by smoke() or by bench.py: /root/reference does not exist on the GPU box.

The physics engine bindings (isaacgym._bindings) are absent from the reference, so `gymapi`,
`gymtorch`, `gymutil`, `rlgpu` are mocked; `isaacgym/torch_utils.py` and `terrain_utils.py` are
pure python and are loaded by path.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REF = os.environ.get("EMLOCO_REFERENCE", "/root/reference")

_MOCK_ROOTS = [
    "gym", "imageio", "aiohttp", "cv2", "lxml", "smplx", "stl", "vtk", "mujoco", "mujoco_py",
    "skimage", "termcolor", "wandb", "rl_games", "pyvirtualdisplay", "open3d", "chumpy", "trimesh",
    "progress", "tensorboard", "tensorboardX", "optuna", "easydict",
]


class _MockFinder:
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MOCK_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        m = MagicMock()
        m.__path__ = []
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def install_pacer():
    """Make `env.tasks.*`, `env.util.*`, `utils.*`, `learning.*` of the reference importable."""
    np.float = float
    np.int = int
    np.Inf = np.inf
    pkg = types.ModuleType("isaacgym")
    pkg.__path__ = []
    sys.modules["isaacgym"] = pkg
    for name in ["gymapi", "gymtorch", "gymutil", "rlgpu"]:
        m = MagicMock()
        sys.modules[f"isaacgym.{name}"] = m
        setattr(pkg, name, m)
    for name in ["torch_utils", "terrain_utils"]:
        spec = importlib.util.spec_from_file_location(
            f"isaacgym.{name}", f"{REF}/isaacgym/python/isaacgym/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"isaacgym.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    sys.meta_path.insert(0, _MockFinder())
    sys.path[:0] = [f"{REF}/pacer/pacer", f"{REF}/pacer"]
    os.chdir(f"{REF}/pacer")


def install_predictor():
    """Make the reference's social-transmotion modules importable (model_jta, dataset_jta, utils.metrics)."""
    np.float = float
    np.int = int
    sys.meta_path.insert(0, _MockFinder())
    # order matters: both trees have a top-level `utils`
    sys.path.insert(0, f"{REF}/social-transmotion")
    sys.path.append(f"{REF}/pacer/pacer")
    os.chdir(f"{REF}/social-transmotion")
