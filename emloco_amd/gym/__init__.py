"""Boundary 1: the Isaac-Gym-shaped API the reference's task code calls, backed by libemloco_hip.so.

Module names mirror /root/reference/isaacgym/python/isaacgym/{gymapi,gymtorch,gymutil,torch_utils,terrain_utils}.py
so task code written against `from isaacgym import gymapi, gymtorch` ports by changing the import.
"""
from . import gymapi, gymtorch, gymutil, terrain_utils, torch_utils  # noqa: F401
