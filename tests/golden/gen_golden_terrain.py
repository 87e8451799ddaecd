#!/usr/bin/env python3
"""Golden vectors of the height-field terrain generators, made by running the REFERENCE's own code here.

    python tests/golden/gen_golden_terrain.py     # writes tests/golden/terrain_generators.npz, terrain_layout.npz

Runs isaacgym/python/isaacgym/terrain_utils.py (generators + height-field -> trimesh) and the `Terrain` class of
pacer/pacer/env/tasks/humanoid_pedestrain_terrain.py with seeded `np.random`, through tests/golden/_ref_shim.py.
Only parameters, seeds and produced arrays are stored.  Not covered (cannot run in this image): `random_uniform_terrain`
(needs scipy.interpolate.interp2d, removed in SciPy 1.14) and `poles_terrain` (needs scikit-image).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402


def main():
    _ref_shim.install_pacer()
    import isaacgym.terrain_utils as RT
    import env.tasks.humanoid_pedestrain_terrain as HPT

    out = {}

    def sub(w, l):
        return RT.SubTerrain("terrain", width=w, length=l, vertical_scale=0.005, horizontal_scale=0.1)

    cases = [
        ("sloped", (64, 48), 3, lambda t: RT.sloped_terrain(t, slope=0.35)),
        ("sloped_neg", (40, 40), 3, lambda t: RT.sloped_terrain(t, slope=-0.2)),
        ("pyramid_sloped", (80, 80), 4, lambda t: RT.pyramid_sloped_terrain(t, slope=0.42, platform_size=3.)),
        ("pyramid_sloped_neg", (64, 64), 4, lambda t: RT.pyramid_sloped_terrain(t, slope=-0.3, platform_size=2.)),
        ("discrete_obstacles", (80, 80), 5, lambda t: RT.discrete_obstacles_terrain(t, 0.1, 1., 2., 40, platform_size=3.)),
        ("wave", (64, 96), 6, lambda t: RT.wave_terrain(t, num_waves=2, amplitude=0.5)),
        ("stairs", (70, 50), 7, lambda t: RT.stairs_terrain(t, step_width=0.31, step_height=0.12)),
        ("pyramid_stairs_up", (80, 80), 8, lambda t: RT.pyramid_stairs_terrain(t, step_width=0.31, step_height=0.15, platform_size=3.)),
        ("pyramid_stairs_down", (80, 80), 8, lambda t: RT.pyramid_stairs_terrain(t, step_width=0.31, step_height=-0.2, platform_size=3.)),
        ("stepping_stones_sq", (80, 80), 9, lambda t: RT.stepping_stones_terrain(t, stone_size=0.9, stone_distance=0.1, max_height=0., platform_size=3.)),
        ("stepping_stones_long", (48, 80), 10, lambda t: RT.stepping_stones_terrain(t, stone_size=0.6, stone_distance=0.2, max_height=0.05, platform_size=1.)),
        ("stepping_stones_wide", (80, 48), 11, lambda t: RT.stepping_stones_terrain(t, stone_size=0.5, stone_distance=0.1, max_height=0.1, platform_size=1., depth=-2)),
    ]
    for name, (w, l), seed, fn in cases:
        np.random.seed(seed)
        t = sub(w, l)
        fn(t)
        out[name] = t.height_field_raw.astype(np.int16)
        out[name + "_seed"] = np.array(seed)
    # height field -> triangle mesh, with and without the slope correction
    np.random.seed(12)
    hf = (np.random.randint(-3, 4, size=(14, 11)) * 40).astype(np.int16)
    for tag, thr in (("plain", None), ("thr", 0.9)):
        v, tr = RT.convert_heightfield_to_trimesh(hf, 0.1, 0.005, thr)
        out["trimesh_" + tag + "_vertices"] = v
        out["trimesh_" + tag + "_triangles"] = tr.astype(np.uint32)
    out["trimesh_field"] = hf
    np.savez_compressed(os.path.join(HERE, "terrain_generators.npz"), **out)
    print("wrote terrain_generators", {k: v.shape for k, v in out.items()})

    # ------------------------------------------------------------------ Terrain layout (curriculum and randomised)
    lay = {}
    props = [0.25, 0.0, 0.2, 0.2, 0.15, 0.1, 0.0, 0.1]          # no slope+noise (interp2d) and no poles (scikit-image) cells
    for tag, cur, seed in (("curriculum", True, 21), ("random", False, 22)):
        cfg = dict(terrainType="trimesh", mapLength=8., mapWidth=8., terrainProportions=props, numLevels=3, numTerrains=6,
                   curriculum=cur, slopeTreshold=0.9)
        np.random.seed(seed)
        t = HPT.Terrain(cfg, 64, "cpu")
        b = t.border
        lay[tag + "_interior"] = t.height_field_raw[b:-b, b:-b].astype(np.int16)
        lay[tag + "_field_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(t.height_field_raw).tobytes()).hexdigest())
        lay[tag + "_env_origins"] = t.env_origins
        lay[tag + "_num_samples"] = np.array(t.num_samples)
        lay[tag + "_coord_sums"] = np.array([float(t.coord_x_scale.double().sum()), float(t.coord_y_scale.double().sum())])
        lay[tag + "_vertices_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(t.vertices).tobytes()).hexdigest())
        lay[tag + "_triangles_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(t.triangles.astype(np.uint32)).tobytes()).hexdigest())
        lay[tag + "_seed"] = np.array(seed)
        np.random.seed(seed + 100)
        import torch
        lay[tag + "_valid_locs"] = t.sample_valid_locations(64, torch.arange(16)).numpy()
    lay["proportions"] = np.array(props)
    np.savez_compressed(os.path.join(HERE, "terrain_layout.npz"), **lay)
    print("wrote terrain_layout", {k: v.shape for k, v in lay.items()})


if __name__ == "__main__":
    main()
