"""terrain_utils: height-field terrain pieces used by the task (reference: isaacgym/python/isaacgym/terrain_utils.py).

Only what the BASELINE configs exercise: a `SubTerrain` container and the height-field -> triangle-mesh
conversion (terrain_utils.py:286-350) without slope correction.  The shaped generators (slopes, stairs,
stepping stones, poles) are listed under 'next' in DESIGN.md.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Regular grid: vertex (i, j) at (i*hs, j*hs, h[i,j]*vs); two triangles per cell."""
    hf = np.asarray(height_field_raw)
    rows, cols = hf.shape
    y = np.linspace(0, (cols - 1) * horizontal_scale, cols)
    x = np.linspace(0, (rows - 1) * horizontal_scale, rows)
    yy, xx = np.meshgrid(y, x)
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0] = xx.flatten()
    vertices[:, 1] = yy.flatten()
    vertices[:, 2] = hf.flatten() * vertical_scale
    idx = np.arange(rows * cols).reshape(rows, cols)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    triangles = np.empty((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    triangles[0::2] = np.stack([a, d, b], 1)
    triangles[1::2] = np.stack([a, c, d], 1)
    return vertices, triangles
