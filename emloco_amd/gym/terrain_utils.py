"""terrain_utils: height-field sub-terrain generators and the height-field -> triangle-mesh conversion
(reference: isaacgym/python/isaacgym/terrain_utils.py:17-364; the poles generator of
pacer/pacer/env/tasks/humanoid_pedestrain_terrain.py:937-993 lives in the task module).

A `SubTerrain` carries an int16 field `height_field_raw[width, length]` (first axis = x) in units of `vertical_scale`
metres on a `horizontal_scale`-metre grid.  Every generator edits the field in place and returns the terrain.  The
generators that draw random numbers consume the legacy global `np.random` stream call for call like the reference
(`np.random.choice(seq)` = one `randint(0, len(seq))` draw), so a seeded run reproduces the reference's maps
(tests/golden/terrain_generators.npz).

`random_uniform_terrain`: the reference up-samples its coarse noise grid with `scipy.interpolate.interp2d(kind='linear')`,
which SciPy removed in 1.14 (this image ships 1.15) -- so that one function cannot be run here and is PARITY UNPINNED;
it is restated with the separable piecewise-linear interpolation `interp2d(linear)` performs on a regular grid.
"""
import numpy as np


class SubTerrain:                                                   # terrain_utils.py:353-364
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _centre_box(terrain, size_px):
    """index bounds of the size_px x size_px platform around the centre of the field"""
    return ((terrain.width - size_px) // 2, (terrain.width + size_px) // 2,
            (terrain.length - size_px) // 2, (terrain.length + size_px) // 2)


def _lerp_axis(coarse, n_out, axis):
    """piecewise-linear resampling of `coarse` along `axis` from linspace(0, L, n_in) knots to linspace(0, L, n_out)"""
    n_in = coarse.shape[axis]
    t = np.linspace(0.0, n_in - 1.0, n_out)
    i0 = np.minimum(t.astype(np.int64), max(n_in - 2, 0))
    w = t - i0
    a = np.take(coarse, i0, axis=axis)
    b = np.take(coarse, np.minimum(i0 + 1, n_in - 1), axis=axis)
    shape = [1, 1]
    shape[axis] = n_out
    w = w.reshape(shape)
    return a * (1.0 - w) + b * w


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):   # :17-51
    """Uniform noise: heights drawn from arange(min, max + step, step) [vertical units] on a grid of
    `downsampled_scale` metres, bilinearly up-sampled to the field and ADDED to it."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale)
    q = int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + q, q)
    n_w = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    n_l = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(levels, (n_w, n_l)).astype(np.float64)
    fine = _lerp_axis(_lerp_axis(coarse, terrain.width, 0), terrain.length, 1)
    terrain.height_field_raw += np.rint(fine).astype(np.int16)
    return terrain


def sloped_terrain(terrain, slope=1):                               # :54-71
    """Constant slope along x: row i gains trunc(max_height * i / width)."""
    top = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    ramp = (top * np.arange(terrain.width) / terrain.width).astype(terrain.height_field_raw.dtype)
    terrain.height_field_raw += ramp[:, None]
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.):      # :74-106
    """Pyramid: product of two tent functions scaled to max_height, clipped at the height of the platform corner."""
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    tent_x = (cx - np.abs(cx - np.arange(terrain.width))) / cx
    tent_y = (cy - np.abs(cy - np.arange(terrain.length))) / cy
    top = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (top * tent_x[:, None] * tent_y[None, :]).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    corner = terrain.height_field_raw[terrain.width // 2 - half, terrain.length // 2 - half]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(corner, 0), max(corner, 0))
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.):   # :109-146
    """`num_rects` axis-aligned boxes of height in {-H, -H//2, H//2, H}, sizes and positions on a 4-pixel lattice;
    flat platform in the centre.  Five draws per box: width, length, start_i, start_j, height."""
    H = int(max_height / terrain.vertical_scale)
    s_lo, s_hi = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    n_i, n_j = terrain.height_field_raw.shape
    heights = [-H, -H // 2, H // 2, H]
    sizes = range(s_lo, s_hi, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes)
        l = np.random.choice(sizes)
        i0 = np.random.choice(range(0, n_i - w, 4))
        j0 = np.random.choice(range(0, n_j - l, 4))
        terrain.height_field_raw[i0:i0 + w, j0:j0 + l] = np.random.choice(heights)
    x1, x2, y1, y2 = _centre_box(terrain, plat)
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def wave_terrain(terrain, num_waves=1, amplitude=1.):               # :149-169
    """A (cos along y + sin along x) with `num_waves` periods across the length, amplitude/2 each."""
    A = int(0.5 * amplitude / terrain.vertical_scale)
    if num_waves > 0:
        div = terrain.length / (num_waves * np.pi * 2)
        sx = A * np.sin(np.arange(terrain.width) / div)
        cy = A * np.cos(np.arange(terrain.length) / div)
        terrain.height_field_raw += (cy[None, :] + sx[:, None]).astype(terrain.height_field_raw.dtype)
    return terrain


def stairs_terrain(terrain, step_width, step_height):               # :172-192
    """Stairs along x: the k-th run of `step_width` rows is raised by (k + 1) step heights."""
    sw, sh = int(step_width / terrain.horizontal_scale), int(step_height / terrain.vertical_scale)
    n = terrain.width // sw
    rise = ((np.arange(n * sw) // sw + 1) * sh).astype(terrain.height_field_raw.dtype)
    terrain.height_field_raw[:n * sw, :] += rise[:, None]
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.):   # :195-224
    """Concentric square steps towards the centre until the remaining square is no larger than the platform."""
    sw, sh = int(step_width / terrain.horizontal_scale), int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    x0, x1, y0, y1, level = 0, terrain.width, 0, terrain.length, 0
    while x1 - x0 > plat and y1 - y0 > plat:
        x0, x1, y0, y1, level = x0 + sw, x1 - sw, y0 + sw, y1 - sw, level + sh
        terrain.height_field_raw[x0:x1, y0:y1] = level
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1., depth=-10):   # :227-283
    """Square stones of random height separated by holes of `depth`; rows of stones are laid along the longer axis,
    each row starting at a random offset (one randint per row, one height draw per stone incl. the partial first)."""
    size = int(stone_size / terrain.horizontal_scale)
    gap = int(stone_distance / terrain.horizontal_scale)
    H = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    levels = np.arange(-H - 1, H, step=1)
    f = terrain.height_field_raw
    f[:, :] = int(depth / terrain.vertical_scale)
    along_y = terrain.length >= terrain.width                       # rows advance along y, stones are laid along x
    n_run, n_lay = (terrain.length, terrain.width) if along_y else (terrain.width, terrain.length)
    view = f if along_y else f.T                                    # view[lay, run]
    run = 0
    while run < n_run:
        run_end = min(n_run, run + size)
        lay = np.random.randint(0, size)
        view[0:max(0, lay - gap), run:run_end] = np.random.choice(levels)
        while lay < n_lay:
            view[lay:min(n_lay, lay + size), run:run_end] = np.random.choice(levels)
            lay += size + gap
        run += size + gap
    x1, x2, y1, y2 = _centre_box(terrain, plat)
    f[x1:x2, y1:y2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):   # :286-350
    """Regular grid: vertex (i, j) at (i*hs, j*hs, h[i,j]*vs); two triangles per cell, (v00, v11, v01) and
    (v00, v10, v11).  With `slope_threshold`, a vertex at the foot of a rise steeper than the threshold is moved one
    cell towards the rise (and the vertex at the top of a drop one cell back), which turns steep faces vertical."""
    hf = np.asarray(height_field_raw)
    rows, cols = hf.shape
    y = np.linspace(0, (cols - 1) * horizontal_scale, cols)
    x = np.linspace(0, (rows - 1) * horizontal_scale, rows)
    yy, xx = np.meshgrid(y, x)
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        mx, my, mc = (np.zeros((rows, cols)) for _ in range(3))
        dx = hf[1:, :] - hf[:-1, :]                                 # int16 arithmetic, as the reference (wraps beyond +-32767)
        dy = hf[:, 1:] - hf[:, :-1]
        dc = hf[1:, 1:] - hf[:-1, :-1]
        mx[:-1, :] += dx > thr
        mx[1:, :] -= -dx > thr
        my[:, :-1] += dy > thr
        my[:, 1:] -= -dy > thr
        mc[:-1, :-1] += dc > thr
        mc[1:, 1:] -= -dc > thr
        xx += (mx + mc * (mx == 0)) * horizontal_scale
        yy += (my + mc * (my == 0)) * horizontal_scale
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


def mesh_vertex_moves(vertices, shape, horizontal_scale, origin=(0.0, 0.0)):
    """The slope correction of `convert_heightfield_to_trimesh` read back from a mesh: how many cells (-1, 0, +1) vertex (i, j)
    sits away from its grid position (i, j) * horizontal_scale + origin along x and y.  Returns (move_x, move_y) int8 [rows][cols];
    raises when the mesh is not a height-field mesh whose vertices moved by whole cells (`add_triangle_mesh` hands these to the
    simulator, which collides with the corrected mesh: `emloco_sim_set_ground_mesh_moves`)."""
    rows, cols = shape
    v = np.asarray(vertices, dtype=np.float64).reshape(rows, cols, 3)
    gx, gy = np.meshgrid(np.arange(rows) * float(horizontal_scale), np.arange(cols) * float(horizontal_scale), indexing="ij")
    fx = (v[:, :, 0] - origin[0] - gx) / float(horizontal_scale)
    fy = (v[:, :, 1] - origin[1] - gy) / float(horizontal_scale)
    mx, my = np.rint(fx), np.rint(fy)
    if np.abs(fx - mx).max() > 1e-3 or np.abs(fy - my).max() > 1e-3 or np.abs(mx).max() > 1 or np.abs(my).max() > 1:
        raise ValueError("not a height-field mesh with whole-cell vertex moves")
    return mx.astype(np.int8), my.astype(np.int8)
