"""draw_utils: random binary shapes for the 'poles' sub-terrain (mirror of pacer/pacer/utils/draw_utils.py:16-75).

The reference rasterises with scikit-image (`skimage.draw.disk / ellipse / polygon / bezier_curve`), which is absent from
this image, so these functions are PARITY UNPINNED: the `np.random` draws are made call for call like the reference,
the shapes are rasterised here from their published definitions (pixel centres strictly inside the conic / inside the
polygon by the even-odd rule; the rational quadratic Bezier sampled densely and rounded), which can differ from
scikit-image at boundary pixels.  Each returns an (img_size, img_size) 0/1 integer image.
"""
import numpy as np
from scipy import ndimage


def _grid(n):
    return np.arange(n)[:, None], np.arange(n)[None, :]


def draw_disk(img_size=80, max_r=10, iterations=3):                 # draw_utils.py:16-25
    x, y = np.random.uniform(max_r, img_size - max_r, size=(2))
    radius = int(np.random.uniform(max_r))
    r, c = _grid(img_size)
    return ((r - x) ** 2 + (c - y) ** 2 < radius * radius).astype(int) if radius > 0 else np.zeros((img_size, img_size), int)


def draw_ellipse(img_size=80, max_size=10):                         # draw_utils.py:64-75
    r0, c0 = np.random.uniform(max_size, img_size - max_size), np.random.uniform(max_size, img_size - max_size)
    rr, cr = np.random.uniform(1, max_size), np.random.uniform(1, max_size)
    r, c = _grid(img_size)
    return (((r - r0) / rr) ** 2 + ((c - c0) / cr) ** 2 < 1.0).astype(int)


def draw_polygon(img_size=80, max_sides=10):                        # draw_utils.py:52-61
    n = int(np.random.uniform(3, max_sides))
    pr = np.random.uniform(0, img_size, size=(n,)).astype(int)
    pc = np.random.uniform(0, img_size, size=(n,)).astype(int)
    r, c = _grid(img_size)
    inside = np.zeros((img_size, img_size), dtype=bool)
    for k in range(n):                                              # even-odd rule over the closed vertex loop
        r0, c0, r1, c1 = pr[k], pc[k], pr[(k + 1) % n], pc[(k + 1) % n]
        if r0 == r1:
            continue
        crosses = (r0 > r) != (r1 > r)
        c_at = c0 + (r - r0) * (c1 - c0) / (r1 - r0)
        inside ^= crosses & (c < c_at)
    return inside.astype(int)


def draw_curve(img_size=80, max_sides=10, iterations=3):            # draw_utils.py:40-49
    r0, c0, r1, c1, r2, c2 = np.random.uniform(0, img_size, size=(6,)).astype(int)
    w = np.random.random()
    t = np.linspace(0.0, 1.0, 4 * img_size)
    b0, b1, b2 = (1 - t) ** 2, 2 * w * t * (1 - t), t ** 2         # rational quadratic Bezier, middle weight w
    den = b0 + b1 + b2
    rr = np.clip(np.rint((b0 * r0 + b1 * r1 + b2 * r2) / den).astype(int), 0, img_size - 1)
    cc = np.clip(np.rint((b0 * c0 + b1 * c1 + b2 * c2) / den).astype(int), 0, img_size - 1)
    img = np.zeros((img_size, img_size), dtype=np.uint8)
    img[rr, cc] = 1
    return ndimage.binary_dilation(img, iterations=iterations).astype(int)
