"""Deterministic weights for the full-width predictor fixtures: a formula both the fixture generator (which loads them into
the REFERENCE model) and the GPU test (which loads them into this repo's model) evaluate, so the ~3 MB of parameters of the
d = 128 / ff = 1024 model are not stored.  Test infrastructure."""
import zlib

import numpy as np


def make_state_dict(shapes, seed):
    """shapes: {state_dict key: shape}.  Returns {key: float32 array}: matrices ~ N(0, 1 / fan_in), biases ~ N(0, 0.05^2),
    LayerNorm weights 1 + N(0, 0.05^2), embedding tables ~ N(0, 0.5^2) (rows beyond norm 1 are renormalised by
    Embedding(max_norm=1) on lookup, on both sides)."""
    out = {}
    for key in sorted(shapes):
        shape = tuple(int(v) for v in shapes[key])
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        x = rng.standard_normal(shape)
        if "encoding" in key:
            x *= 0.5
        elif len(shape) >= 2:
            x /= np.sqrt(shape[-1])
        elif "norm" in key and key.endswith("weight"):
            x = 1.0 + 0.05 * x
        else:
            x *= 0.05
        out[key] = x.astype(np.float32)
    return out


def sample(a, n=4096):
    """Evenly strided sample of a tensor (fixtures keep samples of the large gradients)."""
    f = np.asarray(a).reshape(-1)
    step = max(1, f.size // n)
    return f[::step][:n].copy()
