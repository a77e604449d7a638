"""tests/golden/tile_u32_wrap.npz, tile_i32_wrap.npz from the UNMODIFIED reference: 32-bit imagery
whose band differences reach 2^31.5 .. 2^32, where findNearestNeighbourPixel's int64 sum of squares
wraps (and a sum that wrapped negative counts as "unset" for the next candidate, shepseg.py:731).

    cd oracle/refgen && /opt/conda/bin/python3.9 gen_golden_wrap32.py
"""
import numpy as np

import refenv  # noqa: F401
from gen_golden import tile_case


def image(dt, seed):
    rng = np.random.RandomState(seed)
    info = np.iinfo(dt)
    levels = np.array([info.min, info.min + 3, (int(info.min) + int(info.max)) // 2, info.max - 2, info.max - 1],
                      dtype=np.int64)
    nb, n = 6, 48
    blocks = levels[rng.randint(0, len(levels), size=(nb, n // 6, n // 6))]
    img = np.kron(blocks, np.ones((1, 6, 6), dtype=np.int64))
    # single pixels with limit values, so that the single-pixel stage has many candidates whose
    # distances to their neighbours overflow
    m = rng.rand(n, n) < 0.12
    noise = levels[rng.randint(0, len(levels), size=(nb, n, n))]
    img = np.where(m[None], noise, img)
    img[:, 20:23, :] = info.max                       # a null stripe (every band)
    return img.astype(dt)


tile_case('tile_u32_wrap', image(np.uint32, 5), 4, 12, int(np.iinfo(np.uint32).max), True, 100)
tile_case('tile_i32_wrap', image(np.int32, 6), 4, 12, int(np.iinfo(np.int32).max), False, 100)
