"""Golden vectors for subset.subsetImage's recode from the UNMODIFIED reference (build container
only):   /opt/conda/bin/python3.9 oracle/refgen/gen_golden_subset.py

The reference's driver (subset.py:124-166) needs GDAL files; its per-tile njit kernel
processSubsetTile (subset.py:366-425) does not.  The tile loop below is this harness's own
restatement of the driver's visiting order (tile rows outer, tile columns inner), the recode
itself is the reference's function.  Output: plain arrays only.
"""
import os
import numpy as np

import refenv  # noqa: F401
from numba.typed import Dict
from pyshepseg import subset, tiling
from oracle import oracle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                   'tests', 'golden')


def run_reference(seg, tlx, tly, xs, ys, mask, tile):
    recode = Dict.empty(key_type=tiling.segIdNumbaType, value_type=tiling.segIdNumbaType)
    hist = Dict.empty(key_type=tiling.segIdNumbaType, value_type=tiling.segIdNumbaType)
    out = np.zeros((ys, xs), dtype=np.uint32)
    for ty in range(0, ys, tile):
        for tx in range(0, xs, tile):
            th, tw = min(tile, ys - ty), min(tile, xs - tx)
            inData = np.ascontiguousarray(seg[tly + ty:tly + ty + th, tlx + tx:tlx + tx + tw])
            m = None if mask is None else np.ascontiguousarray(mask[ty:ty + th, tx:tx + tw])
            out[ty:ty + th, tx:tx + tw] = subset.processSubsetTile(inData, recode, hist, m)
    n = len(recode)
    orig = np.zeros(n + 1, dtype=np.uint32)
    for k, v in recode.items():
        orig[v] = k
    h = np.zeros(n + 1, dtype=np.uint32)
    for k, v in hist.items():
        h[k] = v
    return out, orig, h


def main():
    img = oracle.synthimg(21, 3, 150, 170)
    centres, _l, _n = oracle.kmeans_fit(img.reshape(3, -1).T[::7].astype(np.float64),
                                        np.linspace(1500, 4500, 8)[:, None] * np.ones((1, 3)))
    seg = oracle.segment_tile(img, centres, 8, 500.0, None, True)['segimg']
    seg[40:44, :] = 0                                       # a null band through the raster
    rng = np.random.RandomState(5)
    mask = (rng.rand(90, 100) > 0.3).astype(np.uint8)
    mask[10:30, 20:60] = 0
    cases = {}
    for name, (tlx, tly, xs, ys, m, tile) in {
            'a': (13, 21, 100, 90, None, 32), 'b': (13, 21, 100, 90, mask, 32),
            'c': (0, 0, 170, 150, None, 1024), 'd': (50, 30, 100, 90, mask, 64)}.items():
        out, orig, h = run_reference(seg, tlx, tly, xs, ys, m, tile)
        cases.update({name + '_out': out, name + '_orig': orig, name + '_hist': h,
                      name + '_win': np.array([tlx, tly, xs, ys, tile], dtype=np.int64)})
    np.savez_compressed(os.path.join(OUT, 'subset_recode.npz'), seg=seg, mask=mask,
                        stack=np.array(refenv.STACK), **cases)
    print('subset_recode.npz', os.path.getsize(os.path.join(OUT, 'subset_recode.npz')))


if __name__ == '__main__':
    main()
