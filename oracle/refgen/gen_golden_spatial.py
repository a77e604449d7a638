"""Golden vectors for calcPerSegmentSpatialStatsTiled's built-in user functions from the
UNMODIFIED reference (build container only):
    /opt/conda/bin/python3.9 oracle/refgen/gen_golden_spatial.py

The reference's driver (tilingstats.py:1262-1390) needs GDAL files; accumulateSegSpatial and the
three user functions are njit code that does not.  The tile loop and the column bookkeeping below
are this harness's own restatement of the driver (tile rows outer, tile columns inner; intArr
int32 / floatArr float64 filled with missingStatsValue, stored to int64 / float32 columns as
RatPage.setRatVal does); everything that computes is the reference's.  Output: plain arrays.
"""
import os
import numpy as np

import refenv  # noqa: F401
from pyshepseg import tilingstats
from oracle import oracle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                   'tests', 'golden')
MISSING = -9999


def run_reference(seg, band, null_val, tile, userFunc, userParam, nint, nflt):
    segDict = tilingstats.createSegSpatialDataDict()
    noDataDict = tilingstats.createNoDataDict()
    nullv = tilingstats.numbaTypeForImageType(null_val)
    (nr, nc) = seg.shape
    for ty in range(0, nr, tile):
        for tx in range(0, nc, tile):
            th, tw = min(tile, nr - ty), min(tile, nc - tx)
            tilingstats.accumulateSegSpatial(
                segDict, noDataDict, nullv, np.ascontiguousarray(seg[ty:ty + th, tx:tx + tw]),
                np.ascontiguousarray(band[ty:ty + th, tx:tx + tw]), ty, tx)
    S = int(seg.max())
    ic = np.full((nint, S + 1), MISSING, dtype=np.int64)
    fc = np.full((nflt, S + 1), MISSING, dtype=np.float32)
    ic[:, 0] = 0
    fc[:, 0] = 0
    intArr = np.empty(nint, dtype=np.int32)
    floatArr = np.empty(nflt, dtype=np.float64)
    for segId in range(1, S + 1):
        if segId in segDict and len(segDict[segId]) > 0:
            intArr.fill(MISSING)
            floatArr.fill(MISSING)
            userFunc(segDict[segId], nullv, intArr, floatArr, userParam)
            ic[:, segId] = intArr
            fc[:, segId] = floatArr.astype(np.float32)
    return ic, fc


def main():
    img = oracle.synthimg(23, 2, 140, 120)
    centres, _l, _n = oracle.kmeans_fit(img.reshape(2, -1).T[::5].astype(np.float64),
                                        np.linspace(1500, 4500, 6)[:, None] * np.ones((1, 2)))
    seg = oracle.segment_tile(img, centres, 10, 400.0, None, True)['segimg']
    seg[70:73, 20:90] = 0                       # some null-segment pixels
    band = img[1].copy()
    rng = np.random.RandomState(9)
    band[rng.rand(*band.shape) < 0.03] = 0      # scattered nodata inside segments
    sid = int(seg[10, 10])
    band[seg == sid] = 0                        # one segment entirely nodata
    transform = np.array([500000.0, 30.0, 0.0, 6500000.0, 0.0, -30.0])
    rot = np.array([1000.5, 10.0, 0.25, -2000.25, -0.5, -10.0])
    out = dict(seg=seg, band=band, null_val=np.int64(0), tile=np.int64(48), transform=transform,
               rot=rot, stack=np.array(refenv.STACK))
    ic, fc = run_reference(seg, band, 0, 48, tilingstats.userFuncMeanCoord, transform, 0, 2)
    out['mean_fc'] = fc
    ic, fc = run_reference(seg, band, 0, 48, tilingstats.userFuncMeanCoord, rot, 0, 2)
    out['meanrot_fc'] = fc
    ic, fc = run_reference(seg, band, 0, 48, tilingstats.userFuncNumEdgePixels, True, 1, 0)
    out['edge4_ic'] = ic
    ic, fc = run_reference(seg, band, 0, 48, tilingstats.userFuncNumEdgePixels, False, 1, 0)
    out['edge8_ic'] = ic
    ic, fc = run_reference(seg, band, 0, 48, tilingstats.userFuncVariogram, 4, 0, 4)
    out['vario_fc'] = fc
    np.savez_compressed(os.path.join(OUT, 'spatial_stats.npz'), **out)
    print('spatial_stats.npz', os.path.getsize(os.path.join(OUT, 'spatial_stats.npz')))


if __name__ == '__main__':
    main()
