"""GPU: a stitched mosaic at the REAL tile geometry (tile 4096 / overlap 1024) against the oracle.

The full 40000^2 jobs are beyond the oracle (tests/test_gpu_fullsize.py checks them through
properties and one window each); here a window of the same synthetic rasters that still holds
every kind of tile the full job has -- interior tiles, first row / column, grown edge tiles
(tiling.py:417-431) -- goes through `doTiledShepherdSegmentation` on the device and through the
oracle (one tile per host thread, then `oracle.stitch_tiles` = tiling.py:950-1203), with the
reference's own k-means model of the full raster (tests/golden/c3_fit_reference.npz,
c4_fit_reference.npz).  Compared bit for bit: the mosaic, maxSegId, the histogram.

  C3: top-left 14336^2 of synthimg(11, 6, ...): 4 x 4 tiles, the last row / column grown to 5120.
  C4: top-left 23552 x 14336 of synthimg(13, 10, ...): 7 x 4 tiles (last row / column 5120); it holds tile (5, 2)
      of the full job with both its neighbours, whose 10 716-pixel segment the reference's stitch
      recodes to 0 (the mode of the neighbour strip is 0: tests/diag_c4_zeros.py,
      tests/golden/stitch_quirk_zeros) -- labelled pixels that end up unlabelled, a stitch outcome
      that only shows at this scale.
"""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def oracle_mosaic(oracle, window_of, nRows, nCols, centres, msd, tile=4096, overlap=1024, minseg=50,
                  nthreads=None):
    """The oracle's tiled run on a raster given by window_of(x, y, xs, ys) -> (nBands, ys, xs)."""
    tiles, ntc, ntr = oracle.get_tiles(nRows, nCols, tile, overlap)
    if nthreads is None:
        try:
            nthreads = len(os.sched_getaffinity(0))
        except AttributeError:
            nthreads = os.cpu_count() or 1
    nthreads = max(1, min(nthreads, len(tiles)))

    def one(key):
        (x, y, xs, ys) = tiles[key]
        return key, oracle.segment_tile(window_of(x, y, xs, ys), centres, minseg, msd, None, True)['segimg']

    order = sorted(tiles, key=lambda k: -(tiles[k][2] * tiles[k][3]))
    with ThreadPoolExecutor(nthreads) as ex:
        local = dict(ex.map(one, order))
    return oracle.stitch_tiles(local, tiles, ntc, ntr, nRows, nCols, overlap) + (tiles,)


def _download(ptr, nRows, nCols):
    from pyshepseg_amd import _lib
    c = _lib.ctx()
    out = np.empty((nRows, nCols), dtype=np.uint32)
    c.check(c._L.shp_dev_download(c.handle, _lib.ptr(out), ctypes.c_void_p(ptr), out.nbytes))
    return out


def _run(oracle, seed, nb, nRows, nCols, fixture):
    from pyshepseg_amd import tiling, shepseg
    import test_gpu_fullsize as T
    ref = np.load(os.path.join(GOLDEN, fixture))
    km = shepseg.KMeansModel(np.ascontiguousarray(ref['centres'], dtype=np.float64))
    ras = tiling.DeviceRaster.synth(seed, nb, nRows, nCols)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=16)
    try:
        r = tiling.doTiledShepherdSegmentation(
            ras, tiling._KEEP_ON_DEVICE, tileSize=4096, overlapSize=1024, minSegmentSize=50,
            numClusters=60, kmeansObj=km, concurrencyCfg=cfg)
        got = _download(r.outDev[0], nRows, nCols)
        hist = np.asarray(r.hist).astype(np.int64)
        tiling.freeDeviceOutput(r)
        msd = float(r.maxSpectralDiff)
        assert msd == float(shepseg.autoMaxSpectralDiff(km, 'auto', 50))
        (want, wantMax, wantHist, tiles) = oracle_mosaic(
            oracle, lambda x, y, xs, ys: T._device_window(ras, x, y, xs, ys), nRows, nCols,
            km.cluster_centers_, msd)
    finally:
        ras.free()
    assert int(r.maxSegId) == wantMax
    assert np.array_equal(hist, wantHist.astype(np.int64))
    assert np.array_equal(got, want)
    return got, tiles, r


def test_c3_window_mosaic_vs_oracle(oracle):
    """16 tiles of the C3 raster incl. grown edge tiles (5120 wide / high), the reference's model."""
    (got, tiles, r) = _run(oracle, 11, 6, 14336, 14336, 'c3_fit_reference.npz')
    assert len(tiles) == 16 and r.numTileRows == 4 and r.numTileCols == 4
    assert tiles[(3, 3)] == (9216, 9216, 5120, 5120)
    assert (got != 0).all() or (got == 0).sum() < 64


def test_c4_window_mosaic_vs_oracle(oracle):
    """28 ten-band tiles of the C4 raster around the full job's tile (5, 2): the window keeps that
    tile and its upper / left neighbours exactly as the full job has them (same pixels, same
    trimmed windows), so the segment the reference's stitch recodes to 0 is in the comparison."""
    (got, tiles, r) = _run(oracle, 13, 10, 14336, 23552, 'c4_fit_reference.npz')
    assert len(tiles) == 28 and r.numTileRows == 4 and r.numTileCols == 7
    assert tiles[(5, 2)] == (15360, 6144, 4096, 4096)
    # the quirk itself: labelled pixels recoded to 0 inside tile (5, 2)'s trimmed window
    (x, y, xs, ys) = tiles[(5, 2)]
    zeros = int((got[y + 512:y + ys - 512, x + 512:x + xs - 512] == 0).sum())
    assert zeros == 3431
