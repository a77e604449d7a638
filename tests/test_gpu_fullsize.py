"""GPU, BASELINE.json's full sizes.

C2 (8192 x 8192 x 6, one call): the oracle still finishes in seconds, so labels are compared
bit for bit.  C3 (40000 x 40000 x 6 tiled): the oracle cannot run the whole job, so the run is
checked through size-independent properties -- the histogram accounts for every pixel, ids are
1..maxSegId, two runs agree exactly, the sharded driver (world size 1) agrees with the in-process
driver, per-segment pixel counts from the statistics kernel equal the histogram -- and one of its
interior 4096 x 4096 tile windows is compared with the oracle bit for bit.  C4 (the same with 10
bands) and C5 (per-segment statistics over 1.6 Gpx / 50 M segments) go through the same kind of
properties plus one oracle-checked window each."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _segment_window(ras, x, y, xs, ys, centres, minseg, msd, four=True):
    """Local labels of one window of a DeviceRaster (the worker's call, tiling.py)."""
    from pyshepseg_amd import _lib
    c = _lib.ctx()
    n = xs * ys
    d = ctypes.c_void_p()
    c.check(c._L.shp_dev_alloc(c.handle, n * 4, ctypes.byref(d)))
    try:
        mx, s1, s2, ncl = ctypes.c_uint32(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_uint32(0)
        c.check(c._L.shp_segment_window_dev(
            c.handle, ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[ras.dtype], ras.shape[0], ras.shape[1],
            ras.shape[2], x, y, xs, ys, _lib.ptr(centres), centres.shape[0], 0, 0, int(four), minseg,
            float(msd), d, ctypes.byref(mx), ctypes.byref(s1), ctypes.byref(s2), ctypes.byref(ncl), None))
        out = np.empty((ys, xs), dtype=np.uint32)
        c.check(c._L.shp_dev_download(c.handle, _lib.ptr(out), d, n * 4))
    finally:
        c.check(c._L.shp_dev_free(c.handle, d))
    return out, mx.value, s1.value, s2.value, ncl.value


def test_c2_single_call_vs_oracle(oracle):
    """BASELINE configs[1]: synthimg(5, 6, 8192, 8192), k = 60, minSeg = 50, fixed init, 1 %
    sample; the model is fitted on the device, labels are compared for that model."""
    from pyshepseg_amd import shepseg, tiling
    img = oracle.synthimg(5, 6, 8192, 8192)
    km = shepseg.fitSpectralClusters(img, 60, 1, None, True)
    centres = np.ascontiguousarray(km.cluster_centers_, dtype=np.float64)
    msd = float(shepseg.autoMaxSpectralDiff(km, 'auto', 50))
    got = shepseg.doShepherdSegmentation(img, numClusters=60, minSegmentSize=50, kmeansObj=km)
    want = oracle.segment_tile(img, centres, 50, msd, None, True)
    assert np.array_equal(got.segimg, want['segimg'])
    assert got.singlePixelsEliminated == want['singlePixelsEliminated']
    assert got.smallSegmentsEliminated == want['smallSegmentsEliminated']
    # the same image resident on the device, one window = the whole raster
    ras = tiling.DeviceRaster.fromArray(img)
    try:
        seg, mx, _s1, _s2, _ncl = _segment_window(ras, 0, 0, 8192, 8192, centres, 50, msd)
    finally:
        ras.free()
    assert mx == int(want['segimg'].max())
    assert np.array_equal(seg, want['segimg'])


def test_c3_fullsize_properties(oracle):
    """BASELINE configs[2]: 40000 x 40000 x 6, tile 4096 / overlap 1024, k = 60, minSeg = 50."""
    from pyshepseg_amd import tiling, tilingstats, _lib
    N = 40000
    ras = tiling.DeviceRaster.synth(11, 6, N, N)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=16)
    try:
        runs = []
        for _rep in range(2):
            r = tiling.doTiledShepherdSegmentation(
                ras, tiling._KEEP_ON_DEVICE, tileSize=4096, overlapSize=1024, minSegmentSize=50,
                numClusters=60, fixedKMeansInit=True, concurrencyCfg=cfg)
            hist = np.asarray(r.hist).astype(np.int64)
            runs.append((int(r.maxSegId), hist, r.kmeans.cluster_centers_.copy(),
                         float(r.maxSpectralDiff)))
            if _rep == 0:
                # per-segment pixel counts by the statistics kernel == the stitch's histogram
                c = _lib.ctx()
                sel = [('n', 'pixcount'), ('lo', 'min'), ('hi', 'max')]
                (fast, nInt, nFloat) = tilingstats.makeFastStatsSelection([0, 1, 2], sel)
                ic = np.zeros((nInt, r.maxSegId + 1), dtype=np.int64)
                fc = np.zeros((max(nFloat, 1), r.maxSegId + 1), dtype=np.float32)
                c.check(c._L.shp_segstats_dev(
                    c.handle, ctypes.c_void_p(r.outDev[0]), ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[ras.dtype],
                    N * N, r.maxSegId, 0, 0, _lib.ptr(fast), 3, -9999, _lib.ptr(ic), _lib.ptr(fc)))
                assert np.array_equal(ic[fast[0, 3], 1:], hist[1:])
                assert (ic[fast[1, 3], 1:] <= ic[fast[2, 3], 1:]).all()
            tiling.freeDeviceOutput(r)
        (mx, hist, centres, msd) = runs[0]
        assert mx == len(hist) - 1
        # no nulls, so every pixel is labelled -- but for the reference's own stitch quirk (a segment
        # matched across a midline takes the mode of the neighbour's saved strip, which is 0 where that
        # tile owned nothing: tests/golden/stitch_quirk_zeros), which may blank a handful of pixels
        assert hist[0] == 0 and N * N - 64 <= int(hist.sum()) <= N * N
        assert r.numTileRows == 12 and r.numTileCols == 12
        # the reference's stitch can leave a few ids empty; flag and histogram agree
        assert bool(r.hasEmptySegments) == bool((hist[1:] == 0).any())
        assert (hist[1:] == 0).sum() < 1e-3 * mx
        # the whole-image k-means model is the REFERENCE's, bit for bit: tests/golden/c3_fit_reference.npz
        # holds shepseg.fitSpectralClusters' centres for this raster's sub-sample (sklearn 0.24.2, Elkan's
        # algorithm, all 300 iterations; oracle/refgen/gen_golden_c3_fit.py)
        ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'c3_fit_reference.npz'))
        import zlib
        sample = tiling.readSubsampledImage(ras, list(range(1, 7)), np.sqrt(1e6 / (N * N)))
        assert zlib.crc32(np.ascontiguousarray(sample).tobytes()) == int(ref['sample_crc32'])    # the fixture's input
        assert r.kmeans.n_iter_ == int(ref['n_iter'])
        assert np.array_equal(centres.view(np.uint64), ref['centres'].view(np.uint64))
        # deterministic run to run: same model, same ids, same histogram
        assert runs[1][0] == mx and np.array_equal(runs[1][1], hist)
        assert np.array_equal(runs[1][2], centres) and runs[1][3] == msd

        # the sharded (multi-GPU) driver at world size 1 is a second implementation of the chain
        from pyshepseg_amd import distributed
        eng = distributed.HipEngine(lambda yLo, yHi: tiling.DeviceRaster.synth(11, 6, yHi - yLo, N, y0=yLo),
                                    numWorkers=16)
        # ... in the PARALLEL form of the stitch (provisional ids per tile, renumbered at the end).  Whether
        # the form is kept depends on the segmentation: with the reference's model for this raster one of
        # the 144 tiles hides its last new id outside its trimmed window (the reference then reuses the
        # id), the driver notices and redoes the chain sequentially -- either way the same result
        d = distributed.runDistributed(eng, distributed.Comm(None), N, N, 4096, 1024, minSegmentSize=50,
                                       numClusters=60, fixedKMeansInit=True, stitchMode='parallel')
        eng.ras.free()
        assert d.stitchMode in ('parallel', 'parallel->sequential')
        assert d.maxSegId == mx and np.array_equal(np.asarray(d.hist).astype(np.int64), hist)
        assert np.array_equal(d.kmeans.cluster_centers_, centres)

        # one interior tile window at full tile size against the oracle
        ti = tiling.getTilesForFile(ras, 4096, 1024)
        (x, y, xs, ys) = ti.getTile(5, 6)
        assert xs == 4096 and ys == 4096
        seg, _mx, s1, s2, ncl = _segment_window(ras, x, y, xs, ys, centres, 50, msd)
        idx_y = np.arange(y, y + ys, dtype=np.uint32)
        idx_x = np.arange(x, x + xs, dtype=np.uint32)
        sub = np.empty((6, ys, xs), dtype=np.uint16)
        c = _lib.ctx()
        c.check(c._L.shp_dev_subsample(c.handle, ctypes.c_void_p(ras.ptr), 2, 6, N, N,
                                       _lib.ptr(idx_y), ys, _lib.ptr(idx_x), xs, _lib.ptr(sub)))
        assert np.array_equal(sub, oracle.synthimg(11, 6, ys, xs, y0=y, x0=x))
        want = oracle.segment_tile(sub, centres, 50, msd, None, True)
        assert np.array_equal(seg, want['segimg'])
        assert s1 == want['singlePixelsEliminated'] and s2 == want['smallSegmentsEliminated']
    finally:
        ras.free()


def _device_window(ras, x, y, xs, ys):
    """Host copy of window (x, y, xs, ys) of every band of a DeviceRaster."""
    from pyshepseg_amd import _lib
    nb, nr, nc = ras.shape
    sub = np.empty((nb, ys, xs), dtype=ras.dtype)
    idx_y = np.arange(y, y + ys, dtype=np.uint32)
    idx_x = np.arange(x, x + xs, dtype=np.uint32)
    c = _lib.ctx()
    c.check(c._L.shp_dev_subsample(c.handle, ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[ras.dtype], nb, nr, nc,
                                   _lib.ptr(idx_y), ys, _lib.ptr(idx_x), xs, _lib.ptr(sub)))
    return sub


def test_c4_fullsize_properties(oracle):
    """BASELINE configs[3] on one GPU: synthimg(13, 10, 40000, 40000), tile 4096 / overlap 1024,
    k = 60, minSeg = 50 (the 8-GPU sharding of the same job is covered at small scale by
    test_gpu_distributed / test_distributed_cpu)."""
    from pyshepseg_amd import tiling
    N = 40000
    ras = tiling.DeviceRaster.synth(13, 10, N, N)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=16)
    try:
        r = tiling.doTiledShepherdSegmentation(
            ras, tiling._KEEP_ON_DEVICE, tileSize=4096, overlapSize=1024, minSegmentSize=50,
            numClusters=60, fixedKMeansInit=True, concurrencyCfg=cfg)
        tiling.freeDeviceOutput(r)
        hist = np.asarray(r.hist).astype(np.int64)
        mx = int(r.maxSegId)
        assert mx == len(hist) - 1 and mx > 100000
        # (see the C3 test; with the reference's model for this raster the stitch quirk blanks the 3431 pixels
        # that one 10 716-pixel segment of tile (5, 2), crossing both midlines, has in its trimmed window:
        # tests/diag_c4_zeros.py)
        assert hist[0] == 0 and N * N - 20000 <= int(hist.sum()) <= N * N
        assert r.numTileRows == 12 and r.numTileCols == 12
        assert r.kmeans.cluster_centers_.shape == (60, 10)
        # the whole-image k-means model is the REFERENCE's, bit for bit: tests/golden/c4_fit_reference.npz holds
        # shepseg.fitSpectralClusters' centres for this raster's sub-sample (sklearn 0.24.2, Elkan's algorithm,
        # all 300 iterations, one OpenMP thread; oracle/refgen/gen_golden_c3_fit.py)
        ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'c4_fit_reference.npz'))
        import zlib
        sample = tiling.readSubsampledImage(ras, list(range(1, 11)), np.sqrt(1e6 / (N * N)))
        assert zlib.crc32(np.ascontiguousarray(sample).tobytes()) == int(ref['sample_crc32'])    # the fixture's input
        assert r.kmeans.n_iter_ == int(ref['n_iter'])
        assert np.array_equal(np.ascontiguousarray(r.kmeans.cluster_centers_, dtype=np.float64).view(np.uint64),
                              ref['centres'].view(np.uint64))
        assert bool(r.hasEmptySegments) == bool((hist[1:] == 0).any())
        assert (hist[1:] == 0).sum() < 1e-3 * mx
        # one interior 10-band tile window at full tile size against the oracle
        centres = np.ascontiguousarray(r.kmeans.cluster_centers_, dtype=np.float64)
        msd = float(r.maxSpectralDiff)
        ti = tiling.getTilesForFile(ras, 4096, 1024)
        (x, y, xs, ys) = ti.getTile(7, 4)
        assert xs == 4096 and ys == 4096
        seg, _mx, s1, s2, _ncl = _segment_window(ras, x, y, xs, ys, centres, 50, msd)
        sub = _device_window(ras, x, y, xs, ys)
        assert np.array_equal(sub[:, :64, :64], oracle.synthimg(13, 10, 64, 64, y0=y, x0=x))
        want = oracle.segment_tile(sub, centres, 50, msd, None, True)
        assert np.array_equal(seg, want['segimg'])
        assert s1 == want['singlePixelsEliminated'] and s2 == want['smallSegmentsEliminated']
    finally:
        ras.free()


def test_c5_stats_fullsize(oracle):
    """BASELINE configs[4] on one GPU: 1.6 Gpx label raster of 4 x 8-pixel blocks (50 M segments),
    one uint16 band, mean / stddev / median / pixcount.  Every pixel is counted once, every
    block has its 32 pixels, and a window of 100 000 whole segments equals the oracle bit for bit."""
    from pyshepseg_amd import tiling, tilingstats, _lib
    N, BH, BW = 40000, 4, 8
    ncb = N // BW
    c = _lib.ctx()
    ras = tiling.DeviceRaster.synth(11, 1, N, N)
    d_seg = ctypes.c_void_p()
    c.check(c._L.shp_dev_alloc(c.handle, N * N * 4, ctypes.byref(d_seg)))
    try:
        S = ctypes.c_uint32(0)
        c.check(c._L.shp_dev_block_labels(c.handle, N, N, BH, BW, d_seg, ctypes.byref(S)))
        S = S.value
        assert S == (N // BH) * ncb == 50000000
        sel = [('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'), ('n', 'pixcount')]
        fast, ni, nf = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)
        ic = np.zeros((ni, S + 1), dtype=np.int64)
        fc = np.zeros((nf, S + 1), dtype=np.float32)
        c.check(c._L.shp_segstats_dev(c.handle, d_seg, ctypes.c_void_p(ras.ptr), 2, N * N, S, 0, 0,
                                      _lib.ptr(fast), len(sel), -9999, _lib.ptr(ic), _lib.ptr(fc)))
        npx = ic[fast[3, 3]]
        assert int(npx.sum()) == N * N and npx[0] == 0 and (npx[1:] == BH * BW).all()
        med = ic[fast[2, 3]]
        mean = fc[fast[0, 3]]
        assert (np.abs(mean[1:] - med[1:]) < 400).all()         # both sit inside the block's value range
        # 1600 x 2000-pixel window = 400 x 250 blocks = 100 000 segments against the oracle
        wy, wx, y0, x0 = 1600, 2000, 20000, 16000
        band = oracle.synthimg(11, 1, wy, wx, y0=y0, x0=x0)[0]
        lab = (((np.arange(y0, y0 + wy, dtype=np.uint32) // BH)[:, None] * np.uint32(ncb)) +
               (np.arange(x0, x0 + wx, dtype=np.uint32) // BW)[None, :] + np.uint32(1))
        ids, compact = np.unique(lab, return_inverse=True)
        assert len(ids) == 100000
        wi, wf = oracle.segstats((compact.reshape(lab.shape) + 1).astype(np.uint32), band, sel)
        assert np.array_equal(ic[:, ids], wi[:, 1:])
        assert np.array_equal(fc[:, ids].view(np.uint32), wf[:, 1:].view(np.uint32))
        # the same raster with its shape given: segments of 32 pixels take the patch-by-patch path (no global
        # sort at all on this raster); every column must come out the same, bit for bit
        ic2 = np.zeros_like(ic)
        fc2 = np.zeros_like(fc)
        c.check(c._L.shp_segstats2d_dev(c.handle, d_seg, ctypes.c_void_p(ras.ptr), 2, N, N, S, 0, 0,
                                        _lib.ptr(fast), len(sel), -9999, _lib.ptr(ic2), _lib.ptr(fc2)))
        assert np.array_equal(ic2, ic) and np.array_equal(fc2.view(np.uint32), fc.view(np.uint32))
    finally:
        c.check(c._L.shp_dev_free(c.handle, d_seg))
        ras.free()
