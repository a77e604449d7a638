"""GPU parity of the tiled driver + stitch against the reference's golden stitch vectors and
against the oracle on a larger seeded image."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
STITCH = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'stitch_*.npz')))


@pytest.mark.parametrize('name', ['e2e_ties_a', 'e2e_ties_b'])
def test_golden_tiled_end_to_end_model_included(name, golden):
    """The whole job with NO model handed in: the k-means fit of the whole raster, the tiles, the stitch,
    against goldens whose model is the reference's own fitSpectralClusters on rasters where Elkan's
    algorithm (what the reference runs) and Lloyd's end in different models
    (oracle/refgen/gen_golden_e2e_ties.py): centres and n_iter_ bit for bit, then the mosaic."""
    from pyshepseg_amd import tiling
    g = golden(name)
    null = int(g['null_val']) if int(g['has_null']) else None
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=2)
    r = tiling.doTiledShepherdSegmentation(
        g['img'], None, tileSize=int(g['tile_size']), overlapSize=int(g['overlap']),
        minSegmentSize=int(g['min_seg']), numClusters=int(g['k']), imgNullVal=null,
        fourConnected=bool(g['four']), fixedKMeansInit=True, concurrencyCfg=cfg)
    assert r.kmeans.n_iter_ == int(g['n_iter'])
    assert np.array_equal(r.kmeans.cluster_centers_.view(np.uint64), g['centres'].view(np.uint64))
    assert r.maxSpectralDiff == g['msd']
    assert r.maxSegId == int(g['max_seg_id'])
    assert np.array_equal(r.segimg, g['mosaic'])
    assert np.array_equal(r.hist, g['hist'])


@pytest.mark.parametrize('name', STITCH)
@pytest.mark.parametrize('workers', [1, 3])
def test_golden_tiled(name, workers, golden):
    from pyshepseg_amd import tiling, shepseg
    g = golden(name)
    null = int(g['null_val']) if int(g['has_null']) else None
    cfg = tiling.SegmentationConcurrencyConfig(
        concurrencyType=tiling.CONC_THREADS if workers > 1 else tiling.CONC_NONE, numWorkers=workers)
    r = tiling.doTiledShepherdSegmentation(
        g['img'], None, tileSize=int(g['tile_size']), overlapSize=int(g['overlap']),
        minSegmentSize=int(g['min_seg']), imgNullVal=null, fourConnected=bool(g['four']),
        kmeansObj=shepseg.KMeansModel(g['centres']), concurrencyCfg=cfg)
    assert (r.numTileCols, r.numTileRows) == (int(g['ntcols']), int(g['ntrows']))
    assert r.maxSpectralDiff == g['msd']
    assert r.maxSegId == int(g['max_seg_id'])
    assert np.array_equal(r.segimg, g['mosaic'])
    assert np.array_equal(r.hist, g['hist'])


def _oracle_tiled(oracle, img, centres, tile, ov, minseg, msd, null, four, simple=False):
    nr, nc = img.shape[1:]
    tiles, ntc, ntr = oracle.get_tiles(nr, nc, tile, ov)
    local = {}
    for (c, r), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        local[(c, r)] = oracle.segment_tile(sub, centres, minseg, msd, null, four)['segimg']
    return oracle.stitch_tiles(local, tiles, ntc, ntr, nr, nc, ov, simple=simple)


@pytest.mark.parametrize('simple', [False, True])
def test_tiled_device_raster_vs_oracle(simple, oracle):
    """synthetic raster generated in HBM, whole-image k-means (subsample + device Lloyd),
    4 worker streams; compared with the oracle tile by tile + oracle stitch."""
    from pyshepseg_amd import tiling
    ras = tiling.DeviceRaster.synth(11, 6, 1500, 1300)
    try:
        img = ras.toArray()
        assert np.array_equal(img, oracle.synthimg(11, 6, 1500, 1300))
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=4)
        r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=512, overlapSize=128,
                                               minSegmentSize=50, numClusters=30,
                                               fixedKMeansInit=True, simpleTileRecode=simple,
                                               concurrencyCfg=cfg)
    finally:
        ras.free()
    want, mx, hist = _oracle_tiled(oracle, img, r.kmeans.cluster_centers_, 512, 128, 50,
                                   float(r.maxSpectralDiff), None, True, simple=simple)
    assert r.maxSegId == mx
    assert np.array_equal(r.segimg, want)
    assert np.array_equal(r.hist, hist)
    # size-independent properties of a stitched result
    assert r.hist[1:].sum() == (r.segimg != 0).sum()
    assert r.subsamplePcnt == pytest.approx(100 * min(1, np.sqrt(1e6 / (1500 * 1300))) ** 2)


def test_tiled_npy_roundtrip(tmp_path, oracle):
    from pyshepseg_amd import tiling, shepseg
    img = oracle.synthimg(5, 3, 400, 420)
    np.save(tmp_path / 'in.npy', img)
    km = shepseg.fitSpectralClusters(img, 8, 5, None, True)
    r = tiling.doTiledShepherdSegmentation(str(tmp_path / 'in.npy'), str(tmp_path / 'out.npy'),
                                           tileSize=128, overlapSize=32, minSegmentSize=20,
                                           kmeansObj=km)
    out = np.load(tmp_path / 'out.npy')
    want, mx, hist = _oracle_tiled(oracle, img, km.cluster_centers_, 128, 32, 20,
                                   float(r.maxSpectralDiff), None, True)
    assert np.array_equal(out, want) and r.maxSegId == mx
    assert np.array_equal(np.load(tmp_path / 'out_hist.npy'), hist)


@pytest.mark.parametrize('name', ['stitch_3x4_8conn', 'stitch_3x3_null', 'stitch_2x2'])
def test_overviews_vs_reference(name, golden, monkeypatch, tmp_path):
    """the output pyramid layers, written tile by tile like the reference's writeOverviews (levels
    2, 4, 8: the fixtures are below the size where the reference starts building them)"""
    from pyshepseg_amd import tiling, shepseg
    g = golden(name)
    gov = golden('overviews_stats')
    monkeypatch.setattr(tiling, 'overviewLevels', lambda xs, ys: [2, 4, 8])
    km = shepseg.KMeansModel(g['centres'])
    null = int(g['null_val']) if int(g['has_null']) else None
    r = tiling.doTiledShepherdSegmentation(
        g['img'], str(tmp_path / 'o.npy'), tileSize=int(g['tile_size']), overlapSize=int(g['overlap']),
        minSegmentSize=int(g['min_seg']), maxSpectralDiff=float(g['msd']), imgNullVal=null,
        fourConnected=bool(g['four']), kmeansObj=km)
    assert np.array_equal(np.load(tmp_path / 'o.npy'), g['mosaic'])
    for lvl in (2, 4, 8):
        assert np.array_equal(r.overviews[lvl], gov['%s_ov%d' % (name, lvl)]), lvl
        assert np.array_equal(np.load(tmp_path / ('o_ov%d.npy' % lvl)), r.overviews[lvl])
    assert ['%s=%s' % kv for kv in r.bandStatistics] == gov[name + '_stats'].tolist()


def test_streamed_input_and_output_match_the_per_tile_path(tmp_path, oracle, monkeypatch):
    """a .npy raster streamed into HBM block by block (several blocks per tile row) with a band
    selection, finished rows streamed out to a .npy memmap: same labels as the DeviceRaster path
    and as the per-tile read + upload path (SHEPSEG_STREAM_INPUT=0)"""
    from pyshepseg_amd import tiling, shepseg
    monkeypatch.setattr(tiling, 'STREAM_BLOCK_ROWS', 37)
    img = oracle.synthimg(9, 5, 700, 640)
    np.save(tmp_path / 'in.npy', img)
    sel = [1, 3, 4]                                     # 1-based band numbers, as the reference takes them
    sub = np.ascontiguousarray(img[[0, 2, 3]])
    km = shepseg.fitSpectralClusters(sub, 12, 5, None, True)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=4)
    kw = dict(tileSize=256, overlapSize=64, minSegmentSize=25, kmeansObj=km, concurrencyCfg=cfg)
    r = tiling.doTiledShepherdSegmentation(str(tmp_path / 'in.npy'), str(tmp_path / 'out.npy'),
                                           bandNumbers=sel, **kw)
    out = np.load(tmp_path / 'out.npy')
    want, mx, hist = _oracle_tiled(oracle, sub, km.cluster_centers_, 256, 64, 25,
                                   float(r.maxSpectralDiff), None, True)
    assert r.maxSegId == mx and np.array_equal(out, want) and np.array_equal(r.hist, hist)
    assert {'reading', 'writing', 'segmentation', 'stitchtiles'} <= set(r.timings.makeSummaryDict())
    monkeypatch.setenv('SHEPSEG_STREAM_INPUT', '0')
    r2 = tiling.doTiledShepherdSegmentation(img, None, bandNumbers=sel, **kw)
    assert np.array_equal(r2.segimg, want) and r2.maxSegId == mx


def test_failing_reader_surfaces(oracle):
    """an exception in the raster reader thread reaches the caller and nothing hangs"""
    import time
    from pyshepseg_amd import tiling, shepseg

    class Bad(tiling._ArraySource):
        def readRowsInto(self, bands, y0, y1, out):
            if y0 > 0:
                raise IOError("disk on fire")
            tiling._ArraySource.readRowsInto(self, bands, y0, y1, out)

    img = oracle.synthimg(2, 3, 2000, 300)
    km = shepseg.fitSpectralClusters(img, 8, 5, None, True)
    real = tiling._open_source
    tiling._open_source = lambda x: Bad(x) if isinstance(x, np.ndarray) else real(x)
    try:
        t0 = time.time()
        with pytest.raises(tiling.PyShepSegTilingError):
            tiling.doTiledShepherdSegmentation(img, None, tileSize=256, overlapSize=64, minSegmentSize=20,
                                               kmeansObj=km)
        assert time.time() - t0 < 60
    finally:
        tiling._open_source = real
    r = tiling.doTiledShepherdSegmentation(img, None, tileSize=256, overlapSize=64, minSegmentSize=20,
                                           kmeansObj=km)
    assert r.maxSegId > 0


@pytest.mark.parametrize('dtype,null', [(np.uint16, None), (np.int16, -7), (np.uint8, 0), (np.int32, 123456)])
def test_assign_rects_cluster_map_vs_oracle(dtype, null, oracle):
    """shp_assign_rects_dev: rectangles of a device raster into the raster-wide cluster map ==
    the oracle's km.predict of the same pixels (0 = null, else cluster + 1); pixels outside the
    rectangles are not touched; bad rectangles are refused."""
    import ctypes
    from pyshepseg_amd import tiling, _lib
    rng = np.random.default_rng(5)
    (nb, nr, nc, k) = (3, 301, 517, 7)
    info = np.iinfo(dtype)
    img = rng.integers(max(info.min, -3000), min(info.max, 3000), size=(nb, nr, nc)).astype(dtype)
    if null is not None:
        img[:, rng.integers(0, nr, 400), rng.integers(0, nc, 400)] = null
        img[1, 5, 5] = null                                           # null in one band only
    centres = rng.uniform(img.min(), img.max(), size=(k, nb))
    want = oracle.kmeans_assign(img, centres, null)               # 0 = null, else cluster + 1
    ras = tiling.DeviceRaster.fromArray(img, nullVal=null)
    c = _lib.ctx()
    d = ctypes.c_void_p()
    c.check(c._L.shp_dev_alloc(c.handle, nr * nc * 2, ctypes.byref(d)))
    try:
        c.check(c._L.shp_dev_memset(c.handle, d, 0xEE, nr * nc * 2))
        rects = np.array([[0, 0, 257, 5], [300, 17, 217, 284], [0, 300, 517, 1], [10, 40, 1, 3],
                          [20, 60, 0, 9]], dtype=np.int32)
        c.check(c._L.shp_assign_rects_dev(
            c.handle, ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[np.dtype(dtype)], nb, nr, nc,
            _lib.ptr(rects), rects.shape[0], _lib.ptr(centres), k, int(null is not None),
            0 if null is None else int(null), d))
        got = np.empty((nr, nc), dtype=np.uint16)
        c.check(c._L.shp_dev_download(c.handle, _lib.ptr(got), d, got.nbytes))
        touched = np.zeros((nr, nc), dtype=bool)
        for (x, y, w, h) in rects:
            touched[y:y + h, x:x + w] = True
        assert np.array_equal(got[touched], want[touched].astype(np.uint16))
        assert (got[~touched] == 0xEEEE).all()
        bad = np.array([[500, 0, 18, 4]], dtype=np.int32)
        rc = c._L.shp_assign_rects_dev(
            c.handle, ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[np.dtype(dtype)], nb, nr, nc,
            _lib.ptr(bad), 1, _lib.ptr(centres), k, 0, 0, d)
        assert rc != 0
    finally:
        c.check(c._L.shp_dev_free(c.handle, d))
        ras.free()


def test_tiled_device_raster_with_nulls_and_cluster_map(oracle, monkeypatch):
    """Device raster with a null value (null pixels and a null border), 8-connected: the tiled run
    through the raster-wide cluster map == the run with the per-tile assign step == the oracle."""
    from pyshepseg_amd import tiling, shepseg
    img = oracle.synthimg(3, 4, 700, 900)
    img[:, :40, :] = 0
    img[:, 300:340, 500:620] = 0
    img[2, 650, 10] = 0
    km = shepseg.fitSpectralClusters(img, 12, 5, 0, True)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=3)
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('SHEPSEG_CLUSTER_MAP', flag)
        ras = tiling.DeviceRaster.fromArray(img, nullVal=0)
        try:
            r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=256, overlapSize=64,
                                                   minSegmentSize=30, imgNullVal=0, kmeansObj=km,
                                                   fourConnected=False, concurrencyCfg=cfg)
        finally:
            ras.free()
        outs.append(r)
    want, mx, hist = _oracle_tiled(oracle, img, km.cluster_centers_, 256, 64, 30,
                                   float(outs[0].maxSpectralDiff), 0, False)
    for r in outs:
        assert r.maxSegId == mx
        assert np.array_equal(r.segimg, want)
        assert np.array_equal(r.hist, hist)


def test_worker_failure_surfaces_and_does_not_hang(oracle):
    """A failure inside the workers' first call (a model with more clusters than the assign step
    accepts) must come back as an exception -- with the cluster map's claimed blocks, the fill gate
    and the other workers released -- and the next run on the same contexts must work."""
    import time
    from pyshepseg_amd import tiling, shepseg, _lib
    img = oracle.synthimg(2, 3, 600, 700)
    bad = shepseg.KMeansModel(np.zeros((70000, 3)))
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=4)
    ras = tiling.DeviceRaster.fromArray(img)
    try:
        t0 = time.time()
        with pytest.raises((tiling.PyShepSegTilingError, _lib.ShepsegHipError)):
            tiling.doTiledShepherdSegmentation(ras, None, tileSize=256, overlapSize=64, minSegmentSize=20,
                                               kmeansObj=bad, maxSpectralDiff=100.0, concurrencyCfg=cfg)
        assert time.time() - t0 < 30
        km = shepseg.fitSpectralClusters(img, 8, 5, None, True)
        r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=256, overlapSize=64, minSegmentSize=20,
                                               kmeansObj=km, concurrencyCfg=cfg)
    finally:
        ras.free()
    want, mx, hist = _oracle_tiled(oracle, img, km.cluster_centers_, 256, 64, 20,
                                   float(r.maxSpectralDiff), None, True)
    assert r.maxSegId == mx and np.array_equal(r.segimg, want)


def test_knobs_off_paths_match(tmp_path):
    """The documented fall-back paths -- no fill gate, no cluster map (per-tile assign on a copied
    window), pixel sort instead of run sort -- give the same labels.  The knobs are read once per
    process, so the run with them off happens in a child process."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from pyshepseg_amd import tiling
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from pyshepseg_amd import tiling\n"
        "ras = tiling.DeviceRaster.synth(7, 4, 900, 1100)\n"
        "cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=3)\n"
        "r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=384, overlapSize=96, minSegmentSize=40,\n"
        "        numClusters=20, fixedKMeansInit=True, concurrencyCfg=cfg)\n"
        "np.savez(sys.argv[1], seg=r.segimg, hist=r.hist, centres=r.kmeans.cluster_centers_)\n" % ROOT)
    env = dict(os.environ, SHEPSEG_FILL_MAX='0', SHEPSEG_CLUSTER_MAP='0', SHEPSEG_CSR_RUNS='0')
    out = str(tmp_path / 'off.npz')
    p = subprocess.run([sys.executable, '-c', code, out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    off = np.load(out)
    ras = tiling.DeviceRaster.synth(7, 4, 900, 1100)
    try:
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=3)
        r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=384, overlapSize=96, minSegmentSize=40,
                                               numClusters=20, fixedKMeansInit=True, concurrencyCfg=cfg)
    finally:
        ras.free()
    assert np.array_equal(off['centres'], r.kmeans.cluster_centers_)
    assert np.array_equal(off['seg'], r.segimg)
    assert np.array_equal(off['hist'], r.hist)


@pytest.mark.parametrize('dtype,nb,null,four', [(np.uint8, 3, 0, False), (np.int16, 5, -32768, True),
                                                (np.int32, 2, None, False)])
def test_tiled_device_raster_other_pixel_types(dtype, nb, null, four, oracle):
    """Tiled run on device rasters of the other pixel types (the in-place window reads, the
    cluster map and the spectra kernels are all templated on the type), both connectivities,
    1024-px tiles: equal to the oracle tile by tile + oracle stitch."""
    from pyshepseg_amd import tiling, shepseg
    base = oracle.synthimg(9, nb, 2300, 2100).astype(np.int64)
    if dtype == np.uint8:
        img = ((base - base.min()) * 255 // max(int(base.max() - base.min()), 1)).astype(np.uint8)
        img[img == 0] = 1
    elif dtype == np.int16:
        img = (base - 3000).astype(np.int16)
    else:
        img = (base * 70000 - 100000000).astype(np.int32)
    if null is not None:
        img[:, 700:760, 300:1500] = null
        img[:, :, :25] = null
        img[0, 1999, 1999] = null
    km = shepseg.fitSpectralClusters(img, 16, 2, null, True)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=4)
    ras = tiling.DeviceRaster.fromArray(img, nullVal=null)
    try:
        r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=1024, overlapSize=128,
                                               minSegmentSize=60, imgNullVal=null, kmeansObj=km,
                                               fourConnected=four, concurrencyCfg=cfg)
    finally:
        ras.free()
    want, mx, hist = _oracle_tiled(oracle, img, km.cluster_centers_, 1024, 128, 60,
                                   float(r.maxSpectralDiff), null, four)
    assert r.maxSegId == mx
    assert np.array_equal(r.segimg, want)
    assert np.array_equal(r.hist, hist)


def test_worker_count_follows_free_memory(capfd):
    """The tiled driver starts only as many worker contexts as fit the free device memory for the
    job's largest tile (a 2-Gpx tile needs ~150 GB per context: at most one of 20 fits 288 GB)."""
    from pyshepseg_amd import tiling, _lib
    assert tiling._workersThatFit(6, _lib.SHP_DTYPES[np.dtype(np.uint16)], 6, 256 * 256) == 6
    n = tiling._workersThatFit(20, _lib.SHP_DTYPES[np.dtype(np.uint16)], 6, 2000000000)
    assert n == 1
    assert 'worker streams fit' in capfd.readouterr().err
