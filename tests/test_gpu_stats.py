"""GPU parity of the per-segment statistics against the reference's golden vectors and against
the oracle on a stitched segmentation.  Integer statistics exact; float statistics within 1e-6
relative (the bar of BASELINE.json) -- and in fact bit-identical."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_stats import selection

pytestmark = pytest.mark.gpu
STATS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'stats_*.npz')))


@pytest.mark.parametrize('name', STATS)
def test_golden_stats(name, golden):
    from pyshepseg_amd import tilingstats
    g = golden(name)
    null = int(g['null_val']) if int(g['has_null']) else None
    sel = selection(g)
    r = tilingstats.calcPerSegmentStatsTiled(g['band'], 1, g['seg'], sel,
                                             missingStatsValue=int(g['missing']), imgNullVal=null)
    ii = ff = 0
    for (col, stat) in [(s[0], s[1]) for s in sel]:
        got = r.columns[col]
        if stat in ('mean', 'stddev'):
            want = g['floatcols'][ff]; ff += 1
            assert got.dtype == np.float32
            assert np.allclose(got[1:], want[1:], rtol=1e-6, atol=0)
            assert np.array_equal(got[1:], want[1:])            # bit-identical in practice
        else:
            want = g['intcols'][ii]; ii += 1
            assert got.dtype == np.int64 and np.array_equal(got[1:], want[1:])
        assert got[0] == 0


def test_stats_vs_oracle_large(oracle):
    from pyshepseg_amd import tilingstats
    rng = np.random.RandomState(3)
    # blocky labels with ~30-px segments (C5-like density) + some big ones, uint16 band with nodata
    seg = (np.arange(700)[:, None] // 5 * 200 + np.arange(900)[None, :] // 6 + 1).astype(np.uint32)
    seg[300:500, 100:700] = 5
    seg[:7] = 0
    band = oracle.synthimg(17, 1, 700, 900)[0]
    band[rng.rand(700, 900) < 0.02] = 0
    sel = [('mn', 'min'), ('mx', 'max'), ('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'),
           ('mode', 'mode'), ('p90', 'percentile', 90), ('n', 'pixcount')]
    ic, fc = oracle.segstats(seg, band, sel, null_val=0)
    r = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=0)
    for i, name in enumerate(['mn', 'mx', 'med', 'mode', 'p90', 'n']):
        assert np.array_equal(r.columns[name], ic[i]), name
    assert np.allclose(r.columns['mean'], fc[0], rtol=1e-6, atol=0)
    assert np.allclose(r.columns['sd'], fc[1], rtol=1e-6, atol=1e-6)
    # pixcount of every segment sums to the number of valid, non-null-segment pixels
    assert r.columns['n'].sum() == ((seg != 0) & (band != 0)).sum()


def test_stats_streamed_in_pages(oracle, monkeypatch):
    """row blocks smaller than the segments (so most of them straddle a block boundary), RAT pages
    of 100 rows, ids without pixels: same columns as the oracle, pages written once each, in the
    reference's layout (startSegId = multiple of the page size, the last page shorter)"""
    from pyshepseg_amd import tilingstats
    monkeypatch.setattr(tilingstats, 'RAT_PAGE_SIZE', 100)
    rng = np.random.RandomState(5)
    seg = (np.arange(300)[:, None] // 7 * 60 + np.arange(400)[None, :] // 9 + 1).astype(np.uint32)
    seg[100:220, 50:300] = 77                      # a big segment across many blocks
    seg[seg == 300] = 301                          # id 300 has no pixels (the stitch can leave such ids)
    seg[seg == 1234] = 0
    seg[:3] = 0
    band = oracle.synthimg(23, 1, 300, 400)[0]
    band[rng.rand(300, 400) < 0.05] = 7
    sel = [('mn', 'min'), ('mx', 'max'), ('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'),
           ('mode', 'mode'), ('p10', 'percentile', 10), ('n', 'pixcount')]
    S = int(seg.max())
    ic, fc = oracle.segstats(seg, band, sel, null_val=7)
    whole = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=7)
    for chunk in (400 * 5, 400 * 64, 400 * 300):
        r = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=7, chunkPixels=chunk)
        for i, name in enumerate(['mn', 'mx', 'med', 'mode', 'p10', 'n']):
            assert np.array_equal(r.columns[name], ic[i]), (chunk, name)
            assert np.array_equal(r.columns[name], whole.columns[name])
        assert np.array_equal(r.columns['mean'].view(np.uint32), fc[0].view(np.uint32))
        assert np.array_equal(r.columns['sd'].view(np.uint32), fc[1].view(np.uint32))
        starts = sorted(p[0] for p in r.pagesWritten)
        assert starts == list(range(0, S + 1, 100)) and len(set(starts)) == len(starts)
        assert dict(r.pagesWritten)[starts[-1]] == S + 1 - starts[-1]
        # the id without pixels: missing value everywhere, 0 pixels
        assert r.columns['n'][300] == 0 and r.columns['mn'][300] == -9999 and r.columns['mean'][300] == -9999
    assert set(r.timings.makeSummaryDict()) >= {'reading', 'accumulation', 'statscompletion', 'writing'}


def test_rat_page_semantics():
    """RatPage / getRatPageId as the reference defines them (tilingstats.py:1950-2045)"""
    from pyshepseg_amd import tilingstats as ts
    assert ts.getRatPageId(0) == 0 and ts.getRatPageId(99999) == 0 and ts.getRatPageId(100000) == 100000
    p0 = ts.RatPage(2, 1, 0, 5)
    assert p0.complete.tolist() == [True, False, False, False, False]
    assert p0.intcols.dtype == np.int64 and p0.floatcols.dtype == np.float32
    assert (p0.intcols[:, 0] == 0).all() and (p0.floatcols[:, 0] == 0).all() and not p0.pageComplete()
    p0.setRatVal(3, ts.STAT_DTYPE_INT, 1, 42)
    p0.setRatVal(3, ts.STAT_DTYPE_FLOAT, 0, 2.5)
    assert p0.getRatVal(3, ts.STAT_DTYPE_INT, 1) == 42 and p0.getRatVal(3, ts.STAT_DTYPE_FLOAT, 0) == 2.5
    for s_ in (1, 2, 3, 4):
        p0.setSegmentComplete(s_)
    assert p0.getSegmentComplete(2) and p0.pageComplete()
    p1 = ts.RatPage(1, 1, 100000, 3)
    assert not p1.complete.any() and p1.getIndexInPage(100002) == 2


def test_stats_errors():
    from pyshepseg_amd import tilingstats
    with pytest.raises(tilingstats.PyShepSegStatsError):
        tilingstats.calcPerSegmentStats(np.ones((4, 4), np.uint32), np.ones((4, 4), np.float32),
                                        [('m', 'mean')])
    with pytest.raises(tilingstats.PyShepSegStatsError):
        tilingstats.calcPerSegmentStats(np.ones((4, 4), np.uint32), np.ones((4, 5), np.uint16),
                                        [('m', 'mean')])


def _spatial_cases(g):
    from pyshepseg_amd import tilingstats as ts
    R, I = ts.GFT_Real, ts.GFT_Integer
    return [('mean_fc', ts.userFuncMeanCoord, g['transform'], [('e', R), ('n', R)]),
            ('meanrot_fc', ts.userFuncMeanCoord, g['rot'], [('e', R), ('n', R)]),
            ('edge4_ic', ts.userFuncNumEdgePixels, True, [('edges', I)]),
            ('edge8_ic', ts.userFuncNumEdgePixels, False, [('edges', I)]),
            ('vario_fc', ts.userFuncVariogram, 4, [('v%d' % i, R) for i in range(4)])]


def test_golden_spatial_stats(golden):
    """Built-in spatial user functions against the reference's own njit code (goldens)."""
    from pyshepseg_amd import tilingstats as ts
    g = golden('spatial_stats')
    for key, fn, prm, cols in _spatial_cases(g):
        r = ts.calcPerSegmentSpatialStatsTiled(g['band'], 1, g['seg'], cols, fn, prm, imgNullVal=0)
        got = np.stack([r.columns[n] for (n, _t) in cols])
        want = g[key]
        if key.startswith('mean'):
            # float64 sums of transformed coordinates are re-associated on the device (n*t0 +
            # t1*sum(x) + t2*sum(y)): within 1e-6 relative (north_star), almost always bit-equal
            assert np.allclose(got, want, rtol=1e-6, atol=0)
            assert (got.view(np.uint32) == want.view(np.uint32)).mean() > 0.99
        elif got.dtype == np.float32:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        else:
            assert np.array_equal(got, want)


def test_spatial_stats_vs_oracle_large(oracle):
    """2000 x 1800, default 1024 tiles in the oracle's point order, several dtypes' worth of values."""
    from pyshepseg_amd import tilingstats as ts
    rng = np.random.RandomState(4)
    base = rng.permutation(np.arange(1, 30 * 25 + 1)).reshape(30, 25).astype(np.uint32)
    seg = np.kron(base, np.ones((70, 75), dtype=np.uint32))[:2000, :1800]
    seg[rng.rand(*seg.shape) < 0.02] = 0
    band = (oracle.synthimg(8, 1, 2000, 1800)[0]).astype(np.uint16)
    band[rng.rand(*band.shape) < 0.05] = 65535
    S = int(seg.max())
    tr = np.array([300000.0, 10.0, 0.0, 7000000.0, 0.0, -10.0])
    R, I = ts.GFT_Real, ts.GFT_Integer
    ic, fc = ts.calcPerSegmentSpatialStats(seg, band, [R, R], ts.userFuncMeanCoord, tr, 65535)
    _wi, wf = oracle.spatialstats(seg, band, 'meancoord', tr, 65535, 0, 2, max_seg_id=S)
    assert np.array_equal(fc.view(np.uint32), wf.view(np.uint32))     # integer-valued transform: exact sums
    for four in (True, False):
        ic, fc = ts.calcPerSegmentSpatialStats(seg, band, [I, I], ts.userFuncNumEdgePixels, four, 65535)
        wi, _wf = oracle.spatialstats(seg, band, 'numedge', int(four), 65535, 2, 0, max_seg_id=S)
        assert np.array_equal(ic, wi)                                  # second int column stays missing
    ic, fc = ts.calcPerSegmentSpatialStats(seg, band, [R] * 7, ts.userFuncVariogram, 6, 65535)
    _wi, wf = oracle.spatialstats(seg, band, 'variogram', 6, 65535, 0, 7, max_seg_id=S)
    assert np.array_equal(fc.view(np.uint32), wf.view(np.uint32))


def test_spatial_stats_errors():
    from pyshepseg_amd import tilingstats as ts
    seg = np.ones((8, 8), np.uint32)
    img = np.ones((8, 8), np.uint16)
    with pytest.raises(ts.PyShepSegStatsError, match='NoData value must be set'):
        ts.calcPerSegmentSpatialStatsTiled(img, 1, seg, [('a', ts.GFT_Real)], ts.userFuncMeanCoord, [0] * 6)
    with pytest.raises(ts.PyShepSegStatsError, match='one or more columns'):
        ts.calcPerSegmentSpatialStatsTiled(img, 1, seg, [], ts.userFuncMeanCoord, [0] * 6, imgNullVal=0)
    with pytest.raises(ts.PyShepSegStatsError, match='built-in user functions'):
        ts.calcPerSegmentSpatialStatsTiled(img, 1, seg, [('a', ts.GFT_Real)], lambda *a: None, None,
                                           imgNullVal=0)


def test_stats_on_device_resident_rasters(oracle):
    """Segmentation kept in HBM -> statistics without a copy == statistics of the downloaded arrays."""
    from pyshepseg_amd import tiling, tilingstats as ts
    ras = tiling.DeviceRaster.synth(3, 4, 700, 900)
    try:
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=3)
        kw = dict(tileSize=256, overlapSize=64, minSegmentSize=30, numClusters=12, fixedKMeansInit=True,
                  concurrencyCfg=cfg)
        rd = tiling.doTiledShepherdSegmentation(ras, tiling._KEEP_ON_DEVICE, **kw)
        sel = [('m', 'mean'), ('s', 'stddev'), ('p', 'percentile', 75), ('n', 'pixcount'), ('mo', 'mode')]
        got = ts.calcPerSegmentStatsTiled(ras, 2, rd, sel)
        tiling.freeDeviceOutput(rd)
        rh = tiling.doTiledShepherdSegmentation(ras, None, **kw)
        img = ras.toArray()
    finally:
        ras.free()
    want = ts.calcPerSegmentStatsTiled(img, 2, rh.segimg, sel)
    wi, wf = oracle.segstats(rh.segimg, img[1], sel, max_seg_id=rh.maxSegId)
    for name in ('m', 's'):
        assert np.array_equal(got.columns[name].view(np.uint32), want.columns[name].view(np.uint32))
    for name in ('p', 'n', 'mo'):
        assert np.array_equal(got.columns[name], want.columns[name])
    assert np.array_equal(got.columns['n'], wi[1]) and np.array_equal(got.columns['m'].view(np.uint32), wf[0].view(np.uint32))


def test_stats_giant_segment_vs_oracle(oracle):
    """a segment of 1.9 M pixels among small ones: its value run is one wavefront's work
    (k_seg_stats_big: parallel sum, runs of equal values closed in order), same columns as the oracle"""
    from pyshepseg_amd import tilingstats
    rng = np.random.RandomState(12)
    seg = np.ones((1500, 1500), dtype=np.uint32)
    seg[:250] = (np.arange(250 * 1500).reshape(250, 1500) // 300 + 2).astype(np.uint32)
    seg[700:720, 100:900] = 0
    band = rng.randint(0, 900, size=seg.shape).astype(np.uint16)          # few distinct values: long runs
    band[300:1400:7] = 65535
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'), ('f', 'mode'),
           ('g', 'percentile', 90), ('h', 'percentile', 0), ('i', 'pixcount')]
    S = int(seg.max())
    for null in (None, 65535):
        ic, fc, _fast = tilingstats.calcPerSegmentStats(seg, band, sel, imgNullVal=null, maxSegId=S)
        wic, wfc = oracle.segstats(seg, band, sel, null, -9999, max_seg_id=S)
        assert np.array_equal(ic, wic)
        assert np.array_equal(fc.view(np.uint32), wfc.view(np.uint32))


@pytest.mark.parametrize('mode', ['blocks', 'ragged', 'crowded'])
def test_stats_patch_path_equals_sort_path(mode, oracle, monkeypatch):
    """small segments: the patch-by-patch path (segstats.h k_stats_patch) against the sort path and the oracle --
    aligned 4 x 8 blocks (every segment complete in its patch), ragged segments that straddle patches and hold
    nodata pixels (the leftovers go through the sorts), and a patch with more distinct labels than its table takes"""
    from pyshepseg_amd import tilingstats
    rng = np.random.RandomState(5)
    (nr, nc) = (203, 331)
    if mode == 'blocks':
        seg = ((np.arange(nr)[:, None] // 4) * ((nc + 7) // 8) + np.arange(nc)[None, :] // 8 + 1).astype(np.uint32)
    elif mode == 'ragged':
        seg = ((np.arange(nr)[:, None] // 5) * 100 + np.arange(nc)[None, :] // 7 + 1).astype(np.uint32)
        seg[40:90, 100:260] = 7                     # a segment far too long for a thread's sort
        seg[:3] = 0
    else:
        seg = (np.arange(nr * nc, dtype=np.uint32).reshape(nr, nc) // 2 + 1).astype(np.uint32)      # 2-pixel segments
    band = oracle.synthimg(19, 1, nr, nc)[0].copy()
    band[rng.rand(nr, nc) < 0.05] = 0
    sel = [('mn', 'min'), ('mx', 'max'), ('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'),
           ('mode', 'mode'), ('p25', 'percentile', 25), ('n', 'pixcount')]
    ic, fc = oracle.segstats(seg, band, sel, null_val=0)
    got = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('SHEPSEG_STATS_PATCH', flag)
        r = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=0)
        got[flag] = r
    for r in got.values():
        for i, name in enumerate(['mn', 'mx', 'med', 'mode', 'p25', 'n']):
            assert np.array_equal(r.columns[name], ic[i]), (mode, name)
        assert np.array_equal(r.columns['mean'].view(np.uint32), fc[0].view(np.uint32)), mode
        assert np.array_equal(r.columns['sd'].view(np.uint32), fc[1].view(np.uint32)), mode


@pytest.mark.parametrize('dtype', ['uint8', 'int16', 'int32', 'uint32'])
@pytest.mark.parametrize('mode', ['blocks', 'ragged'])
def test_stats_patch_path_dtypes(mode, dtype, oracle, monkeypatch):
    """the patch path's two rank forms -- value << 6 | position in one compare for 8/16-bit bands, (value, position)
    for 32-bit ones -- on bands with many equal values (uint8), negative values (int16) and values that need all
    32 bits, against the sort path and the oracle (segstats.h k_stats_patch)"""
    from pyshepseg_amd import tilingstats
    (nr, nc) = (161, 273)
    if mode == 'blocks':
        seg = ((np.arange(nr)[:, None] // 4) * ((nc + 7) // 8) + np.arange(nc)[None, :] // 8 + 1).astype(np.uint32)
    else:
        seg = ((np.arange(nr)[:, None] // 6) * 100 + np.arange(nc)[None, :] // 9 + 1).astype(np.uint32)
        seg[-2:] = 0
    base = oracle.synthimg(23, 1, nr, nc)[0].astype(np.int64)
    if dtype == 'uint8':
        band = (base >> 5).astype(np.uint8)                      # ~50 levels: ties everywhere, the mode matters
    elif dtype == 'int16':
        band = (base - 32768).astype(np.int16)
    elif dtype == 'int32':
        band = ((base - 2720) * 2500000).astype(np.int32)        # both signs, up to +-2.08e9
        band[::3, ::2] = band[0, 0]                              # ties among wide values
    else:
        band = (base * 1200000).astype(np.uint32)                # up to 4.26e9
        band[1::3, ::2] = np.uint32(0xFFFFFFFF)
    null = int(band[3, 5])
    sel = [('mn', 'min'), ('mx', 'max'), ('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'),
           ('mode', 'mode'), ('p80', 'percentile', 80), ('n', 'pixcount')]
    ic, fc = oracle.segstats(seg, band, sel, null_val=null)
    for flag in ('1', '0'):
        monkeypatch.setenv('SHEPSEG_STATS_PATCH', flag)
        r = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=null)
        for i, name in enumerate(['mn', 'mx', 'med', 'mode', 'p80', 'n']):
            assert np.array_equal(r.columns[name], ic[i]), (mode, dtype, flag, name)
        assert np.array_equal(r.columns['mean'].view(np.uint32), fc[0].view(np.uint32)), (mode, dtype, flag)
        assert np.array_equal(r.columns['sd'].view(np.uint32), fc[1].view(np.uint32)), (mode, dtype, flag)
