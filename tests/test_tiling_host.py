"""CPU: host logic of the tiled driver (tile grid, subsample rule) and the oracle's stitch
restatement against the reference's golden stitch vectors."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

STITCH = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'stitch_*.npz')))


class _Ds(object):
    def __init__(self, ys, xs):
        self.RasterYSize, self.RasterXSize = ys, xs


def test_tile_grid_matches_reference_rule(oracle):
    from pyshepseg_amd import tiling
    # C3 geometry: 40000^2, tile 4096 / overlap 1024 -> 12x12 tiles, last one 6208
    ti = tiling.getTilesForFile(_Ds(40000, 40000), 4096, 1024)
    assert (ti.ncols, ti.nrows, ti.getNumTiles()) == (12, 12, 144)
    assert ti.getTile(0, 0) == (0, 0, 4096, 4096)
    assert ti.getTile(11, 11) == (33792, 33792, 6208, 6208)
    assert ti.getTile(10, 3) == (30720, 9216, 4096, 4096)
    for (ys, xs, t, o) in [(200, 200, 96, 32), (300, 280, 96, 32), (260, 330, 80, 24), (50, 1000, 64, 16)]:
        ti = tiling.getTilesForFile(_Ds(ys, xs), t, o)
        tiles, nc, nr = oracle.get_tiles(ys, xs, t, o)
        assert (ti.ncols, ti.nrows) == (nc, nr) and ti.tiles == tiles


def test_tile_grid_sweep_against_loop_form(oracle):
    """the closed-form grid against the oracle's loop restatement of tiling.py:376-443"""
    from pyshepseg_amd import tiling
    rng = np.random.RandomState(3)
    cases = [(1, 1, 8, 2), (0, 10, 8, 2), (10, 0, 8, 2), (16, 16, 8, 2), (15, 17, 8, 2), (8, 8, 8, 0)]
    for _ in range(300):
        t = int(rng.randint(2, 60))
        o = int(rng.randint(0, t))
        cases.append((int(rng.randint(1, 400)), int(rng.randint(1, 400)), t, o))
    for (ys, xs, t, o) in cases:
        ti = tiling.getTilesForFile(_Ds(ys, xs), t, o)
        tiles, nc, nr = oracle.get_tiles(ys, xs, t, o)
        assert (ti.ncols, ti.nrows) == (nc, nr) and ti.tiles == tiles, (ys, xs, t, o)
    with pytest.raises(tiling.PyShepSegTilingError):
        tiling.getTilesForFile(_Ds(100, 100), 16, 16)


def test_diagonal_centres_cast_like_reference():
    """centres[i] = bandMin + (i + 1) * (bandMax - bandMin) / (k + 1) truncated to the pixel type
    (shepseg.py:388-397), written out per cluster here"""
    from pyshepseg_amd import shepseg
    rng = np.random.RandomState(5)
    for dt in (np.uint8, np.int16, np.uint16, np.int32):
        x = rng.randint(0, 200, size=(500, 4)).astype(dt)
        if dt == np.int16:
            x -= 90
        for k in (1, 7, 60):
            got = shepseg.diagonalClusterCentres(x, k)
            mn, mx = x.min(axis=0), x.max(axis=0)
            want = np.empty((k, 4), dtype=dt)
            for i in range(k):
                want[i] = mn + (i + 1) * ((mx - mn) / (k + 1))
            assert got.dtype == x.dtype and np.array_equal(got, want)


def test_overview_levels_and_band_statistics(golden):
    """setupOverviews' level rule and estimateStatsFromHisto's metadata strings against the
    reference's own output (oracle/refgen/gen_golden_overviews.py)"""
    from pyshepseg_amd import tiling
    g = golden('overviews_stats')
    for key in [k for k in g if k.startswith('levels_')]:
        size = int(key.split('_')[1])
        assert tiling.overviewLevels(size, size // 2) == g[key].tolist(), key
        assert tiling.overviewLevels(size // 2, size) == g[key].tolist(), key
    for case in ('stitch_3x4_8conn', 'stitch_3x3_null', 'stitch_2x2'):
        hist = golden(case)['hist']
        got = ['%s=%s' % kv for kv in tiling.estimateStatsFromHisto(hist)]
        assert got == g[case + '_stats'].tolist(), case


def test_band_statistics_compiled_equals_numpy():
    """estimateStatsFromHisto evaluates the reference's numpy expressions (utils.py:54-70) in one compiled pass
    (shp_hist_stats, host only): same strings as numpy's own evaluation, also where a .sum() spans several
    8192-element reduction blocks and for histograms of a few million bins (the C3 benchmark has 2.6 M)."""
    from pyshepseg_amd import tiling

    def ref(hist):
        mask = hist > 0
        nVals = hist.sum()
        minVal = mask.argmax()
        maxVal = hist.shape[0] - np.flip(mask).argmax() - 1
        values = np.arange(hist.shape[0])
        meanVal = (values * hist).sum() / nVals
        sd = np.sqrt((hist * np.power(values - meanVal, 2)).sum() / nVals)
        med = (hist.cumsum() >= hist.sum() / 2).nonzero()[0][0]
        return [repr(int(minVal)), repr(int(maxVal)), repr(float(meanVal)), repr(float(sd)),
                repr(int(np.argmax(hist))), repr(int(med))]

    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 20000, 131073, 700001):
        for trial in range(4):
            h = rng.integers(0, 5000, size=n).astype(np.uint32)
            if trial == 1 and n > 4:
                h[:n // 3] = 0
                h[-(n // 4):] = 0
            if trial == 3:
                h = (h % 3 == 0).astype(np.uint32) * rng.integers(1, 2 ** 31, size=n).astype(np.uint32)
            if h.sum() == 0:
                h[0] = 1
            got = [v for (_k, v) in tiling.estimateStatsFromHisto(h)][:6]
            assert got == ref(h), (n, trial)


def test_subsample_indices_restart_per_block():
    from pyshepseg_amd import tiling
    idx = tiling._subsample_indices(2500, 40)
    want = np.concatenate([np.arange(s, min(s + 1024, 2500), 40) for s in range(0, 2500, 1024)])
    assert np.array_equal(idx, want)
    src = tiling._ArraySource(np.arange(3 * 2100 * 1500, dtype=np.uint32).reshape(3, 2100, 1500))
    sub = tiling.readSubsampledImage(src, [1, 3], 1 / 7.0)
    ry = tiling._subsample_indices(2100, 7)
    rx = tiling._subsample_indices(1500, 7)
    assert np.array_equal(sub, src.arr[[0, 2]][:, ry][:, :, rx])


def test_odd_overlap_rejected():
    from pyshepseg_amd import tiling
    with pytest.raises(tiling.PyShepSegTilingError):
        tiling.doTiledShepherdSegmentation(np.zeros((1, 8, 8), np.uint8), None, overlapSize=3)
    with pytest.raises(ValueError):
        tiling.SegmentationConcurrencyConfig(concurrencyType='nope')


@pytest.mark.parametrize('name', STITCH)
def test_oracle_stitch_vs_reference(name, golden, oracle):
    g = golden(name)
    nr, nc = g['mosaic'].shape
    tiles, ntc, ntr = oracle.get_tiles(nr, nc, int(g['tile_size']), int(g['overlap']))
    assert (ntc, ntr) == (int(g['ntcols']), int(g['ntrows']))
    local = {(c, r): g['local_%d_%d' % (c, r)] for (c, r) in tiles}
    out, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, nr, nc, int(g['overlap']))
    assert mx == int(g['max_seg_id'])
    assert np.array_equal(out, g['mosaic'])
    assert np.array_equal(hist, g['hist'])


@pytest.mark.parametrize('name', STITCH)
def test_oracle_tiles_vs_reference(name, golden, oracle):
    """per-tile labels of the stitch fixtures (doShepherdSegmentation with a shared model)"""
    g = golden(name)
    nr, nc = g['mosaic'].shape
    tiles, _c, _r = oracle.get_tiles(nr, nc, int(g['tile_size']), int(g['overlap']))
    null = int(g['null_val']) if int(g['has_null']) else None
    for (c, r), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(g['img'][:, y:y + ys, x:x + xs])
        got = oracle.segment_tile(sub, g['centres'], int(g['min_seg']), float(g['msd']), null,
                                  bool(g['four']))
        assert np.array_equal(got['segimg'], g['local_%d_%d' % (c, r)])
