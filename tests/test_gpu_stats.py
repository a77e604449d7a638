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


def test_stats_errors():
    from pyshepseg_amd import tilingstats
    with pytest.raises(tilingstats.PyShepSegStatsError):
        tilingstats.calcPerSegmentStats(np.ones((4, 4), np.uint32), np.ones((4, 4), np.float32),
                                        [('m', 'mean')])
    with pytest.raises(tilingstats.PyShepSegStatsError):
        tilingstats.calcPerSegmentStats(np.ones((4, 4), np.uint32), np.ones((4, 5), np.uint16),
                                        [('m', 'mean')])
