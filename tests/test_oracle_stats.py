"""CPU: the oracle's per-segment statistics against the reference's golden vectors
(accumulateSegDict / SegmentStats / RatPage run under real numba)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

STATS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'stats_*.npz')))


def selection(g):
    sel = []
    for name, par in zip(g['sel_names'].tolist(), g['sel_params'].tolist()):
        sel.append(('c%d' % len(sel), name, int(par)) if name == 'percentile' else ('c%d' % len(sel), name))
    return sel


@pytest.mark.parametrize('name', STATS)
def test_oracle_segstats(name, golden, oracle):
    g = golden(name)
    null = int(g['null_val']) if int(g['has_null']) else None
    ic, fc = oracle.segstats(g['seg'], g['band'], selection(g), null, int(g['missing']))
    done = g['complete']
    assert done[1:].all()                       # every id of these fixtures has pixels
    assert np.array_equal(ic[:, done], g['intcols'][:, done])
    assert np.array_equal(fc[:, done], g['floatcols'][:, done])      # bit-exact float32


def test_spatial_stats_vs_reference(golden, oracle):
    """Built-in user functions of calcPerSegmentSpatialStatsTiled (reference njit code driven by
    oracle/refgen/gen_golden_spatial.py): the oracle restatement is bit-identical."""
    g = golden('spatial_stats')
    seg, band, tile = g['seg'], g['band'], int(g['tile'])
    _ic, fc = oracle.spatialstats(seg, band, 'meancoord', g['transform'], 0, 0, 2, tile_size=tile)
    assert np.array_equal(fc.view(np.uint32), g['mean_fc'].view(np.uint32))
    _ic, fc = oracle.spatialstats(seg, band, 'meancoord', g['rot'], 0, 0, 2, tile_size=tile)
    assert np.array_equal(fc.view(np.uint32), g['meanrot_fc'].view(np.uint32))
    ic, _fc = oracle.spatialstats(seg, band, 'numedge', 1, 0, 1, 0, tile_size=tile)
    assert np.array_equal(ic, g['edge4_ic'])
    ic, _fc = oracle.spatialstats(seg, band, 'numedge', 0, 0, 1, 0, tile_size=tile)
    assert np.array_equal(ic, g['edge8_ic'])
    _ic, fc = oracle.spatialstats(seg, band, 'variogram', 4, 0, 0, 4, tile_size=tile)
    assert np.array_equal(fc.view(np.uint32), g['vario_fc'].view(np.uint32))
    # the fixture exercises the corners: an all-nodata segment and unset variogram bins stay missing
    assert (g['mean_fc'][0, 1:] == -9999).any() and (g['edge4_ic'][0, 1:] > 0).any()
