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
