"""GPU: subset.subsetImage's recode (shp_subset_recode) against the reference's goldens and the
oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,masked', [('a', False), ('b', True), ('c', False), ('d', True)])
def test_golden_subset(name, masked, golden):
    from pyshepseg_amd import subset
    g = golden('subset_recode')
    (tlx, tly, xs, ys, tile) = [int(v) for v in g[name + '_win']]
    r = subset.subsetImage(g['seg'], None, tlx, tly, xs, ys, maskImage=g['mask'] if masked else None,
                           tileSize=tile, origSegIdColName='orig')
    assert np.array_equal(r.segimg, g[name + '_out'])
    assert np.array_equal(r.origSegIds, g[name + '_orig'])
    assert np.array_equal(r.hist, g[name + '_hist'])
    assert np.array_equal(r.columns['orig'], g[name + '_orig'].astype(np.int32))


def test_subset_vs_oracle_large(oracle, tmp_path):
    """Several 1024-tiles, ragged last tiles, a mask, RAT columns gathered to the new ids."""
    from pyshepseg_amd import subset
    rng = np.random.RandomState(11)
    # blocky labels: ids in scrambled order so first-seen order differs from id order
    base = rng.permutation(np.arange(1, 40 * 36 + 1)).reshape(40, 36).astype(np.uint32)
    seg = np.kron(base, np.ones((70, 80), dtype=np.uint32))[:2700, :2800]
    seg[rng.rand(*seg.shape) < 0.01] = 0
    mask = (rng.rand(2300, 2500) > 0.2).astype(np.uint8)
    col = np.arange(40 * 36 + 1, dtype=np.float64) * 1.5
    np.save(tmp_path / 'seg.npy', seg)
    for m in (None, mask):
        want, worig, whist = oracle.subset_recode(seg, 150, 200, 2500, 2300, m, 1024)
        r = subset.subsetImage(str(tmp_path / 'seg.npy'), str(tmp_path / 'out.npy'), 150, 200, 2500, 2300,
                               maskImage=m, ratColumns={'v': col})
        assert np.array_equal(r.segimg, want)
        assert np.array_equal(np.load(tmp_path / 'out.npy'), want)
        assert np.array_equal(r.origSegIds, worig) and np.array_equal(r.hist, whist)
        assert np.array_equal(r.columns['v'][1:], col[worig[1:]])
        assert np.array_equal(r.columns['Histogram'], whist.astype(np.float64))
        # ids are 1..n without gaps, every id used
        assert (whist[1:] > 0).all() and int(want.max()) == len(whist) - 1
