"""The reference's own CI scenario (pyshepseg/cmdline/runtests.py:63-137) on the GPU path: a Voronoi
palette image with a null border -> doTiledShepherdSegmentation(numClusters=100, fixedKMeansInit,
fourConnected=False) -> per-band mean / stddev -> mean coordinates (userFuncMeanCoord) -> subsetImage
with the new -> old lookup.  It is the only functional test the reference holds.

  * at 1000 x 1000 every result is compared bit for bit with tests/golden/ci_scenario_1000.npz, which the
    UNMODIFIED reference produced for the same data (oracle/refgen/gen_golden_ci_scenario.py);
  * at the reference's own 8000 x 8000 its three checks are asserted as it states them
    (runtests.py:324-431): every pixel's colour within 0.5 of its segment's mean in all bands, mean
    coordinates within 3e-4 of the segments' true centroids, and the subset's lookup column translating
    every new id back to the old one."""
import numpy as np
import pytest

import ci_scenario

pytestmark = pytest.mark.gpu
TRANSFORM = np.array([0, 1, 0, 0, 0, 1], dtype=np.float64)      # eastings / northings = columns / rows


def _run(img, tileSize, overlapSize):
    from pyshepseg_amd import tiling, tilingstats as ts, subset
    r = tiling.doTiledShepherdSegmentation(
        img, None, tileSize=tileSize, overlapSize=overlapSize, numClusters=len(ci_scenario.CENTRES),
        fixedKMeansInit=True, fourConnected=False, imgNullVal=ci_scenario.NULLVAL)
    seg = r.segimg
    means, stds = [], []
    for b in range(ci_scenario.NBANDS):
        sel = [('Band_%d_mean' % (b + 1), 'mean'), ('Band_%d_stddev' % (b + 1), 'stddev')]
        st = ts.calcPerSegmentStatsTiled(img, b + 1, seg, sel, imgNullVal=ci_scenario.NULLVAL)
        means.append(st.columns[sel[0][0]])
        stds.append(st.columns[sel[1][0]])
    sp = ts.calcPerSegmentSpatialStatsTiled(
        img, 1, seg, [('Band_1_easting', ts.GFT_Real), ('Band_1_northing', ts.GFT_Real)],
        ts.userFuncMeanCoord, TRANSFORM, imgNullVal=ci_scenario.NULLVAL)
    return r, seg, means, stds, sp, subset


def test_ci_scenario_1000_equals_reference(golden):
    g = golden('ci_scenario_1000')
    trueseg = ci_scenario.true_segments(1000, 8)
    assert np.array_equal(trueseg, g['trueseg'])                 # the data the reference was given
    img = ci_scenario.multispectral(trueseg)
    r, seg, means, stds, sp, subset = _run(img, int(g['tile_size']), int(g['overlap']))
    assert r.kmeans.n_iter_ == int(g['n_iter'])
    assert np.array_equal(r.kmeans.cluster_centers_.view(np.uint64), g['centres'].view(np.uint64))
    assert r.maxSpectralDiff == g['msd']
    assert (r.numTileCols, r.numTileRows) == (int(g['ntcols']), int(g['ntrows']))
    assert r.maxSegId == int(g['max_seg_id'])
    assert np.array_equal(seg, g['mosaic'])
    assert np.array_equal(np.asarray(r.hist).astype(np.int64), g['hist'].astype(np.int64))
    for b in range(ci_scenario.NBANDS):
        assert np.array_equal(means[b][1:].view(np.uint32), g['band%d_mean' % (b + 1)][1:].view(np.uint32)), b
        assert np.array_equal(stds[b][1:].view(np.uint32), g['band%d_stddev' % (b + 1)][1:].view(np.uint32)), b
    got = np.stack([sp.columns['Band_1_easting'], sp.columns['Band_1_northing']])
    assert np.array_equal(got[:, 1:].view(np.uint32), g['meancoord_fc'][:, 1:].view(np.uint32))
    sub = subset.subsetImage(seg, None, 500, 500, 125, 125, origSegIdColName='orig_val')
    assert np.array_equal(sub.segimg, g['subset_out'])
    assert np.array_equal(sub.origSegIds, g['subset_orig']) and np.array_equal(sub.hist, g['subset_hist'])
    assert np.array_equal(sub.columns['orig_val'], g['subset_orig'].astype(np.int32))


def test_ci_scenario_8000_reference_checks():
    N = 8000
    trueseg = ci_scenario.true_segments(N, 1)
    img = ci_scenario.multispectral(trueseg)
    r, seg, means, stds, sp, subset = _run(img, 4096, 1024)      # the reference's default tiling
    nonNull = seg != 0
    # checkSegmentation (runtests.py:324-376): colours match the segment means, nulls where the image is null
    match = np.ones(seg.shape, dtype=bool)
    for b in range(ci_scenario.NBANDS):
        diff = np.abs(img[b].astype(np.float64) - means[b][seg].astype(np.float64))
        diff[~nonNull] = 0
        match &= diff < 0.5
    assert match.all()
    assert (img[ci_scenario.NBANDS - 1][~nonNull] == ci_scenario.NULLVAL).all()
    assert np.array_equal(nonNull, trueseg != 0)
    # every true cell is recovered up to the size cap's cuts: a segment lies in exactly one cell
    cellOf = np.zeros(int(r.maxSegId) + 1, dtype=np.int64)
    cellOf[seg[nonNull]] = trueseg[nonNull]
    assert np.array_equal(cellOf[seg][nonNull], trueseg[nonNull])
    for b in range(ci_scenario.NBANDS):
        assert (stds[b][1:] == 0).all()                      # one colour per segment
    # checkSpatialColumns (runtests.py:379-410): mean coordinates of every segment within 3e-4
    ids = seg[nonNull].astype(np.int64)
    (rows, cols) = np.nonzero(nonNull)
    cnt = np.bincount(ids, minlength=int(r.maxSegId) + 1).astype(np.float64)
    assert (cnt[1:] > 0).all()
    east = np.bincount(ids, weights=cols, minlength=len(cnt))[1:] / cnt[1:]
    north = np.bincount(ids, weights=rows, minlength=len(cnt))[1:] / cnt[1:]
    assert np.abs(east - sp.columns['Band_1_easting'][1:]).max() <= 3e-4
    assert np.abs(north - sp.columns['Band_1_northing'][1:]).max() <= 3e-4
    # checkSubset (runtests.py:413-431)
    sub = subset.subsetImage(seg, None, 4000, 4000, 1000, 1000, origSegIdColName='orig_val')
    assert sub.segimg.min() == 1
    assert np.array_equal(sub.columns['orig_val'][sub.segimg], seg[4000:5000, 4000:5000])
