"""GPU parity tests of the per-tile hot path: HIP (through the C-ABI, via the drop-in Python
API) against the golden vectors of the reference and against the C oracle on seeded inputs.
Bar: bit-exact for every label image and count."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TILE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'tile_*.npz')))
CLUMP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'clump_*.npz')))


@pytest.fixture(scope='module')
def shepseg():
    from pyshepseg_amd import shepseg as m
    from pyshepseg_amd import _lib
    assert _lib.lib().shp_device_count() > 0, 'no GPU: the HIP path cannot run'
    return m


def _null(g):
    return int(g['null_val']) if int(g['has_null']) else None


@pytest.mark.parametrize('name', TILE_CASES)
def test_golden_stages(name, golden, shepseg):
    g = golden(name)
    img, four, null = g['img'], bool(g['four']), _null(g)
    km = shepseg.KMeansModel(g['centres'])
    cl = shepseg.applySpectralClusters(km, img, null)
    assert cl.dtype == np.int32 and np.array_equal(cl, g['clusters'])
    seg, nxt = shepseg.clump(g['clusters'], shepseg.SEGNULLVAL, fourConnected=four,
                             clumpId=shepseg.MINSEGID)
    assert seg.dtype == np.uint32
    assert np.array_equal(seg, g['clump']) and nxt - 1 == int(g['num_clumps'])
    assert np.array_equal(shepseg.makeSegSize(g['clump']), np.bincount(g['clump'].ravel()))
    seg1 = g['clump'].copy()
    shepseg.eliminateSinglePixels(img, seg1, shepseg.makeSegSize(seg1), shepseg.MINSEGID,
                                  int(g['num_clumps']), four)
    assert np.array_equal(seg1, g['seg_single'])
    seg2 = g['seg_single'].copy()
    ne = shepseg.eliminateSmallSegments(seg2, img, int(seg2.max()), int(g['min_seg']),
                                        float(g['msd']), four, shepseg.MINSEGID)
    assert np.array_equal(seg2, g['seg_final']) and ne == int(g['num_small'])


@pytest.mark.parametrize('name', TILE_CASES)
def test_golden_fused(name, golden, shepseg):
    g = golden(name)
    km = shepseg.KMeansModel(g['centres'])
    r = shepseg.doShepherdSegmentation(g['img'], numClusters=int(g['k']),
                                       minSegmentSize=int(g['min_seg']), imgNullVal=_null(g),
                                       fourConnected=bool(g['four']), kmeansObj=km)
    assert np.array_equal(r.segimg, g['seg_final'])
    assert r.maxSpectralDiff == g['msd']
    assert r.singlePixelsEliminated == int(g['num_single'])
    assert r.smallSegmentsEliminated == int(g['num_small'])


@pytest.mark.parametrize('name', CLUMP_CASES)
def test_golden_clump(name, golden, shepseg):
    g = golden(name)
    seg, nxt = shepseg.clump(g['clusters'].astype(np.int32), 0, fourConnected=bool(g['four']))
    assert np.array_equal(seg, g['clump']) and nxt == int(g['next_id'])


def test_synthimg_device_matches_oracle(shepseg, oracle):
    import ctypes
    from pyshepseg_amd import _lib
    c = _lib.ctx()
    out = np.empty((3, 70, 90), dtype=np.uint16)
    c.check(c._L.shp_synthimg(c.handle, 5, 3, 1000, 37, 70, 90, _lib.ptr(out)))
    assert np.array_equal(out, oracle.synthimg(5, 3, 70, 90, y0=1000, x0=37))


@pytest.mark.parametrize('seed,nb,size,k,minseg,four', [
    (1, 3, 512, 10, 20, True),
    (3, 6, 512, 60, 50, True),
    (4, 6, 384, 60, 50, False),
    (1, 3, 1024, 10, 20, True),        # BASELINE config[0] (C1) size
])
def test_oracle_parity_seeded(seed, nb, size, k, minseg, four, shepseg, oracle):
    img = oracle.synthimg(seed, nb, size, size)
    xs = shepseg._sample_rows(img, 1, None)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    centres, _lab, _nit = oracle.kmeans_fit(xs.astype(np.float64), init)
    km = shepseg.KMeansModel(centres)
    msd = float(shepseg.autoMaxSpectralDiff(km, 'auto', 50))
    want = oracle.segment_tile(img, centres, minseg, msd, None, four)
    got = shepseg.doShepherdSegmentation(img, numClusters=k, minSegmentSize=minseg,
                                         fourConnected=four, kmeansObj=km)
    assert np.array_equal(got.segimg, want['segimg'])
    assert got.singlePixelsEliminated == want['singlePixelsEliminated']
    assert got.smallSegmentsEliminated == want['smallSegmentsEliminated']
    # stage-wise at this size too (clump includes cut components)
    cl = shepseg.applySpectralClusters(km, img, None)
    assert np.array_equal(cl, oracle.kmeans_assign(img, centres, None))
    seg, nxt = shepseg.clump(cl, 0, fourConnected=four)
    oseg, onxt = oracle.clump(cl, 0, four, 1)
    assert nxt == onxt and np.array_equal(seg, oseg)


def test_c1_known_counts(shepseg, oracle):
    """BASELINE config[0]: synthimg(1,3,1024,1024), k=10, minSeg=20, fixed init, 1 % sample.
    SURVEY 8(d): 32318 clumps -> 10297 after singles -> 1603 final (reference run)."""
    img = oracle.synthimg(1, 3, 1024, 1024)
    xs = shepseg._sample_rows(img, 1, None)
    centres, _l, _n = oracle.kmeans_fit(xs.astype(np.float64),
                                        shepseg.diagonalClusterCentres(xs, 10).astype(np.float64))
    r = shepseg.doShepherdSegmentation(img, numClusters=10, minSegmentSize=20,
                                       kmeansObj=shepseg.KMeansModel(centres))
    assert int(r.segimg.max()) == 1603
    assert r.singlePixelsEliminated == 32318 - 10297
    assert abs(float(r.maxSpectralDiff) - 809.8255004882812) < 1e-9


@pytest.mark.parametrize('name', ['kmeans_fit_synth512', 'kmeans_fit_c1', 'kmeans_fit_10band',
                                  'kmeans_fit_nulls'])
def test_kmeans_fit_device(name, golden, shepseg, oracle):
    """the device fit against the reference's (one OpenMP thread): iteration count, labels and centres
    bit for bit, whichever path the tie guard chose -- both use the reference's row-order M-step sums"""
    g = golden(name)
    km = shepseg._fit(g['sample'], g['init'])
    assert km.n_iter_ == int(g['n_iter'])
    assert np.array_equal(km.labels_, g['labels'])
    assert np.array_equal(km.cluster_centers_.view(np.uint64), g['centres'].view(np.uint64))
    # run-to-run determinism of the device reduction
    km2 = shepseg._fit(g['sample'], g['init'])
    assert np.array_equal(km.cluster_centers_, km2.cluster_centers_)


def test_predict_exact_ties_device(golden, shepseg):
    """exact ties between two centres: the device E-step follows the reference's evaluation order"""
    g = golden('kmeans_predict_ties')
    for tag in 'abcd':
        km = shepseg.KMeansModel(g[tag + '_centres'])
        got = shepseg.applySpectralClusters(km, g[tag + '_img'], None)
        assert np.array_equal(got, g[tag + '_clusters']), tag


def test_segment_spectra_and_locations_device(golden, shepseg):
    """the public buildSegmentSpectra / makeSegmentLocations run on the device tables of the
    elimination stage (ordered float32 sums, CSR of pixels by segment): bit for bit the reference's"""
    g = golden('spectra_segloc')
    for tag in 'abc':
        seg, img = g[tag + '_seg'], g[tag + '_img']
        S = int(seg.max())
        ss = shepseg.buildSegmentSpectra(seg, img, S)
        assert ss.dtype == np.float32 and ss.shape == g[tag + '_spect_sum'].shape
        assert np.array_equal(ss.view(np.uint32), g[tag + '_spect_sum'].view(np.uint32)), tag
        segSize = shepseg.makeSegSize(seg)
        loc = shepseg.makeSegmentLocations(seg, segSize)
        off, rc = g[tag + '_segloc_off'], g[tag + '_segloc_rc']
        assert sorted(int(k) for k in loc) == list(range(1, S + 1))
        for s in range(1, S + 1):
            (rows, cols) = loc[shepseg.SegIdType(s)].getSegmentIndices()
            want = rc[off[s]:off[s + 1]]
            assert np.array_equal(rows, want[:, 0]) and np.array_equal(cols, want[:, 1]), (tag, s)
            assert (seg[rows, cols] == s).all()


def test_edge_shapes(shepseg, oracle):
    km = shepseg.KMeansModel(np.array([[10., 10.], [200., 200.]]))
    # all-null image
    img = np.full((2, 9, 11), 255, dtype=np.uint8)
    r = shepseg.doShepherdSegmentation(img, imgNullVal=255, kmeansObj=km, minSegmentSize=5)
    assert r.segimg.shape == (9, 11) and not r.segimg.any()
    # 1x1
    img = np.array([[[7]], [[9]]], dtype=np.uint8)
    r = shepseg.doShepherdSegmentation(img, kmeansObj=km, minSegmentSize=5)
    assert r.segimg.tolist() == [[1]]
    # properties on a larger random image: ids contiguous, 4-connected segments, idempotent sizes
    rng = np.random.RandomState(5)
    img = rng.randint(0, 255, size=(2, 200, 300)).astype(np.uint8)
    r = shepseg.doShepherdSegmentation(img, kmeansObj=km, minSegmentSize=10)
    want = oracle.segment_tile(img, km.cluster_centers_, 10, float(r.maxSpectralDiff), None, True)
    assert np.array_equal(r.segimg, want['segimg'])
    ids = np.unique(r.segimg)
    assert ids[0] >= 1 and np.array_equal(ids, np.arange(ids[0], ids[-1] + 1))


@pytest.mark.parametrize('dtype,nb,null,four', [
    (np.uint16, 10, 65535, True),       # C4-like: 10 bands (generic-NB kernels), null border
    (np.int16, 9, -32768, False),       # odd band count -> runtime-NB assign path, 8-connected
    (np.uint8, 3, None, True),
    (np.int32, 2, None, True),
])
def test_oracle_parity_dtypes_bands(dtype, nb, null, four, shepseg, oracle):
    rng = np.random.RandomState(nb)
    base = oracle.synthimg(40 + nb, nb, 300, 340).astype(np.int64)
    if dtype == np.uint8:
        img = (base // 24).astype(np.uint8)
    elif dtype == np.int16:
        img = (base - 3000).astype(np.int16)
    elif dtype == np.int32:
        img = (base * 37 - 50000).astype(np.int32)
    else:
        img = base.astype(np.uint16)
    if null is not None:
        img[:, :5, :] = null
        img[nb // 2][rng.rand(300, 340) < 0.01] = null
    xs = shepseg._sample_rows(img, 2, null)
    init = shepseg.diagonalClusterCentres(xs, 20).astype(np.float64)
    centres, _l, _n = oracle.kmeans_fit(xs.astype(np.float64), init)
    km = shepseg.KMeansModel(centres)
    got = shepseg.doShepherdSegmentation(img, minSegmentSize=30, imgNullVal=null,
                                         fourConnected=four, kmeansObj=km)
    want = oracle.segment_tile(img, centres, 30, float(got.maxSpectralDiff), null, four)
    assert np.array_equal(got.segimg, want['segimg'])
    assert got.singlePixelsEliminated == want['singlePixelsEliminated']
    assert got.smallSegmentsEliminated == want['smallSegmentsEliminated']


def test_default_kmeanspp_path_runs(shepseg, oracle):
    """fixedKMeansInit=False (the reference's default): k-means++ seeding x 5 device fits.  No
    parity definition (unseeded in the reference); the result must be a valid segmentation that
    the oracle reproduces from the model that was chosen."""
    img = oracle.synthimg(77, 3, 200, 220)
    r = shepseg.doShepherdSegmentation(img, numClusters=8, clusterSubsamplePcnt=10, minSegmentSize=15)
    assert r.kmeans.cluster_centers_.shape == (8, 3) and r.kmeans.inertia_ > 0
    want = oracle.segment_tile(img, r.kmeans.cluster_centers_, 15, float(r.maxSpectralDiff), None, True)
    assert np.array_equal(r.segimg, want['segimg'])


def test_many_sources_one_target(shepseg, oracle):
    """Hundreds of equal-size small segments merge into ONE target in the same pass: the sources
    must be absorbed in ascending id (the float32 spectral sums depend on the order), which the
    pass loop gets from a lock-free sorted insert under heavy contention."""
    rng = np.random.RandomState(12)
    nb, nr, nc = 3, 300, 400
    img = np.empty((nb, nr, nc), dtype=np.uint16)
    img[:] = np.array([20000, 30000, 40000], dtype=np.uint16)[:, None, None]       # the background
    blobs = 0
    for y in range(4, nr - 4, 6):
        for x in range(4, nc - 4, 7):
            v = np.array([20000, 30000, 40000]) + rng.randint(3000, 9000, size=nb)    # distinct values
            img[:, y, x:x + 3] = v[:, None].astype(np.uint16)                      # 3-pixel blobs
            blobs += 1
    km = shepseg.KMeansModel(np.array([[20000., 30000., 40000.], [26000., 36000., 46000.]]))
    elim = []
    for minseg, msd in ((10, 1e9), (10, 11000.0)):
        r = shepseg.doShepherdSegmentation(img, kmeansObj=km, minSegmentSize=minseg, maxSpectralDiff=msd)
        want = oracle.segment_tile(img, km.cluster_centers_, minseg, msd, None, True)
        assert np.array_equal(r.segimg, want['segimg'])
        assert r.smallSegmentsEliminated == want['smallSegmentsEliminated']
        elim.append(want['smallSegmentsEliminated'])
    # every blob joins the background when nothing is too different; only some with a threshold
    assert blobs > 2000 and elim[0] >= blobs and 0 < elim[1] < blobs


@pytest.mark.parametrize('four', [True, False])
def test_tiny_min_segment_size_patch_tables(four, shepseg, oracle):
    """minSegmentSize 2 on noise leaves hundreds of segments per 32 x 64 patch: the per-patch
    LDS aggregation tables overflow and their direct-to-global path is taken (stitch prepare),
    odd image sizes exercise the ragged patches of the patch-local CCL."""
    from pyshepseg_amd import tiling
    rng = np.random.RandomState(3)
    img = rng.randint(0, 4, size=(2, 333, 517)).astype(np.uint8) * 60
    km = shepseg.KMeansModel(np.array([[0., 0.], [60., 60.], [120., 120.], [180., 180.]]))
    r = shepseg.doShepherdSegmentation(img, kmeansObj=km, minSegmentSize=2, fourConnected=four)
    want = oracle.segment_tile(img, km.cluster_centers_, 2, float(r.maxSpectralDiff), None, four)
    assert np.array_equal(r.segimg, want['segimg'])
    # the tiled driver on the same image: stitch tables with > 128 segments per patch
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=2)
    t = tiling.doTiledShepherdSegmentation(img, None, tileSize=160, overlapSize=48, minSegmentSize=2,
                                           kmeansObj=km, fourConnected=four, concurrencyCfg=cfg)
    tiles, ntc, ntr = oracle.get_tiles(333, 517, 160, 48)
    local = {}
    for (c, rr), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        local[(c, rr)] = oracle.segment_tile(sub, km.cluster_centers_, 2, float(t.maxSpectralDiff),
                                             None, four)['segimg']
    wt, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, 333, 517, 48)
    assert t.maxSegId == mx and np.array_equal(t.segimg, wt) and np.array_equal(t.hist, hist)


@pytest.mark.parametrize('k,n', [(300, 9000), (7, 5000), (60, 700)])
def test_kmeans_fit_vs_oracle_shapes(k, n, shepseg, oracle):
    """The fit against the oracle's sklearn-0.24.2 restatement: k above the in-LDS counting sort's limit (plain
    partial-sum kernel on the Lloyd path, the generic bounds kernel on the Elkan path), small k, and fewer
    than three chunks of rows."""
    rng = np.random.RandomState(k)
    cent = rng.randint(500, 60000, size=(k, 4))
    xs = (cent[rng.randint(0, k, size=n)] + rng.randint(-200, 200, size=(n, 4))).astype(np.uint16)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    want_c, want_l, want_n = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')   # the reference's
    km = shepseg._fit(xs, init)
    assert km.n_iter_ == want_n
    assert np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64))
    assert np.array_equal(km.labels_, want_l)


@pytest.mark.parametrize('nb,n', [(17, 4000), (33, 2500), (64, 3000), (70, 1500)])
def test_kmeans_fit_many_bands(nb, n, shepseg, oracle):
    """The M-step's row-order sums run as chains of v_mfma_f64_4x4x4 (16 bands per accumulator): more than
    16 bands take two to four accumulators, 64 bands fill the LDS staging buffer to its last run, more than
    64 take the unstaged kernel.  Real-valued rows (no exact ties), the reference's algorithm bit for bit."""
    rng = np.random.RandomState(nb)
    k = 9
    cent = rng.randint(500, 60000, size=(k, nb))
    xs = cent[rng.randint(0, k, size=n)] + rng.randint(-300, 300, size=(n, nb)) + rng.random_sample((n, nb))
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    want_c, want_l, want_n = oracle.kmeans_fit(xs, init, algorithm='elkan')
    km = shepseg._fit(xs, init)
    assert km.n_iter_ == want_n
    assert np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64))
    assert np.array_equal(km.labels_, want_l)


@pytest.mark.parametrize('four', [True, False])
def test_cut_components_both_connectivities(four, shepseg, oracle):
    """Components far above the 10001-pixel cap, 4- and 8-connected: a few large smooth blobs with
    diagonal-only bridges (8-connected merges them, 4-connected does not), so the replay kernel's
    LDS path (both size classes) and its neighbour order are exercised for both connectivities."""
    rng = np.random.RandomState(21)
    nr, nc = 700, 900
    cl = np.ones((nr, nc), dtype=np.int32)
    yy, xx = np.mgrid[0:nr, 0:nc]
    for _i in range(14):
        cy, cx, ry, rx = rng.randint(50, nr - 50), rng.randint(50, nc - 50), rng.randint(40, 160), rng.randint(40, 200)
        cl[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0] = rng.randint(2, 5)
    cl[(yy + xx) % 97 == 0] = 5                         # thin diagonal lines: 8-connected only
    cl[rng.rand(nr, nc) < 0.01] = 0                     # some nulls
    seg, nxt = shepseg.clump(cl, 0, fourConnected=four)
    oseg, onxt = oracle.clump(cl, 0, four, 1)
    assert nxt == onxt and np.array_equal(seg, oseg)
    sizes = np.bincount(oseg.ravel())[1:]
    assert (sizes >= 10001).sum() >= 10                 # many capped pieces


@pytest.mark.parametrize('four', [True, False])
def test_uniform_region_global_replay(four, shepseg, oracle):
    """One component whose bounding box (1200 x 1100) does not fit the 64 KiB LDS bitmap: the
    global-memory replay path, with deep stacks (uniform region) that spill out of the LDS window."""
    cl = np.full((1200, 1100), 3, dtype=np.int32)
    cl[400:420, 300:900] = 0                            # a null bar inside
    seg, nxt = shepseg.clump(cl, 0, fourConnected=four)
    oseg, onxt = oracle.clump(cl, 0, four, 1)
    assert nxt == onxt and np.array_equal(seg, oseg)
    # and one that does fit the LDS bitmap but still spills its stack window
    cl = np.full((500, 900), 2, dtype=np.int32)
    seg, nxt = shepseg.clump(cl, 0, fourConnected=four)
    oseg, onxt = oracle.clump(cl, 0, four, 1)
    assert nxt == onxt and np.array_equal(seg, oseg)


def test_random_shapes_vs_oracle(shepseg, oracle):
    """Forty random small tiles (ragged against the 32 x 64 patches and the 64-pixel wavefront rows
    in every way: 1 x N, N x 1, widths just below / above 64 ...), random band counts, dtypes,
    nulls, connectivity and minimum sizes: fused pipeline and clump stage against the oracle."""
    rng = np.random.RandomState(2024)
    shapes = [(1, 1), (1, 257), (300, 1), (2, 63), (33, 64), (32, 65), (65, 129), (31, 191)]
    while len(shapes) < 40:
        shapes.append((int(rng.randint(1, 220)), int(rng.randint(1, 330))))
    for i, (nr, nc) in enumerate(shapes):
        nb = int(rng.randint(1, 5))
        dtype = [np.uint8, np.uint16, np.int16][i % 3]
        levels = int(rng.randint(2, 6))
        base = rng.randint(0, levels, size=(nb, (nr + 7) // 8, (nc + 7) // 8))
        img = np.kron(base, np.ones((1, 8, 8), dtype=np.int64))[:, :nr, :nc] * 40 + rng.randint(0, 12, size=(nb, nr, nc))
        img = img.astype(dtype)
        null = None
        if i % 4 == 1:
            null = 255 if dtype == np.uint8 else 999
            img[:, rng.rand(nr, nc) < 0.05] = null
        four = bool(i % 2)
        minseg = int(rng.randint(2, 30))
        k = levels
        centres = (np.arange(k, dtype=np.float64)[:, None] * 40 + 6) * np.ones((1, nb))
        km = shepseg.KMeansModel(centres)
        msd = float(rng.choice([15.0, 60.0, 1e6]))
        got = shepseg.doShepherdSegmentation(img, kmeansObj=km, minSegmentSize=minseg, maxSpectralDiff=msd,
                                             imgNullVal=null, fourConnected=four)
        want = oracle.segment_tile(img, centres, minseg, msd, null, four)
        assert np.array_equal(got.segimg, want['segimg']), (i, nr, nc, nb, dtype, null, four, minseg, msd)
        assert got.singlePixelsEliminated == want['singlePixelsEliminated']
        assert got.smallSegmentsEliminated == want['smallSegmentsEliminated']
        cl = oracle.kmeans_assign(img, centres, null)
        seg, nxt = shepseg.clump(cl, 0, fourConnected=four)
        oseg, onxt = oracle.clump(cl, 0, four, 1)
        assert nxt == onxt and np.array_equal(seg, oseg), (i, nr, nc)


def test_ctx_reserve_and_argument_errors(shepseg):
    """shp_ctx_reserve grows the workspace up front; bad arguments come back as errors, not crashes."""
    import ctypes
    from pyshepseg_amd import _lib
    c = _lib.Context()
    try:
        assert c._L.shp_ctx_reserve(c.handle, _lib.SHP_DTYPES[np.dtype(np.uint16)], 6, 1 << 20) == 0
        assert c._L.shp_ctx_reserve(c.handle, 99, 6, 1 << 20) != 0            # unknown dtype
        assert b'bad argument' in c._L.shp_last_error(c.handle)
        assert c._L.shp_ctx_reserve(c.handle, 2, 0, 1 << 20) != 0             # no bands
        assert c._L.shp_ctx_reserve(c.handle, 2, 6, -5) != 0
        # a context that has reserved still segments correctly
        img = (np.arange(3 * 40 * 50).reshape(3, 40, 50) % 7 * 30).astype(np.uint16)
        seg = np.empty((40, 50), np.uint32)
        cen = np.array([[0., 0, 0], [90, 90, 90], [180, 180, 180]])
        mx, s1, s2, ncl = ctypes.c_uint32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_uint32()
        rc = c._L.shp_segment_tile(c.handle, _lib.ptr(img), 2, 3, 40, 50, _lib.ptr(cen), 3, 0, 0, 1, 5,
                                   ctypes.c_double(1e9), _lib.ptr(seg), ctypes.byref(mx), ctypes.byref(s1),
                                   ctypes.byref(s2), ctypes.byref(ncl))
        assert rc == 0 and mx.value == seg.max() and seg.min() >= 1
    finally:
        c.close()


@pytest.mark.parametrize('case', range(8))
def test_planar_fit_equals_row_fit(case, shepseg, oracle):
    """fitSpectralClusters' band-planar form (threaded sample preparation inside the library, no host
    transposition) against the row form: the same rows kept, the same diagonal initial centres, the
    same centres / n_iter_ / labels bit for bit."""
    rng = np.random.RandomState(100 + case)
    dt = [np.uint16, np.uint8, np.int16, np.int32, np.uint32, np.uint16, np.int16, np.uint8][case]
    nb = [6, 3, 4, 2, 5, 1, 10, 7][case]
    (nr, nc) = [(300, 257), (64, 90), (120, 33), (77, 200), (90, 90), (400, 50), (55, 81), (1, 700)][case]
    base = oracle.synthimg(case + 3, nb, nr, nc).astype(np.int64)
    info = np.iinfo(dt)
    span = (int(info.max) - int(info.min)) * 3 // 4
    img = base * span // 65535 + int(info.min)
    img = np.clip(img + rng.randint(0, 5, size=img.shape), info.min, info.max).astype(dt)
    null = None
    if case in (1, 2, 4, 6):
        null = int(info.max) if case != 2 else int(img.flat[17])
        img[rng.randint(0, nb), rng.rand(nr, nc) < 0.07] = null
    k = [60, 7, 12, 5, 30, 9, 20, 4][case]
    import os
    a = shepseg.fitSpectralClusters(img, k, 100, null, True)
    os.environ['SHEPSEG_FIT_PLANAR'] = '0'
    try:
        b = shepseg.fitSpectralClusters(img, k, 100, null, True)
    finally:
        del os.environ['SHEPSEG_FIT_PLANAR']
    assert a.n_iter_ == b.n_iter_
    assert np.array_equal(a.cluster_centers_.view(np.uint64), b.cluster_centers_.view(np.uint64))
    assert np.array_equal(a.labels_, b.labels_)
