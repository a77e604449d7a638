"""The fit as the reference evaluates it (sklearn 0.24.2 picks Elkan's k-means for k > 1) on samples
where exact distance ties and relocations among equally far samples decide the result:
tests/golden/fit_ties.npz = outputs of the reference's fitSpectralClusters with one OpenMP thread
(oracle/refgen/gen_golden_fit_ties.py).  CPU: the oracle's restatement, bit for bit.  GPU: the HIP
fit (its tie guard must send these samples down the Elkan path), bit for bit as well."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ties():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'fit_ties.npz'))


def _case(g, i):
    p = 'c%02d_' % i
    return g[p + 'xs'], g[p + 'init'], int(g[p + 'n_iter']), g[p + 'labels'], g[p + 'centres']


def test_oracle_elkan_equals_reference(ties, oracle):
    differs_from_lloyd = 0
    for i in range(int(ties['ncases'])):
        xs, init, n_iter, labels, centres = _case(ties, i)
        c, l, n = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
        assert n == n_iter, i
        assert np.array_equal(l, labels), i
        assert np.array_equal(c.view(np.uint64), centres.view(np.uint64)), i
        c2, l2, n2 = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='full')
        differs_from_lloyd += not (n2 == n_iter and np.array_equal(l2, labels) and np.array_equal(c2, centres))
    assert differs_from_lloyd == int(ties['ncases'])      # the fixture only holds samples where ties decide


def test_oracle_numpy_pieces(ties, oracle):
    """np.argpartition's introselect and the pairwise row sum, against the reference stack's numpy"""
    L = oracle.lib()
    L.orc_np_pairwise_sum.restype = ctypes.c_double
    t = 0
    while 'ap%02d_v' % t in ties.files:
        v = np.ascontiguousarray(ties['ap%02d_v' % t])
        ne = int(ties['ap%02d_ne' % t])
        out = np.empty(len(v), dtype=np.int64)
        L.orc_np_argpartition(v.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(v)),
                              ctypes.c_int64(len(v) - ne), out.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out, ties['ap%02d_out' % t]), t
        t += 1
    assert t >= 10
    t = 0
    while 'ps%02d_a' % t in ties.files:
        a = np.ascontiguousarray(ties['ps%02d_a' % t])
        got = L.orc_np_pairwise_sum(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(a)))
        assert got == float(ties['ps%02d_sum' % t]), t
        t += 1
    assert t >= 10
    # runs longer than the ufunc buffer: numpy adds the 8192-element blocks' pairwise sums one after the other
    rng = np.random.RandomState(8192)
    for n in (8193, 20000, 200000):
        a = np.ascontiguousarray(rng.random_sample(n) * 1e12)
        assert L.orc_np_pairwise_sum(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == float(a.sum()), n


def test_oracle_elkan_equals_lloyd_without_ties(oracle):
    """well separated real-valued samples: both restatements visit the same partitions"""
    rng = np.random.RandomState(4)
    for (n, nb, k) in [(3000, 6, 20), (500, 1, 7), (2000, 9, 60)]:
        cent = rng.rand(k, nb) * 5000
        x = cent[rng.randint(0, k, n)] + rng.randn(n, nb) * 40
        init = cent + rng.randn(k, nb) * 30
        ce, le, ne = oracle.kmeans_fit(x, init, algorithm='elkan')
        cl, ll, nl = oracle.kmeans_fit(x, init, algorithm='full')
        assert ne == nl and np.array_equal(le, ll)
        assert np.array_equal(ce, cl)           # same partitions, same row-order sums


@pytest.mark.parametrize('name', ['e2e_ties_a', 'e2e_ties_b'])
def test_oracle_end_to_end_model_included(name, golden, oracle):
    """the reference's whole job on rasters where the model hangs on ties (oracle/refgen/gen_golden_e2e_ties.py):
    Elkan fit of the whole raster -> tiles -> stitch, all through the oracle, against the reference's mosaic"""
    from pyshepseg_amd import shepseg as host          # host-side sample selection / diagonal init only
    g = golden(name)
    img = g['img']
    null = int(g['null_val']) if int(g['has_null']) else None
    xs = host._sample_rows(img, 100, null)
    init = host.diagonalClusterCentres(xs, int(g['k'])).astype(np.float64)
    c, _l, nit = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    assert nit == int(g['n_iter']) and np.array_equal(c.view(np.uint64), g['centres'].view(np.uint64))
    c2, _l2, _n2 = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='full')
    assert not np.array_equal(c2, g['centres'])         # the fixture is one where Lloyd's model differs
    tiles, ntc, ntr = oracle.get_tiles(img.shape[1], img.shape[2], int(g['tile_size']), int(g['overlap']))
    local = {}
    for (col, row), (x, y, xs_, ys_) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys_, x:x + xs_])
        local[(col, row)] = oracle.segment_tile(sub, c, int(g['min_seg']), float(g['msd']), null, bool(g['four']))['segimg']
    want, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, img.shape[1], img.shape[2], int(g['overlap']))
    assert mx == int(g['max_seg_id']) and np.array_equal(want, g['mosaic']) and np.array_equal(hist, g['hist'])


# ---- GPU ----------------------------------------------------------------------------------------

@pytest.fixture(scope='module')
def shepseg():
    from pyshepseg_amd import shepseg as m
    from pyshepseg_amd import _lib
    assert _lib.lib().shp_device_count() > 0, 'no GPU: the HIP path cannot run'
    return m


@pytest.mark.gpu
def test_device_fit_equals_reference_on_ties(ties, shepseg):
    for i in range(int(ties['ncases'])):
        xs, init, n_iter, labels, centres = _case(ties, i)
        km = shepseg._fit(xs, init)
        assert km.fit_path_ == 'elkan', i       # a tie decided a label: the guard must have fired
        assert km.n_iter_ == n_iter, i
        assert np.array_equal(km.labels_, labels), i
        assert np.array_equal(km.cluster_centers_.view(np.uint64), centres.view(np.uint64)), i


@pytest.mark.gpu
@pytest.mark.parametrize('n,nb,k', [(20000, 6, 60), (5000, 1, 12), (9000, 10, 30), (700, 3, 300), (130000, 4, 25),
                                    (3000, 70, 5), (64, 2, 64), (4000, 8, 65)])
def test_device_elkan_path_equals_oracle(n, nb, k, shepseg, oracle, monkeypatch):
    """SHEPSEG_FIT_ALGO=elkan on integer samples with a few hundred distinct rows (ties, empty
    clusters): the device's Elkan path against the oracle's, bit for bit; and without the override the
    same answer whichever path the guard chooses"""
    rng = np.random.RandomState(n + k)
    cent = rng.randint(0, 4000, size=(min(k, 40), nb))
    xs = (cent[rng.randint(0, len(cent), size=n)] + rng.randint(-3, 4, size=(n, nb))).astype(np.int16)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    want_c, want_l, want_n = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    monkeypatch.setenv('SHEPSEG_FIT_ALGO', 'elkan')
    km = shepseg._fit(xs, init)
    assert km.fit_path_ == 'elkan'
    assert km.n_iter_ == want_n
    assert np.array_equal(km.labels_, want_l)
    assert np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64))
    # the two other forms of the same path: round 3's exact table of bounds (k > 64 always takes it), and the
    # run that checks, every iteration, the cluster ranges read off the sort's histogram against a label count
    for (var, val) in (('SHEPSEG_ELK_TABLE', '1'), ('SHEPSEG_FIT_CHECK_DIGITS', '1'), ('SHEPSEG_ELK_UNFUSED', '1')):
        monkeypatch.setenv(var, val)
        km1 = shepseg._fit(xs, init)
        monkeypatch.delenv(var)
        assert km1.n_iter_ == want_n and np.array_equal(km1.labels_, want_l), var
        assert np.array_equal(km1.cluster_centers_.view(np.uint64), want_c.view(np.uint64)), var
    monkeypatch.delenv('SHEPSEG_FIT_ALGO')
    km2 = shepseg._fit(xs, init)
    # whichever path the guard chose: the same partitions, the same row-order sums, the same bits
    assert km2.n_iter_ == want_n and np.array_equal(km2.labels_, want_l)
    assert np.array_equal(km2.cluster_centers_.view(np.uint64), want_c.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(4))
def test_device_fast_path_equals_elkan_when_guard_is_quiet(seed, shepseg, oracle, monkeypatch):
    """smooth 16-bit samples: the guard stays quiet, and the fast path's iteration count, labels and
    centres are the reference algorithm's bit for bit (its M-step adds in the reference's row order)"""
    img = oracle.synthimg(40 + seed, 6, 300, 300)
    xs = shepseg._sample_rows(img, 20, None)
    k = (60, 10, 25, 40)[seed]
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    km = shepseg._fit(xs, init)
    ce, le, ne = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    assert km.n_iter_ == ne and np.array_equal(km.labels_, le)
    assert np.array_equal(km.cluster_centers_.view(np.uint64), ce.view(np.uint64))
    monkeypatch.setenv('SHEPSEG_FIT_ALGO', 'elkan')
    km_e = shepseg._fit(xs, init)
    assert km_e.n_iter_ == ne and np.array_equal(km_e.labels_, le)
    assert np.array_equal(km_e.cluster_centers_.view(np.uint64), ce.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize('max_iter', [1, 2, 3, 7])
def test_device_elkan_path_stops_at_max_iter(max_iter, shepseg, oracle, monkeypatch):
    """the iteration limit on the Elkan path: same labels (the final E-step included) and centres as the oracle"""
    rng = np.random.RandomState(max_iter)
    xs = rng.randint(0, 40, size=(6000, 3)).astype(np.uint8)
    init = shepseg.diagonalClusterCentres(xs, 12).astype(np.float64)
    want_c, want_l, want_n = oracle.kmeans_fit(xs.astype(np.float64), init, max_iter=max_iter, algorithm='elkan')
    monkeypatch.setenv('SHEPSEG_FIT_ALGO', 'elkan')
    km = shepseg._fit(xs, init, max_iter=max_iter)
    assert km.n_iter_ == want_n and np.array_equal(km.labels_, want_l)
    assert np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize('shards', [2, 3, 8])
def test_device_sharded_estep_equals_reference(shards, ties, shepseg, oracle, monkeypatch):
    """The E-step sharded by sample rows (the multi-GPU form of the fit: fit_elkan.h FitShard), ONE process playing
    every rank in turn -- each shard its own bounds, the labels in one array as after the all-gather, the count of
    changed labels from the labels themselves: bit for bit the unsharded fit, i.e. the reference's, on the
    tie-decided fixtures and on a lattice sample with empty clusters."""
    monkeypatch.setenv('SHEPSEG_FIT_SHARDS', str(shards))
    for i in range(int(ties['ncases'])):
        xs, init, n_iter, labels, centres = _case(ties, i)
        km = shepseg._fit(xs, init)
        assert km.fit_path_ == 'elkan', i
        assert km.n_iter_ == n_iter, i
        assert np.array_equal(km.labels_, labels), i
        assert np.array_equal(km.cluster_centers_.view(np.uint64), centres.view(np.uint64)), i
    rng = np.random.RandomState(77 + shards)
    cent = rng.randint(0, 4000, size=(40, 6))
    xs = (cent[rng.randint(0, 40, size=50001)] + rng.randint(-3, 4, size=(50001, 6))).astype(np.int16)
    init = shepseg.diagonalClusterCentres(xs, 60).astype(np.float64)
    want_c, want_l, want_n = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    monkeypatch.setenv('SHEPSEG_FIT_ALGO', 'elkan')
    km = shepseg._fit(xs, init)
    assert km.n_iter_ == want_n and np.array_equal(km.labels_, want_l)
    assert np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64))
