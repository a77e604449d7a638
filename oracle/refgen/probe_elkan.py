"""Pin the oracle's Elkan restatement (orc_kmeans_fit_elkan) against the reference's own fit.

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/probe_elkan.py [ncases]

shepseg.fitSpectralClusters (shepseg.py:305-312) calls KMeans(n_clusters, n_init=1, init=<array>)
with sklearn's default algorithm="auto", which in the pinned stack (sklearn 0.24.2, _kmeans.py:824-825)
is Elkan's variant for k > 1.  For every case of fuzz_vs_reference.py's generator (lattice-valued
imagery: the tie-heavy kind) the reference's fit is compared with the oracle's Lloyd ('full') and
Elkan restatements: n_iter_, labels_ and cluster_centers_ bit for bit.  One OpenMP thread: with more,
sklearn adds per-thread partial sums of the M-step in scheduling order (not reproducible run to run).
Also compares euclidean_distances(centres) / 2 with elk_half_distances' arithmetic (experiment 2).
Build container only (refenv.py); results in results/probe_elkan.txt."""
import sys
import warnings
import numpy as np
warnings.filterwarnings('ignore')
import refenv                                   # noqa: E402
from refenv import shepseg                      # noqa: E402
from oracle import oracle                       # noqa: E402
import fuzz_vs_reference as fz                  # noqa: E402



def partition_key(labels):
    m = {}
    return tuple(m.setdefault(int(v), len(m)) for v in labels)


def sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(777)
n = 0
bad = {'full': [], 'elkan': []}
nperm = 0
for case in range(ncases):
    img, null_val = fz.make_img(rng, case) if case % 3 else fz.make_img_wide(rng, case)
    nb, nr, nc = img.shape
    k = int(rng.choice([2, 5, 10, 60]))
    pcnt = int(rng.choice([1, 10, 50, 100]))
    x = np.transpose(img, (1, 2, 0)).reshape(nr * nc, nb)
    if null_val is not None:
        x = x[(x != null_val).all(axis=1)]
    xs = x[::int(round(100. / pcnt))]
    if xs.shape[0] < k:
        continue
    km = shepseg.fitSpectralClusters(img, k, pcnt, null_val, True)
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    n += 1
    for alg in ('full', 'elkan'):
        c, l, it = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm=alg)
        rc = np.asarray(km.cluster_centers_, dtype=np.float64)
        same = (it == km.n_iter_ and np.array_equal(l, km.labels_) and
                np.array_equal(c.view(np.uint64), rc.view(np.uint64)))
        if not same and alg == 'elkan':
            perm = (it == km.n_iter_ and partition_key(l) == partition_key(km.labels_) and
                    np.array_equal(sorted_rows(c).view(np.uint64), sorted_rows(rc).view(np.uint64)))
            if perm:
                nperm += 1
        if not same:
            bad[alg].append(case)
            if alg == 'elkan':
                print('case %d %s nb=%d k=%d n=%d: elkan oracle n_iter %d / reference %d, same partition %s, max centre diff (sorted) %.3g' % (
                    case, img.dtype.name, nb, k, xs.shape[0], it, km.n_iter_, partition_key(l) == partition_key(km.labels_),
                    float(np.abs(sorted_rows(c) - sorted_rows(rc)).max())))
print(refenv.STACK)
print('%d fits (n_iter_, labels_ and cluster_centers_ bit for bit): oracle Lloyd differs from the reference on %d, oracle Elkan on %d (%d of them only by a permutation of cluster indices)' % (
    n, len(bad['full']), len(bad['elkan']), nperm))
print('Lloyd differs on cases', bad['full'])
print('Elkan differs on cases', bad['elkan'])

# ---- experiments 2-4: the numpy / sklearn pieces restated inside the fit, against the originals ----
import ctypes
from sklearn.metrics.pairwise import euclidean_distances
L = oracle.lib()
rng = np.random.RandomState(3)
nbad = ntot = 0
for trial in range(4000):
    m = int(rng.choice([5, 17, 100, 1000, 20000])) if trial % 10 else int(rng.randint(2, 60))
    kind = trial % 4
    if kind == 0:
        v = rng.rand(m)
    elif kind == 1:
        v = rng.randint(0, 5, m).astype(np.float64)
    elif kind == 2:
        v = np.round(rng.rand(m) * 20) / 4
    else:
        v = np.sort(rng.randint(0, 50, m)).astype(np.float64)[::(-1 if trial % 8 < 4 else 1)].copy()
    ne = int(rng.randint(1, min(m, 40)))
    out = np.empty(m, dtype=np.int64)
    L.orc_np_argpartition(v.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(m), ctypes.c_int64(m - ne),
                          out.ctypes.data_as(ctypes.c_void_p))
    ntot += 1
    nbad += not np.array_equal(out, np.argpartition(v, -ne))
print('np.argpartition(v, -n_empty) on %d arrays with many equal values (2..20000 elements): %d differ from orc_np_argpartition' % (ntot, nbad))
L.orc_np_pairwise_sum.restype = ctypes.c_double
nbad = ntot = 0
for trial in range(3000):
    m = int(rng.randint(1, 300))
    a = np.ascontiguousarray((rng.rand(4, m) * 1e3) ** 2)
    s_np = a.sum(axis=1)
    for r in range(4):
        ntot += 1
        nbad += L.orc_np_pairwise_sum(a[r].ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(m)) != s_np[r]
print('float64 .sum(axis=1) of %d rows of 1..299 elements: %d differ from orc_np_pairwise_sum' % (ntot, nbad))
