"""Golden vectors for the fit on tie-heavy samples: tests/golden/fit_ties.npz.

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_fit_ties.py

Inputs: sub-samples of lattice-valued imagery (fuzz_vs_reference.py's generators) on which exact
distance ties or relocations of empty clusters among equally far samples occur, i.e. where Elkan's
algorithm as sklearn 0.24.2 evaluates it and a Lloyd restatement part ways.  Expected outputs: the
REFERENCE's shepseg.fitSpectralClusters(img, k, pcnt, null, fixedKMeansInit=True) -- n_iter_, labels_,
cluster_centers_ -- with one OpenMP thread (the only reproducible setting).
Also a few np.argpartition / row-sum vectors from the reference stack's numpy (1.26.4).
Build container only (refenv.py)."""
import os
import warnings
import numpy as np
warnings.filterwarnings('ignore')
import refenv                                   # noqa: E402
from refenv import shepseg                      # noqa: E402
from oracle import oracle                       # noqa: E402
import fuzz_vs_reference as fz                  # noqa: E402

assert os.environ.get('OMP_NUM_THREADS') == '1', 'run with OMP_NUM_THREADS=1'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rng = np.random.RandomState(777)                # the stream of probe_elkan.py
out = {}
kept = 0
for case in range(400):
    img, null_val = fz.make_img(rng, case) if case % 3 else fz.make_img_wide(rng, case)
    nb, nr, nc = img.shape
    k = int(rng.choice([2, 5, 10, 60]))
    pcnt = int(rng.choice([1, 10, 50, 100]))
    x = np.transpose(img, (1, 2, 0)).reshape(nr * nc, nb)
    if null_val is not None:
        x = x[(x != null_val).all(axis=1)]
    xs = x[::int(round(100. / pcnt))]
    if xs.shape[0] < k or xs.shape[0] > 4000 or kept >= 16:
        continue
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    c_l, l_l, n_l = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='full')
    km = shepseg.fitSpectralClusters(img, k, pcnt, null_val, True)
    rc = np.asarray(km.cluster_centers_, dtype=np.float64)
    if n_l == km.n_iter_ and np.array_equal(l_l, km.labels_) and np.array_equal(c_l, rc):
        continue                                 # no tie decided anything here
    c_e, l_e, n_e = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    assert n_e == km.n_iter_ and np.array_equal(l_e, km.labels_) and np.array_equal(c_e.view(np.uint64), rc.view(np.uint64))
    p = 'c%02d_' % kept
    out[p + 'xs'] = np.ascontiguousarray(xs)
    out[p + 'init'] = init
    out[p + 'n_iter'] = np.int32(km.n_iter_)
    out[p + 'labels'] = np.asarray(km.labels_, dtype=np.int32)
    out[p + 'centres'] = rc
    kept += 1
    print('case %d -> %s: %s nb=%d k=%d n=%d, reference n_iter %d (Lloyd restatement %d)' % (
        case, p, img.dtype.name, nb, k, xs.shape[0], km.n_iter_, n_l))
out['ncases'] = np.int32(kept)
# numpy pieces
r2 = np.random.RandomState(9)
for t in range(12):
    m = int(r2.choice([7, 40, 300, 3000]))
    v = (r2.randint(0, 6, m) * (0.25 if t % 2 else 1.0)).astype(np.float64)
    if t % 4 == 3:
        v = np.sort(v)[::-1].copy()
    ne = int(r2.randint(1, min(m, 30)))
    out['ap%02d_v' % t] = v
    out['ap%02d_ne' % t] = np.int32(ne)
    out['ap%02d_out' % t] = np.argpartition(v, -ne).astype(np.int64)
for t, m in enumerate([1, 5, 7, 8, 9, 16, 31, 127, 128, 129, 300]):
    a = (r2.rand(m) * 1e3) ** 2
    out['ps%02d_a' % t] = a
    out['ps%02d_sum' % t] = np.float64(a.reshape(1, m).sum(axis=1)[0])
out['stack'] = np.array(refenv.STACK)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'fit_ties.npz'), **out)
print('wrote tests/golden/fit_ties.npz: %d fits' % kept)
