"""Re-pins the four small k-means fit fixtures (kmeans_fit_synth512 / _c1 / _10band / _nulls) to the
reference with ONE OpenMP thread: sklearn adds per-thread partial sums of the M-step in the order the
threads finish, so only the one-thread run is reproducible -- the fixtures'
centres came from an 8-thread run and sat a few ulps away from it.  Sample and initial centres are
read from the fixtures themselves and stay; centres, labels and n_iter are rewritten.

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_fit_repin.py

Build container only (imports the unmodified reference stack through refenv.py)."""
import os
import warnings
import numpy as np
warnings.filterwarnings('ignore')
import refenv                                   # noqa: E402,F401
from sklearn.cluster import KMeans              # noqa: E402

assert os.environ.get('OMP_NUM_THREADS') == '1', 'run with OMP_NUM_THREADS=1'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')
for name in ('kmeans_fit_synth512', 'kmeans_fit_c1', 'kmeans_fit_10band', 'kmeans_fit_nulls'):
    path = os.path.join(OUT, name + '.npz')
    g = dict(np.load(path))
    # what shepseg.fitSpectralClusters does with a fixed initialisation (shepseg.py:305-312)
    km = KMeans(n_clusters=len(g['init']), n_init=1, init=g['init']).fit(g['sample'])
    same = (np.array_equal(km.labels_, g['labels']), int(km.n_iter_) == int(g['n_iter']),
            float(np.abs(km.cluster_centers_ - g['centres']).max()))
    g['centres'] = np.asarray(km.cluster_centers_, dtype=np.float64)
    g['labels'] = km.labels_.astype(np.int32)
    g['n_iter'] = np.int64(km.n_iter_)
    g['stack'] = np.array(refenv.STACK + ' / OMP_NUM_THREADS=1')
    np.savez_compressed(path, **g)
    print(name, 'labels same %s, n_iter same %s, centres moved by at most %.3g' % same)
