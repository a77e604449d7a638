"""More k-means fixtures from the unmodified reference stack (sklearn 0.24.2).

    /opt/conda/bin/python3.9 oracle/refgen/gen_golden_fit.py

kmeans_fit_c1        C1's sample: synthimg(1, 3, 1024, 1024), 1 % -> 10 486 rows, k = 10
kmeans_fit_10band    a 10-band sample (C4-like), k = 60
kmeans_fit_nulls     a 6-band sample from an image with null pixels (rows dropped, shepseg.py:290-296)
kmeans_predict_ties  KMeans.predict on imagery with few grey levels and on duplicated centres:
                     exact ties, where the label depends on the evaluation order of the reference's
                     E-step (numpy row_norms + BLAS dgemm); 1-, 3-, 6- and 10-band cases
Each fit fixture: sample (pixel dtype), init, centres, labels, n_iter.  Build container only."""
import os

import numpy as np

import refenv
from refenv import shepseg
from oracle import oracle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                   'tests', 'golden')


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in arrays.items()})


def fit_case(name, img, k, pcnt, null_val):
    km = shepseg.fitSpectralClusters(img, k, pcnt, null_val, True)
    nb = img.shape[0]
    x = np.transpose(img, (1, 2, 0)).reshape(-1, nb)
    if null_val is not None:
        x = x[(x != null_val).all(axis=1)]
    xs = x[::int(round(100. / pcnt))]
    init = shepseg.diagonalClusterCentres(xs, k)
    save(name, sample=xs, init=init, centres=km.cluster_centers_, labels=km.labels_.astype(np.int32),
         n_iter=np.int64(km.n_iter_), null_val=np.int64(-1 if null_val is None else null_val))


class FakeKM(object):
    pass


def main():
    fit_case('kmeans_fit_c1', oracle.synthimg(1, 3, 1024, 1024), 10, 1, None)
    fit_case('kmeans_fit_10band', oracle.synthimg(13, 10, 700, 700, 1000, 2000), 60, 2, None)
    img = oracle.synthimg(7, 6, 600, 640, 300, 5000).copy()
    rng = np.random.RandomState(3)
    img[2][rng.rand(600, 640) < 0.07] = 65535
    img[:, :11, :] = 65535
    fit_case('kmeans_fit_nulls', img, 60, 2, 65535)
    # predict with exact ties
    from sklearn.cluster import KMeans
    out = {}
    rng = np.random.RandomState(11)
    for (tag, nb, k, levels) in (('a', 1, 7, 9), ('b', 3, 60, 3), ('c', 6, 60, 4), ('d', 10, 20, 3)):
        img = (rng.randint(0, levels, size=(nb, 70, 90)) * 50 + 10).astype(np.uint16)
        x = np.transpose(img, (1, 2, 0)).reshape(-1, nb).astype(np.float64)
        cen = np.array([x[rng.randint(0, len(x), 3)].mean(axis=0) for _ in range(k)])
        cen[k // 2] = cen[1]                                    # a duplicated centre
        km = KMeans(n_clusters=k, init=cen, n_init=1, max_iter=1)
        km.cluster_centers_ = np.ascontiguousarray(cen)
        km._n_threads = 1
        ref = shepseg.applySpectralClusters(km, img, None)
        out[tag + '_img'] = img
        out[tag + '_centres'] = cen
        out[tag + '_clusters'] = ref.astype(np.int32)
    save('kmeans_predict_ties', **out)


if __name__ == '__main__':
    main()
