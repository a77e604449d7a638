"""End-to-end goldens where the k-means model itself hangs on ties: tests/golden/e2e_ties_{a,b}.npz.

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_e2e_ties.py

Small integer rasters on which Elkan's k-means as the reference runs it and a Lloyd restatement end in
different models (searched for with the oracle's two restatements).  The model is the REFERENCE's
fitSpectralClusters of the whole raster (what fitSpectralClustersWholeFile does for a raster below a
million pixels, tiling.py:154-226), the tiles and the stitch the reference's own functions
(gen_golden.stitch_case's harness).  The test runs doTiledShepherdSegmentation with fixedKMeansInit and NO
kmeansObj and must reproduce centres, n_iter_ and the mosaic bit for bit.  Build container only."""
import os
import numpy as np
import refenv  # noqa: F401
from refenv import shepseg
import gen_golden
from oracle import oracle

assert os.environ.get('OMP_NUM_THREADS') == '1', 'run with OMP_NUM_THREADS=1'
found = 0
for seed in range(1, 400):
    rng = np.random.RandomState(seed)
    nr, nc = int(rng.randint(150, 230)), int(rng.randint(180, 280))
    if seed % 2:      # 8-bit blocks + noise, three bands
        base = rng.randint(0, 200, size=(3, nr // 6 + 1, nc // 6 + 1))
        img = (np.kron(base, np.ones((1, 6, 6), dtype=np.int64))[:, :nr, :nc] + rng.randint(0, 12, size=(3, nr, nc))).astype(np.uint8)
        null_val, four, k = None, True, 12
    else:             # one 16-bit band of a smooth field + noise, a null value, 8-connected
        img = (oracle.synthimg(seed, 1, nr, nc).astype(np.int64) // 64 + rng.randint(0, 6, size=(1, nr, nc))).astype(np.uint16)
        img[0][rng.rand(nr, nc) < 0.02] = 65535
        null_val, four, k = 65535, False, 10
    img = np.ascontiguousarray(img)
    xs = np.transpose(img, (1, 2, 0)).reshape(nr * nc, img.shape[0])
    if null_val is not None:
        xs = xs[(xs != null_val).all(axis=1)]
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    ce, le, ne = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')
    cf, lf, nf = oracle.kmeans_fit(xs.astype(np.float64), init, algorithm='full')
    if np.array_equal(le, lf) and ne == nf:
        continue
    name = 'e2e_ties_' + 'ab'[found]
    want = found % 2
    if (seed % 2) == want:      # one raster of each kind
        continue
    gen_golden.stitch_case(name, img, 96 if seed % 2 else 80, 32 if seed % 2 else 24, k, 10, null_val, four, pcnt=100)
    g = np.load(os.path.join(gen_golden.OUT, name + '.npz'))
    assert np.array_equal(ce, g['centres']) and ne == int(g['n_iter'])       # the Elkan restatement IS the reference here
    print('%s: seed %d, %s %s, reference n_iter %d, Lloyd restatement %d, %d of %d labels differ' % (
        name, seed, img.dtype.name, img.shape, ne, nf, int((le != lf).sum()), len(le)))
    found += 1
    if found == 2:
        break
