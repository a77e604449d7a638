"""Timings of the whole-image k-means model: device sub-sample, host sample preparation, Lloyd fit
(set SHEPSEG_FIT_TIMING=1 for the split inside shp_kmeans_fit)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, shepseg, _lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
ras = tiling.DeviceRaster.synth(11, 6, size, size)
for rep in range(3):
    t0 = time.time(); img = tiling.readSubsampledImage(ras, [1, 2, 3, 4, 5, 6], np.sqrt(1e6 / (size * size))); t1 = time.time()
    xs, mm = shepseg._sample_rows(img, 100, None, wantMinMax=True); t2 = time.time()
    init = shepseg.diagonalClusterCentres(xs, 60, mm); t3 = time.time()
    km = shepseg._fit(xs, init); t4 = time.time()
    print('subsample %.1f ms  sample_rows+minmax %.1f ms  diag init %.1f ms  _fit %.1f ms  n_iter %d  rows %d'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, km.n_iter_, xs.shape[0]))
