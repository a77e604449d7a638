import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, shepseg, _lib
ras = tiling.DeviceRaster.synth(11, 6, 20000, 20000)
for rep in range(2):
    t0 = time.time(); img = tiling.readSubsampledImage(ras, [1,2,3,4,5,6], np.sqrt(1e6/(4e8))); t1 = time.time()
    xs = shepseg._sample_rows(img, 100, None); t2 = time.time()
    init = shepseg.diagonalClusterCentres(xs, 60); t3 = time.time()
    x = np.ascontiguousarray(xs, dtype=np.float64); t4 = time.time()
    km = shepseg._fit(xs, init); t5 = time.time()
    print('subsample %.3f sample_rows %.3f diag %.3f tofloat %.3f fit %.3f n_iter %d shape %s' % (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, km.n_iter_, xs.shape))
