"""Stage timings of one tile on the GPU (parity against the oracle lives in tests/: test_gpu_tile.py,
tests/fuzz_gpu.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import shepseg, _lib

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nb, k, minseg = 6, 60, 50
c = _lib.ctx()
img = np.empty((nb, size, size), dtype=np.uint16)
t = time.time(); c.check(c._L.shp_synthimg(c.handle, 4, nb, 0, 0, size, size, _lib.ptr(img))); print('synth %.3fs' % (time.time() - t))
t = time.time(); km = shepseg.fitSpectralClusters(img, k, 1, None, True); print('fit %.3fs n_iter %d' % (time.time() - t, km.n_iter_))
for rep in range(3):
    t = time.time()
    r = shepseg.doShepherdSegmentation(img, minSegmentSize=minseg, kmeansObj=km)
    dt = time.time() - t
    print('rep %d wall %.3fs  %.1f Mpix/s  segs %d  timings(ms) %s' % (rep, dt, size * size / dt / 1e6, r.segimg.max(), {k2: round(v, 2) for k2, v in r.timings.items()}))
