import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from pyshepseg_amd import shepseg
for size in (1024, 2048, 4096):
    cl = np.full((size, size), 2, dtype=np.int32)
    shepseg.clump(cl[:64, :64].copy(), 0)
    t = time.time(); seg, nxt = shepseg.clump(cl, 0); dt = time.time() - t
    print('uniform %d x %d: clump %.3f s  (%d pieces, %.0f ns per pixel)' % (size, size, dt, nxt - 1, dt / (size * size) * 1e9), flush=True)
