"""Where are the zero (unlabelled) pixels of a C4 run, and does the reference's stitch rule explain them?"""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, _lib
from oracle import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
N = 40000
ras = tiling.DeviceRaster.synth(13, 10, N, N)
cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=16)
r = tiling.doTiledShepherdSegmentation(ras, tiling._KEEP_ON_DEVICE, tileSize=4096, overlapSize=1024, minSegmentSize=50,
                                       numClusters=60, fixedKMeansInit=True, concurrencyCfg=cfg)
hist = np.asarray(r.hist).astype(np.int64)
print('maxSegId', r.maxSegId, 'labelled', int(hist.sum()), 'of', N * N, 'n_iter', r.kmeans.n_iter_)
c = _lib.ctx()
zeros = []
rows = 2000
buf = np.empty((rows, N), dtype=np.uint32)
for y0 in range(0, N, rows):
    c.check(c._L.shp_dev_download(c.handle, _lib.ptr(buf), ctypes.c_void_p(r.outDev[0] + y0 * N * 4), buf.nbytes))
    yy, xx = np.nonzero(buf == 0)
    zeros += [(int(y) + y0, int(x)) for y, x in zip(yy, xx)]
print('zero pixels', zeros)
ti = tiling.getTilesForFile(ras, 4096, 1024)
centres = np.ascontiguousarray(r.kmeans.cluster_centers_, dtype=np.float64); msd = float(r.maxSpectralDiff)
import test_gpu_fullsize as T
for (y, x) in zeros[:3]:
    for (col, row), (xp, yp, xs, ys) in ti.tiles.items():
        (top, bottom, left, right, xout, yout) = tiling.trimmedWindow(ti, col, row, xp, yp, xs, ys, 1024)
        if yout <= y < yout + (bottom - top) and xout <= x < xout + (right - left):
            seg, mx, s1, s2, ncl = T._segment_window(ras, xp, yp, xs, ys, centres, 50, msd)
            ly, lx = y - yp, x - xp
            s = seg[ly, lx]
            m = seg == s
            rr, cc = np.nonzero(m)
            print('pixel', (y, x), 'tile', (col, row), 'local', (ly, lx), 'segment', s, 'size', m.sum(), 'rows', rr.min(), rr.max(), 'cols', cc.min(), cc.max(),
                  'trimmed window rows', (top, bottom), 'cols', (left, right))
            crossTop = row > 0 and rr.min() < 512 <= rr.max() and rr.min() < 1024
            crossLeft = col > 0 and cc.min() < 512 <= cc.max() and cc.min() < 1024
            owned = left <= cc[rr == rr.min()].min() * 0 + cc.min() < right and top <= rr.min() < bottom
            print('  crosses top midline', crossTop, 'left midline', crossLeft, ' bbox corner (%d,%d) inside trimmed window: %s' % (rr.min(), cc.min(), owned))
            sub = T._device_window(ras, xp, yp, xs, ys)
            t = time.time(); want = oracle.segment_tile(sub, centres, 50, msd, None, True); print('  oracle tile equal:', np.array_equal(want['segimg'], seg), '%.1fs' % (time.time() - t))
