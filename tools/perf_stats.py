"""Throughput of the per-segment statistics (tilingstats) on device-resident rasters."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, tilingstats, _lib

size = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ras = tiling.DeviceRaster.synth(11, 6, size, size)
cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=16)
t = time.time()
r = tiling.doTiledShepherdSegmentation(ras, tiling._KEEP_ON_DEVICE, minSegmentSize=50, numClusters=60,
                                       fixedKMeansInit=True, concurrencyCfg=cfg)
print('segmentation %.2fs maxSegId %d' % (time.time() - t, r.maxSegId))
sel = [('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'), ('n', 'pixcount')]
fast, ni, nf = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)
c = _lib.ctx()
npix = size * size
for rep in range(3):
    ic = np.zeros((ni, r.maxSegId + 1), dtype=np.int64); fc = np.zeros((nf, r.maxSegId + 1), dtype=np.float32)
    t = time.time()
    c.check(c._L.shp_segstats_dev(c.handle, ctypes.c_void_p(r.outDev[0]), ctypes.c_void_p(ras.ptr), 2, npix,
                                  r.maxSegId, 0, 0, _lib.ptr(fast), len(sel), -9999, _lib.ptr(ic), _lib.ptr(fc)))
    dt = time.time() - t
    print('stats on %d segments, %.0f Mpx: %.3fs  %.0f Mpix/s  %.2e segs/s' % (r.maxSegId, npix / 1e6, dt, npix / dt / 1e6, r.maxSegId / dt))
assert ic[1].sum() == int((r.hist[1:]).sum()), (ic[1].sum(), r.hist[1:].sum())
print('pixcount column sums to the histogram total: ok')
