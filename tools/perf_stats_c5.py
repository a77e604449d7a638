"""BASELINE configs C5: per-segment statistics on a 1.6 Gpx label raster with ~50 M segments
(4 x 8-pixel blocks), one image band, stats = mean, stddev, median, pixcount."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, tilingstats, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
BH, BW = 4, 8
ncb = (N + BW - 1) // BW
c = _lib.ctx()
ras = tiling.DeviceRaster.synth(11, 1, N, N)
d_seg = ctypes.c_void_p()
c.check(c._L.shp_dev_alloc(c.handle, N * N * 4, ctypes.byref(d_seg)))
t = time.time()
CH = 2000
colid = (np.arange(N, dtype=np.uint32) // BW)[None, :]
for y0 in range(0, N, CH):
    rows = min(CH, N - y0)
    lab = ((np.arange(y0, y0 + rows, dtype=np.uint32) // BH)[:, None] * np.uint32(ncb) + colid + np.uint32(1))
    lab = np.ascontiguousarray(lab, dtype=np.uint32)
    c.check(c._L.shp_dev_upload(c.handle, ctypes.c_void_p(d_seg.value + y0 * N * 4), _lib.ptr(lab), lab.nbytes))
S = int(((N - 1) // BH) * ncb + (N - 1) // BW + 1)
print('label raster %d x %d, %d segments, built + uploaded in %.1fs' % (N, N, S, time.time() - t))
sel = [('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'), ('n', 'pixcount')]
fast, ni, nf = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)
for rep in range(2):
    ic = np.zeros((ni, S + 1), dtype=np.int64); fc = np.zeros((nf, S + 1), dtype=np.float32)
    t = time.time()
    c.check(c._L.shp_segstats_dev(c.handle, d_seg, ctypes.c_void_p(ras.ptr), 2, N * N, S, 0, 0, _lib.ptr(fast),
                                  len(sel), -9999, _lib.ptr(ic), _lib.ptr(fc)))
    dt = time.time() - t
    print('stats on %d segments, %.0f Mpx: %.3fs  %.0f Mpix/s  %.2e segs/s' % (S, N * N / 1e6, dt, N * N / dt / 1e6, S / dt))
# (parity of this workload against the oracle: tests/test_gpu_fullsize.py::test_c5_stats_fullsize)
assert int(ic[fast[3, 3]].sum()) == N * N
print('pixcount sums to the pixel count')
