"""Timings of the whole-image k-means model: device sub-sample, sample preparation, Lloyd fit, in the
row form (host transposition + single-threaded centring) and the band-planar form (threaded
preparation inside the library).  SHEPSEG_FIT_TIMING=1 prints the split inside the library."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyshepseg_amd import tiling, shepseg, _lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # fit 1/shards of the sample's rows: what one rank's
                                                           # share of a row-sharded E-step would cost
ras = tiling.DeviceRaster.synth(11, nb, size, size)
for rep in range(6):
    mode = 'planar' if rep % 2 else 'rows'
    os.environ['SHEPSEG_FIT_PLANAR'] = '1' if mode == 'planar' else '0'
    t0 = time.time()
    img = tiling.readSubsampledImage(ras, list(range(1, nb + 1)), np.sqrt(1e6 / (size * size)))
    if shards > 1:
        img = np.ascontiguousarray(img[:, :img.shape[1] // shards, :])
    t1 = time.time()
    km = shepseg.fitSpectralClusters(img, 60, 100, None, True)
    t2 = time.time()
    print('%-6s subsample %.1f ms  fitSpectralClusters %.1f ms  n_iter %d  rows %d  path %s'
          % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, km.n_iter_, img.shape[1] * img.shape[2], getattr(km, 'fit_path_', '?')))
