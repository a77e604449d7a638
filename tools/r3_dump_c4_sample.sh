# dump the k-means sub-sample of the C4 raster (10 bands, seed 13) for oracle/refgen/gen_golden_c3_fit.py,
# then the fit tests (the Lloyd path's M-step now uses the row-order sums)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 300 python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from pyshepseg_amd import tiling, shepseg
ras = tiling.DeviceRaster.synth(13, 10, 40000, 40000)
img = tiling.readSubsampledImage(ras, list(range(1, 11)), np.sqrt(1e6 / (40000 * 40000)))
np.save('gpurun_out/c4_sample.npy', img)
km = shepseg.fitSpectralClusters(img, 60, 100, None, True)
print('c4 fit', km.n_iter_, km.fit_path_, float(km.cluster_centers_.sum()))
np.save('gpurun_out/c4_centres_device.npy', km.cluster_centers_)
PY
timeout -k 10 600 python -m pytest tests/test_fit_elkan.py tests/test_gpu_tile.py -x -q -m gpu -k "fit or kmeans or elkan" > gpurun_out/r3_fit_tests.log 2>&1; tail -15 gpurun_out/r3_fit_tests.log
