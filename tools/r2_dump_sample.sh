# dump the benchmark's k-means sub-sample (C3) for a comparison with the reference on the CPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from pyshepseg_amd import tiling, shepseg
ras = tiling.DeviceRaster.synth(11, 6, 40000, 40000)
img = tiling.readSubsampledImage(ras, list(range(1, 7)), np.sqrt(1e6 / (40000 * 40000)))
np.save('gpurun_out/c3_sample.npy', img)
import os
for algo in ('lloyd', 'elkan', 'auto'):
    os.environ['SHEPSEG_FIT_ALGO'] = algo
    km = shepseg.fitSpectralClusters(img, 60, 100, None, True)
    print(algo, km.n_iter_, km.fit_path_, float(km.cluster_centers_.sum()))
    np.save('gpurun_out/c3_centres_%s.npy' % algo, km.cluster_centers_)
PY
