"""The reference's k-means model of a benchmark raster (BASELINE configs C3, C4):
tests/golden/c3_fit_reference.npz, tests/golden/c4_fit_reference.npz.

    # on the GPU box: the 1016 x 1016 x 6 sub-sample of synth(11, 6, 40000, 40000) as the tiled driver takes it
    gpurun -- 'bash tools/r2_dump_sample.sh'            # -> gpurun_out/c3_sample.npy
    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_c3_fit.py gpurun_out/c3_sample.npy
    # C4 (10 bands, seed 13): tools/r3_dump_c4_sample.sh -> gpurun_out/c4_sample.npy, then
    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_c3_fit.py gpurun_out/c4_sample.npy c4_fit_reference

Runs the REFERENCE's shepseg.fitSpectralClusters(sample, 60, 100, None, True) (about two minutes: Elkan's
k-means does not converge on this sample and stops at max_iter = 300) and keeps n_iter_, the centres and a
checksum of the sample.  One OpenMP thread: DESIGN.md section 4.  Build container only (refenv.py)."""
import os
import sys
import zlib
import warnings
import numpy as np
warnings.filterwarnings('ignore')
import refenv                                   # noqa: E402
from refenv import shepseg                      # noqa: E402

assert os.environ.get('OMP_NUM_THREADS') == '1', 'run with OMP_NUM_THREADS=1'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
img = np.load(sys.argv[1])
km = shepseg.fitSpectralClusters(img, 60, 100, None, True)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', (sys.argv[2] if len(sys.argv) > 2 else 'c3_fit_reference') + '.npz'),
                    centres=np.asarray(km.cluster_centers_, dtype=np.float64), n_iter=np.int32(km.n_iter_),
                    sample_shape=np.array(img.shape), sample_crc32=np.uint32(zlib.crc32(np.ascontiguousarray(img).tobytes())),
                    stack=np.array(refenv.STACK))
print('n_iter', km.n_iter_, 'sample crc32 %08x' % zlib.crc32(np.ascontiguousarray(img).tobytes()))
