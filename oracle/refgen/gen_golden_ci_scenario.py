"""The reference's own CI scenario (pyshepseg/cmdline/runtests.py:63-137) at 1000 x 1000, through the
UNMODIFIED reference: tests/golden/ci_scenario_1000.npz.

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 oracle/refgen/gen_golden_ci_scenario.py

runtests.py needs GDAL files; everything that computes does not.  Data: tests/ci_scenario.py (the
Voronoi palette image of runtests.py:145-265, the 100 centres divided by 8).  Then, with the reference's
own functions: the tiled segmentation (numClusters = 100, fixedKMeansInit, fourConnected = False, null
65535; tiles of 400 with a 100-pixel overlap so that the stitch takes part: gen_golden.stitch_case),
per-band mean / stddev (the njit accumulators of calcPerSegmentStatsTiled, fuzz_stats_vs_reference's
harness), mean coordinates with userFuncMeanCoord (gen_golden_spatial's harness, transform
[0, 1, 0, 0, 0, 1]) and the subset recode of the window (500, 500, 125, 125) (gen_golden_subset's
harness) -- the checks of runtests.py:324-431 scaled by 8.  Output: plain arrays.  Build container only."""
import os
import sys
import numpy as np
import refenv  # noqa: F401
from pyshepseg import tilingstats
import gen_golden
import gen_golden_spatial
import gen_golden_subset
import fuzz_stats_vs_reference

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import ci_scenario  # noqa: E402

assert os.environ.get('OMP_NUM_THREADS') == '1', 'run with OMP_NUM_THREADS=1'
N, SCALE = 1000, 8
trueseg = ci_scenario.true_segments(N, SCALE)
img = ci_scenario.multispectral(trueseg)
name = 'ci_scenario_1000'
gen_golden.stitch_case(name, img, 400, 100, len(ci_scenario.CENTRES), 50, ci_scenario.NULLVAL, False, pcnt=100)
path = os.path.join(gen_golden.OUT, name + '.npz')
g = dict(np.load(path))
g = {k: v for (k, v) in g.items() if not (k.startswith('local_') or k.startswith('recoded_') or k == 'img')}
seg = g['mosaic']
g['trueseg'] = trueseg          # (the image is ci_scenario.multispectral(trueseg): checked by the test)
missing = -9999
for b in range(ci_scenario.NBANDS):
    sel = [('Band_%d_mean' % (b + 1), 'mean'), ('Band_%d_stddev' % (b + 1), 'stddev')]
    (_ic, fc, _sz) = fuzz_stats_vs_reference.reference_stats(seg, img[b], ci_scenario.NULLVAL, sel, missing, 256)
    g['band%d_mean' % (b + 1)] = fc[0]
    g['band%d_stddev' % (b + 1)] = fc[1]
(ic, fc) = gen_golden_spatial.run_reference(seg, img[0], ci_scenario.NULLVAL, 256, tilingstats.userFuncMeanCoord,
                                            np.array([0, 1, 0, 0, 0, 1], dtype=np.float64), 0, 2)
g['meancoord_fc'] = fc
(out, orig, h) = gen_golden_subset.run_reference(seg, 500, 500, 125, 125, None, 1024)
g.update(subset_out=out, subset_orig=orig, subset_hist=h)
np.savez_compressed(path, **g)
print(name, os.path.getsize(path), 'maxSegId', int(g['max_seg_id']), 'n_iter', int(g['n_iter']))
