"""Import the UNMODIFIED reference (from /root/reference) under real numba.

Run only in the build container with /opt/conda/bin/python3.9
(numba 0.54.1, numpy 1.26.4, scikit-learn 0.24.2, scipy 1.7.1).  Nothing in
here, and nothing it imports from /root/reference, ever travels to the GPU
box: the outputs are plain numpy arrays written to tests/golden/.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [_HERE]
import numba_bootstrap  # noqa: E402,F401
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))   # repo root (for `oracle`)

from pyshepseg import shepseg  # noqa: E402

STACK = 'python %s / numba %s / numpy %s / sklearn %s' % (
    sys.version.split()[0], __import__('numba').__version__, __import__('numpy').__version__,
    __import__('sklearn').__version__)


def ref_stages(img, k, min_seg, null_val, four, pcnt=1, km=None, msd='auto', pctile=50):
    """Run the reference stage by stage; return dict of plain arrays."""
    import numpy as np
    if km is None:
        km = shepseg.fitSpectralClusters(img, k, pcnt, null_val, True)
    clusters = shepseg.applySpectralClusters(km, img, null_val)
    seg, nxt = shepseg.clump(clusters, shepseg.SEGNULLVAL, fourConnected=four,
                             clumpId=shepseg.MINSEGID)
    max_seg = shepseg.SegIdType(nxt - 1)
    seg_size = shepseg.makeSegSize(seg)
    seg1 = seg.copy()
    shepseg.eliminateSinglePixels(img, seg1, seg_size.copy(), shepseg.MINSEGID, max_seg, four)
    msd_val = shepseg.autoMaxSpectralDiff(km, msd, pctile)
    seg2 = seg1.copy()
    nelim = shepseg.eliminateSmallSegments(seg2, img, seg1.max(), min_seg, msd_val, four,
                                           shepseg.MINSEGID)
    return dict(centres=np.asarray(km.cluster_centers_, dtype=np.float64),
                clusters=clusters.astype(np.int32), clump=seg, num_clumps=np.int64(max_seg),
                seg_single=seg1, msd=np.float64(msd_val), seg_final=seg2,
                num_single=np.int64(int(max_seg) - int(seg1.max())),
                num_small=np.int64(nelim)), km
