"""buildSegmentSpectra / makeSegmentLocations of the unmodified reference (shepseg.py:780-915) on
the label images of existing tile fixtures (after single-pixel elimination, i.e. what
eliminateSmallSegments feeds them, shepseg.py:954-956):

    /opt/conda/bin/python3.9 oracle/refgen/gen_golden_spectra.py

tests/golden/spectra_segloc.npz: per case  img, seg, spect_sum (float32 (S+1, nBands)),
segloc_off (S+2 offsets into segloc_rc, ids 0..S; id 0 is empty: the reference's dict has no
entry for the null segment) and segloc_rc (uint32 (N, 2) row/col pairs in the reference's order).
Cases: a = tile_synth96_4conn, b = tile_f32_inexact (float32 sums beyond 2^24: order matters),
c = tile_many_null (null pixels: row 0 of spect_sum is their sum).  Build container only."""
import os

import numpy as np

import refenv
from refenv import shepseg

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                    'tests', 'golden')


def main():
    out = {}
    for tag, name in (('a', 'tile_synth96_4conn'), ('b', 'tile_f32_inexact'), ('c', 'tile_many_null')):
        with np.load(os.path.join(GOLD, name + '.npz')) as z:
            img = z['img']
            seg = z['seg_single']
        maxSegId = int(seg.max())
        ss = shepseg.buildSegmentSpectra(seg, img, maxSegId)
        segSize = shepseg.makeSegSize(seg)
        loc = shepseg.makeSegmentLocations(seg, segSize)
        offs = np.zeros(maxSegId + 2, dtype=np.uint32)
        rcs = []
        for s in range(1, maxSegId + 1):
            rc = np.asarray(loc[shepseg.SegIdType(s)].rowcols)
            offs[s + 1] = offs[s] + rc.shape[0]
            rcs.append(rc)
        offs[1] = 0
        for s in range(1, maxSegId + 1):
            offs[s + 1] = offs[s] + rcs[s - 1].shape[0]
        out[tag + '_img'] = img
        out[tag + '_seg'] = seg
        out[tag + '_spect_sum'] = np.asarray(ss, dtype=np.float32)
        out[tag + '_segloc_off'] = offs
        out[tag + '_segloc_rc'] = np.concatenate(rcs, axis=0).astype(np.uint32)
        print(tag, name, 'segments', maxSegId, 'spectSum max', float(ss.max()),
              'inexact' if float(ss.max()) > 2 ** 24 else 'exact')
    np.savez_compressed(os.path.join(GOLD, 'spectra_segloc.npz'), **out)


if __name__ == '__main__':
    main()
