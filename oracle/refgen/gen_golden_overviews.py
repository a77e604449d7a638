"""Overview layers and band statistics of the reference's stitch (tiling.py:1343-1404,
utils.py:47-95) on the mosaics of existing stitch fixtures, through the reference's own
setupOverviews / writeOverviews / estimateStatsFromHisto with stand-in GDAL band objects:

    /opt/conda/bin/python3.9 oracle/refgen/gen_golden_overviews.py

tests/golden/overviews_stats.npz:
  levels_<size>        setupOverviews' level list for a raster whose larger side is <size>
  <case>_ov<lvl>       overview arrays of the case's mosaic written tile by tile (levels 2, 4 and 8:
                       the fixtures are far below the 4096 pixels where the reference starts)
  <case>_stats         STATISTICS_* metadata strings of estimateStatsFromHisto(hist), in order
Build container only."""
import os

import numpy as np

import refenv  # noqa: F401
import osgeo  # noqa: F401  (import-only stub next to this script)
from pyshepseg import tiling, utils

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                    'tests', 'golden')


class FakeOvBand(object):
    def __init__(self, ysize, xsize):
        (self.YSize, self.XSize) = (ysize, xsize)
        self.arr = np.zeros((ysize, xsize), dtype=np.uint32)

    def WriteArray(self, a, xoff, yoff):
        self.arr[yoff:yoff + a.shape[0], xoff:xoff + a.shape[1]] = a


class FakeBand(object):
    DataType = 4            # an integer GDAL type (not in gdalFloatTypes)

    def __init__(self, ysize, xsize, levels):
        self.ovs = [FakeOvBand((ysize + lv - 1) // lv, (xsize + lv - 1) // lv) for lv in levels]
        self.meta = []

    def GetOverview(self, j):
        return self.ovs[j]

    def SetMetadataItem(self, k, v):
        self.meta.append((k, v))


class FakeDs(object):
    def BuildOverviews(self, method, levels):
        self.levels = list(levels)


class Mgr(object):
    writeOverviews = tiling.SegmentationConcurrencyMgr.writeOverviews
    setupOverviews = tiling.SegmentationConcurrencyMgr.setupOverviews


def main():
    out = {}
    for size in (1000, 4095, 4096, 4097, 8191, 8192, 20000, 40000, 70000):
        m = Mgr()
        (m.inXsize, m.inYsize) = (size, size // 2)
        ds = FakeDs()
        m.setupOverviews(ds)
        out['levels_%d' % size] = np.array(ds.levels, dtype=np.int64)
    for case in ('stitch_3x4_8conn', 'stitch_3x3_null', 'stitch_2x2'):
        with np.load(os.path.join(GOLD, case + '.npz')) as z:
            mosaic = z['mosaic']
            hist = z['hist']
            (tile, overlap, ntc, ntr) = (int(z['tile_size']), int(z['overlap']), int(z['ntcols']), int(z['ntrows']))
        (nr, nc) = mosaic.shape
        levels = [2, 4, 8]
        m = Mgr()
        m.overviewLevels = levels
        band = FakeBand(nr, nc, levels)
        # the trimmed windows exactly as stitchTiles derives them (tiling.py:997-1022)
        from oracle import oracle
        tiles, ntc2, ntr2 = oracle.get_tiles(nr, nc, tile, overlap)
        assert (ntc2, ntr2) == (ntc, ntr)
        margin = int(overlap / 2)
        for row in range(ntr):
            for col in range(ntc):
                (xpos, ypos, xsize, ysize) = tiles[(col, row)]
                (top, bottom, left, right) = (margin, ysize - margin, margin, xsize - margin)
                (xout, yout) = (xpos + margin, ypos + margin)
                if row == 0:
                    top = 0; yout = ypos
                if row == ntr - 1:
                    bottom = ysize
                if col == 0:
                    left = 0; xout = xpos
                if col == ntc - 1:
                    right = xsize
                trimmed = mosaic[yout:yout + (bottom - top), xout:xout + (right - left)]
                m.writeOverviews(band, trimmed, xout, yout)
        for (j, lv) in enumerate(levels):
            out['%s_ov%d' % (case, lv)] = band.ovs[j].arr
        utils.estimateStatsFromHisto(band, hist)
        out[case + '_stats'] = np.array(['%s=%s' % kv for kv in band.meta])
        print(case, mosaic.shape, [o.arr.shape for o in band.ovs], band.meta[:4])
    np.savez_compressed(os.path.join(GOLD, 'overviews_stats.npz'), **out)
    print({k: v.tolist() for k, v in out.items() if k.startswith('levels_')})


if __name__ == '__main__':
    main()
