"""Pin oracle.segstats against the unmodified reference's statistics loop (accumulateSegDict /
calcStatsForCompletedSegs / RatPage, tilingstats.py:183-206, :466-617, :922-1008) on random label
rasters and bands of every pixel type, values at the types' limits included.

    cd oracle/refgen && /opt/conda/bin/python3.9 fuzz_stats_vs_reference.py [ncases]
"""
import sys

import numpy as np

import refenv
import osgeo  # noqa: F401  (import-only stub next to this script)
from pyshepseg import tilingstats as ts
from oracle import oracle

DTYPES = [np.uint8, np.int16, np.uint16, np.int32, np.uint32]


def reference_stats(seg, band, null_val, sel, missing, tile):
    seg_size = np.bincount(seg.ravel()).astype(np.uint32)
    fast, nint, nflt = ts.makeFastStatsSelection(list(range(len(sel))), sel)
    segDict = ts.createSegDict()
    noData = ts.createNoDataDict()
    paged = ts.createPagedRat()
    nullv = None if null_val is None else ts.numbaTypeForImageType(null_val)
    (nr, nc) = seg.shape
    for y in range(0, nr, tile):
        for x in range(0, nc, tile):
            ts.accumulateSegDict(segDict, noData, nullv, seg[y:y + tile, x:x + tile], band[y:y + tile, x:x + tile])
            ts.calcStatsForCompletedSegs(segDict, noData, missing, paged, fast, seg_size, nint, nflt)
    assert len(segDict) == 0
    ns = len(seg_size)
    ic = np.zeros((nint, ns), dtype=np.int64)
    fc = np.zeros((nflt, ns), dtype=np.float32)
    for pid in paged:
        pg = paged[pid]
        n = pg.intcols.shape[1] if nint else pg.floatcols.shape[1]
        ic[:, pid:pid + n] = pg.intcols
        fc[:, pid:pid + n] = pg.floatcols
    return ic, fc, seg_size


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.RandomState(2024)
    bad = 0
    for case in range(ncases):
        dt = DTYPES[case % 5]
        info = np.iinfo(dt)
        nr, nc = int(rng.randint(8, 90)), int(rng.randint(8, 90))
        bh, bw = int(rng.randint(1, 12)), int(rng.randint(1, 12))
        base = rng.permutation(np.arange(1, (nr // bh + 1) * (nc // bw + 1) + 1)).reshape(nr // bh + 1, nc // bw + 1)
        seg = np.kron(base, np.ones((bh, bw), dtype=np.int64))[:nr, :nc].astype(np.uint32)
        seg[rng.rand(nr, nc) < 0.03] = 0
        kind = (case // 5) % 4
        if kind == 0:
            band = rng.randint(info.min, int(info.max) + 1, size=(nr, nc), dtype=np.int64)
        elif kind == 1:      # the limits and their neighbours
            lv = np.array([info.min, info.min + 1, info.max - 1, info.max, 0], dtype=np.int64)
            band = lv[rng.randint(0, 5, size=(nr, nc))]
        elif kind == 2:      # few values: ties in mode / percentile
            band = rng.randint(0, 4, size=(nr, nc)) * (int(info.max) // 5)
        else:
            band = rng.randint(0, 200, size=(nr, nc)) + int(info.max) - 300
        band = np.clip(band, info.min, info.max).astype(dt)
        null_val = None
        if rng.rand() < 0.5:
            null_val = int(band.flat[rng.randint(0, band.size)]) if rng.rand() < 0.5 else int(info.max)
        sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'), ('f', 'mode'),
               ('g', 'percentile', int(rng.randint(0, 101))), ('h', 'percentile', 0), ('i', 'percentile', 100),
               ('j', 'pixcount')]
        missing = int(rng.choice([-9999, 0, -1]))
        ric, rfc, seg_size = reference_stats(seg, band, null_val, sel, missing, int(rng.choice([16, 64, 1024])))
        oic, ofc = oracle.segstats(seg, band, sel, null_val, missing, max_seg_id=len(seg_size) - 1)
        present = seg_size > 0
        present[0] = False
        ok = np.array_equal(oic[:, present], ric[:, present]) and \
            np.array_equal(ofc[:, present].view(np.uint32), rfc[:, present].view(np.uint32))
        if not ok:
            bad += 1
            print('MISMATCH case %d %s %dx%d kind %d null=%s' % (case, np.dtype(dt).name, nr, nc, kind, null_val))
    print('stack:', refenv.STACK)
    print('DONE: %d cases, %d failures' % (ncases, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
