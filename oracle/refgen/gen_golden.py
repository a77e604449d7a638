"""Generate tests/golden/*.npz from the UNMODIFIED reference (build container only).

    /opt/conda/bin/python3.9 oracle/refgen/gen_golden.py

Every file holds plain numpy arrays only: the inputs and the reference's
stage-by-stage outputs (no pickled reference objects, no reference source).
Oracle stack recorded in each file under key 'stack'.
"""
import os
import sys

import numpy as np

import refenv
from refenv import shepseg
from oracle import oracle

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                   'tests', 'golden')


def save(name, **arrs):
    arrs['stack'] = np.array(refenv.STACK)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-28s %7.1f kB' % (name, os.path.getsize(path) / 1024.0))


def tile_case(name, img, k, min_seg, null_val, four, pcnt):
    ref, km = refenv.ref_stages(img, k, min_seg, null_val, four, pcnt=pcnt)
    # whole-call cross-check of the staged run
    r = shepseg.doShepherdSegmentation(img, numClusters=k, clusterSubsamplePcnt=pcnt,
                                       minSegmentSize=min_seg, imgNullVal=null_val,
                                       fourConnected=four, fixedKMeansInit=True, kmeansObj=km)
    assert np.array_equal(r.segimg, ref['seg_final'])
    assert r.singlePixelsEliminated == ref['num_single']
    assert r.smallSegmentsEliminated == ref['num_small']
    save(name, img=img, k=np.int64(k), min_seg=np.int64(min_seg),
         null_val=np.int64(-1 if null_val is None else null_val),
         has_null=np.int64(null_val is not None), four=np.int64(four), pcnt=np.int64(pcnt),
         km_n_iter=np.int64(km.n_iter_), km_labels=km.labels_.astype(np.int32), **ref)


def main():
    os.makedirs(OUT, exist_ok=True)
    # (1)/(2) synthetic 3-band 96x96, 4- and 8-connected
    img = oracle.synthimg(1, 3, 96, 96)
    tile_case('tile_synth96_4conn', img, 10, 20, None, True, 10)
    tile_case('tile_synth96_8conn', img, 10, 20, None, False, 10)
    # (3) 6-band 128^2 k=60 minSeg=50 with null border + hole
    img = oracle.synthimg(7, 6, 128, 128, 300, 900)
    img[:, :3, :] = 65535; img[:, -2:, :] = 65535; img[:, :, :4] = 65535; img[:, :, -1:] = 65535
    img[2, 60:75, 40:70] = 65535                      # hole in ONE band only (any-band rule)
    tile_case('tile_synth128_null', img, 60, 50, 65535, True, 5)
    # (4) crafted edge cases
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(3, 40, 50)).astype(np.uint8)
    tile_case('tile_u8_noise', img, 5, 10, None, True, 50)
    img = rng.randint(-500, 500, size=(4, 33, 47)).astype(np.int16)
    tile_case('tile_i16_noise_8conn', img, 5, 10, None, False, 50)
    img = (rng.randint(0, 3, size=(2, 60, 60)) * 700 + 100).astype(np.uint16)   # ties everywhere
    tile_case('tile_ties', img, 4, 30, None, True, 100)
    img = rng.randint(0, 3000, size=(3, 1, 200)).astype(np.uint16)
    tile_case('tile_1xN', img, 5, 5, None, True, 100)
    img = rng.randint(0, 3000, size=(3, 200, 1)).astype(np.uint16)
    tile_case('tile_Nx1', img, 5, 5, None, True, 100)
    img = oracle.synthimg(9, 3, 64, 64)
    img[1, 20, 31] = 65535                            # exactly one null pixel: segSize[0]==1
    tile_case('tile_one_null', img, 10, 20, 65535, True, 10)
    img = oracle.synthimg(11, 3, 48, 48)
    img[0][rng.rand(48, 48) < 0.35] = 65535            # lots of nulls: null is a merge target
    tile_case('tile_many_null', img, 10, 20, 65535, True, 10)
    # (6) float32-inexact spectSum: values near 60000, a > 5000-px flat segment
    img = (60000 + rng.randint(0, 4000, size=(3, 120, 120))).astype(np.uint16)
    img[:, :70, :] = np.array([61001, 60503, 63999], dtype=np.uint16)[:, None, None]
    tile_case('tile_f32_inexact', img, 8, 40, None, True, 20)
    # (5) clump splitting (N9): 150x150 uniform; two-value maze; synthetic with a big component
    cl = np.ones((150, 150), dtype=np.int32)
    for four in (True, False):
        seg, nxt = shepseg.clump(cl, 0, fourConnected=four, clumpId=1)
        save('clump_uniform150_%dconn' % (4 if four else 8), clusters=cl, four=np.int64(four),
             clump=seg, next_id=np.int64(nxt))
    cl = np.ones((180, 170), dtype=np.int32)
    cl[::4, :-3] = 2; cl[2::4, 3:] = 2                # serpentine walls -> long thin component
    cl[50:60, 20:40] = 0
    for four in (True, False):
        seg, nxt = shepseg.clump(cl, 0, fourConnected=four, clumpId=1)
        save('clump_maze_%dconn' % (4 if four else 8), clusters=cl, four=np.int64(four),
             clump=seg, next_id=np.int64(nxt))
    img = oracle.synthimg(1, 3, 256, 256)
    km = shepseg.fitSpectralClusters(img, 4, 1, None, True)
    cl = shepseg.applySpectralClusters(km, img, None).astype(np.int32)
    for four in (True, False):
        seg, nxt = shepseg.clump(cl, 0, fourConnected=four, clumpId=1)
        save('clump_synth256_%dconn' % (4 if four else 8), clusters=cl.astype(np.uint8),
             four=np.int64(four), clump=seg, next_id=np.int64(nxt))
    # (9) k-means fit on a realistic sample (partition / n_iter / centres)
    img = oracle.synthimg(2, 6, 512, 512)
    km = shepseg.fitSpectralClusters(img, 60, 1, None, True)
    x = np.transpose(img, (1, 2, 0)).reshape(-1, 6)[::100]
    init = shepseg.diagonalClusterCentres(x, 60)
    save('kmeans_fit_synth512', sample=x, init=init, centres=km.cluster_centers_,
         labels=km.labels_.astype(np.int32), n_iter=np.int64(km.n_iter_),
         inertia=np.float64(km.inertia_))
    # autoMaxSpectralDiff variants (host code): 'auto' pctiles, None, number
    c = km.cluster_centers_
    save('auto_msd', centres=c,
         p50=np.float64(shepseg.autoMaxSpectralDiff(km, 'auto', 50)),
         p25=np.float64(shepseg.autoMaxSpectralDiff(km, 'auto', 25)),
         p90=np.float64(shepseg.autoMaxSpectralDiff(km, 'auto', 90)),
         none=np.float64(shepseg.autoMaxSpectralDiff(km, None, 50)))
    stitch_cases()
    stats_cases()


def stats_case(name, seg, band, null_val, sel, missing=-9999, tile=64):
    """tilingstats.calcPerSegmentStatsTiled's compute loop (tilingstats.py:183-206) driven
    through the reference's accumulateSegDict / calcStatsForCompletedSegs / RatPage, on arrays."""
    import osgeo  # noqa: F401  (import-only stub next to this script)
    from pyshepseg import tilingstats as ts
    seg_size = np.bincount(seg.ravel()).astype(np.uint32)
    fast, nint, nflt = ts.makeFastStatsSelection(list(range(len(sel))), sel)
    segDict = ts.createSegDict()
    noData = ts.createNoDataDict()
    paged = ts.createPagedRat()
    nullv = None if null_val is None else ts.numbaTypeForImageType(null_val)
    (nr, nc) = seg.shape
    for y in range(0, nr, tile):
        for x in range(0, nc, tile):
            ts.accumulateSegDict(segDict, noData, nullv, seg[y:y + tile, x:x + tile],
                                 band[y:y + tile, x:x + tile])
            ts.calcStatsForCompletedSegs(segDict, noData, missing, paged, fast, seg_size, nint, nflt)
    assert len(segDict) == 0
    ns = len(seg_size)
    ic = np.zeros((nint, ns), dtype=np.int64)
    fc = np.zeros((nflt, ns), dtype=np.float32)
    done = np.zeros(ns, dtype=bool)
    for pid in paged:
        pg = paged[pid]
        n = pg.intcols.shape[1] if nint else pg.floatcols.shape[1]
        ic[:, pid:pid + n] = pg.intcols
        fc[:, pid:pid + n] = pg.floatcols
        done[pid:pid + n] = pg.complete
    save(name, seg=seg, band=band, null_val=np.int64(-1 if null_val is None else null_val),
         has_null=np.int64(null_val is not None), missing=np.int64(missing),
         sel_names=np.array([t[1] for t in sel]),
         sel_params=np.array([t[2] if len(t) > 2 else -1 for t in sel], dtype=np.int64),
         intcols=ic, floatcols=fc, complete=done)


def stats_cases():
    g = np.load(os.path.join(OUT, 'stitch_3x3_null.npz'))
    seg = g['mosaic']
    img = g['img']
    sel = [('mn', 'min'), ('mx', 'max'), ('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'),
           ('mode', 'mode'), ('p0', 'percentile', 0), ('p25', 'percentile', 25),
           ('p75', 'percentile', 75), ('p100', 'percentile', 100), ('n', 'pixcount')]
    stats_case('stats_u16_nonull', seg, img[0], None, sel)
    band = img[1].copy()
    band[seg == 7] = 65535                      # one segment entirely nodata
    stats_case('stats_u16_null', seg, band, 65535, sel)
    rng = np.random.RandomState(8)
    segb = (np.arange(96 * 120).reshape(96, 120) // 7 % 40 + 1).astype(np.uint32)
    segb[:5] = 0
    bandi = rng.randint(-300, 300, size=segb.shape).astype(np.int16)
    stats_case('stats_i16_ties', segb, bandi, -7, [('mode', 'mode'), ('med', 'median'),
                                                    ('mean', 'mean'), ('sd', 'stddev'),
                                                    ('n', 'pixcount'), ('p10', 'percentile', 10)])
    big = (rng.randint(0, 4, size=(200, 200)) + 1).astype(np.uint32)    # 4 segments x ~10k px
    bandb = rng.randint(60000, 65535, size=big.shape).astype(np.uint16)
    stats_case('stats_big_segments', big, bandb, None, [('mean', 'mean'), ('sd', 'stddev'),
                                                        ('med', 'median'), ('mode', 'mode')])


def stitch_case(name, img, tile_size, overlap, k, min_seg, null_val, four, pcnt=5):
    """tiling.stitchTiles loop (tiling.py:979-1043) driven through the reference's own
    recodeTile / recodeSharedSegments / relabelSegments / crossesMidline, in memory."""
    from pyshepseg import tiling

    class FakeDs(object):
        RasterXSize = img.shape[2]
        RasterYSize = img.shape[1]

    class FakeMgr(object):
        pass
    ti = tiling.getTilesForFile(FakeDs(), tile_size, overlap)
    km = shepseg.fitSpectralClusters(img, k, pcnt, null_val, True)
    cache = {}
    mgr = FakeMgr()
    mgr.overlapSize = overlap
    mgr.overlapCacheKey = lambda col, row, edge: '{}_{}_{}'.format(edge, col, row)
    mgr.loadOverlap = lambda key: cache[key]
    mgr.recodeSharedSegments = tiling.SegmentationConcurrencyMgr.recodeSharedSegments
    mgr.relabelSegments = tiling.SegmentationConcurrencyMgr.relabelSegments
    margin = int(overlap / 2)
    out = np.zeros(img.shape[1:], dtype=np.uint32)
    hist = tiling.HistogramAccumulator()
    max_seg = 0
    arrs = {}
    msd = None
    for (col, row) in sorted(ti.tiles.keys(), key=lambda x: (x[1], x[0])):
        (xpos, ypos, xsize, ysize) = ti.getTile(col, row)
        sub = np.ascontiguousarray(img[:, ypos:ypos + ysize, xpos:xpos + xsize])
        r = shepseg.doShepherdSegmentation(sub, minSegmentSize=min_seg, imgNullVal=null_val,
                                           fourConnected=four, kmeansObj=km)
        msd = r.maxSpectralDiff
        tileData = r.segimg
        arrs['local_%d_%d' % (col, row)] = tileData.copy()
        top, bottom, left, right = margin, ysize - margin, margin, xsize - margin
        xout, yout = xpos + margin, ypos + margin
        rightName = mgr.overlapCacheKey(col, row, tiling.RIGHT_OVERLAP)
        bottomName = mgr.overlapCacheKey(col, row, tiling.BOTTOM_OVERLAP)
        if row == 0:
            top = 0; yout = ypos
        if row == ti.nrows - 1:
            bottom = ysize; bottomName = None
        if col == 0:
            left = 0; xout = xpos
        if col == ti.ncols - 1:
            right = xsize; rightName = None
        tileData = tiling.SegmentationConcurrencyMgr.recodeTile(mgr, tileData, max_seg, row, col,
                                                                top, bottom, left, right)
        arrs['recoded_%d_%d' % (col, row)] = tileData.copy()
        trimmed = tileData[top:bottom, left:right]
        out[yout:yout + trimmed.shape[0], xout:xout + trimmed.shape[1]] = trimmed
        hist.doHistAccum(trimmed)
        if rightName is not None:
            cache[rightName] = tileData[:, -overlap:].copy()
        if bottomName is not None:
            cache[bottomName] = tileData[-overlap:, :].copy()
        max_seg = max(max_seg, trimmed.max())
    save(name, img=img, tile_size=np.int64(tile_size), overlap=np.int64(overlap), k=np.int64(k),
         min_seg=np.int64(min_seg), null_val=np.int64(-1 if null_val is None else null_val),
         has_null=np.int64(null_val is not None), four=np.int64(four),
         centres=np.asarray(km.cluster_centers_, dtype=np.float64), msd=np.float64(msd),
         n_iter=np.int64(km.n_iter_), pcnt=np.int64(pcnt),
         ntcols=np.int64(ti.ncols), ntrows=np.int64(ti.nrows), mosaic=out,
         max_seg_id=np.int64(max_seg), hist=hist.hist.astype(np.uint32), **arrs)
    print('   tiles %dx%d maxSegId %d empty ids %d' % (ti.ncols, ti.nrows, max_seg,
                                                        int((hist.hist[1:] == 0).sum())))


def stitch_cases():
    img = oracle.synthimg(21, 3, 200, 200)
    stitch_case('stitch_2x2', img, 96, 32, 8, 12, None, True)
    img = oracle.synthimg(22, 3, 300, 280, 50, 70)
    img[:, :6, :] = 65535
    img[:, :, -5:] = 65535
    img[1, 140:170, 100:160] = 65535
    stitch_case('stitch_3x3_null', img, 96, 32, 8, 15, 65535, True)
    img = oracle.synthimg(23, 3, 260, 330)
    stitch_case('stitch_3x4_8conn', img, 80, 24, 6, 10, None, False)


if __name__ == '__main__':
    sys.exit(main())
