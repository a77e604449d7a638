"""Randomised GPU-vs-oracle parity run (a diagnostic, not a test: needs a GPU and minutes).
Every case draws a raster (dtype, shape, bands, nulls, noise level), a model and the segmentation
parameters, runs the HIP path and the C oracle, and compares bit for bit:
  tile   shepseg.doShepherdSegmentation           vs oracle.segment_tile
  tiled  tiling.doTiledShepherdSegmentation        vs oracle tiles + oracle.stitch_tiles
  stats  tilingstats.calcPerSegmentStats           vs oracle.segstats
  fit    shepseg._fit                             vs oracle.kmeans_fit(algorithm='elkan'): n_iter_, labels_, centres (bit for bit on the Elkan path)
  subset subset.subsetImage                        vs oracle.subset_recode
  spatial tilingstats.calcPerSegmentSpatialStats  vs oracle.spatialstats (edge counts, variogram, mean coordinates)
  spectra shepseg.buildSegmentSpectra / makeSegmentLocations vs oracle
  sharded distributed.runDistributed (HIP engine, one rank) in the sequential AND the parallel form of
         the stitch vs oracle tiles + oracle.stitch_tiles
  paged  tilingstats.calcPerSegmentStatsTiled in small chunks (RAT pages) vs the whole-raster statistics
  big    the same as tile on 1000-2600-pixel rasters with few value levels (components of 10^5-10^6
         pixels: the depth-first cut, its stack spills and the global-memory walk), alternating with
         tiled runs of 1400-3200-pixel rasters (several cluster-map blocks, up to 23 workers)
usage: python tests/fuzz_gpu.py [ncases] [seed] [big|more]     (prints one line per failure and a summary;
       `more` runs the fit / subset / spatial / spectra / paged kinds instead of tile / tiled / stats)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle                                   # noqa: E402
from pyshepseg_amd import shepseg, tiling, tilingstats, subset      # noqa: E402

DTYPES = [np.uint8, np.int16, np.uint16, np.int32, np.uint32]


def make_image(rng, dtype, nb, nr, nc, maxLevels=40):
    """blobby rasters (a few value levels per band + noise) in the dtype's range"""
    base = oracle.synthimg(int(rng.integers(1, 1 << 30)), nb, nr, nc).astype(np.int64)
    levels = int(rng.integers(2, maxLevels))
    img = (base * levels // 65536)
    noise = int(rng.integers(0, 3))
    if noise:
        img = img * (noise + 1) + rng.integers(0, noise + 1, size=img.shape)
    info = np.iinfo(dtype)
    scale = int(rng.choice([1, 7, 250, 30000]))
    img = img * scale
    if info.min < 0 and rng.random() < 0.7:
        img = img - int(img.max()) // 2
    img = np.clip(img, info.min, info.max)
    return np.ascontiguousarray(img.astype(dtype))


def one_case(rng, kind):
    dtype = DTYPES[int(rng.integers(0, len(DTYPES)))]
    nb = int(rng.integers(1, 11))
    shape_kind = rng.random()
    if kind == 'big':
        (nr, nc) = (int(rng.integers(1000, 2600)), int(rng.integers(1000, 2600)))
        nb = int(rng.integers(1, 4))
    elif kind == 'bigtiled':
        (nr, nc) = (int(rng.integers(1400, 3200)), int(rng.integers(1400, 3200)))
        nb = int(rng.integers(1, 7))
    elif kind == 'tiled':
        (nr, nc) = (int(rng.integers(150, 420)), int(rng.integers(150, 420)))
    elif shape_kind < 0.1:
        (nr, nc) = (1, int(rng.integers(1, 500)))
    elif shape_kind < 0.2:
        (nr, nc) = (int(rng.integers(1, 500)), 1)
    else:
        (nr, nc) = (int(rng.integers(2, 400)), int(rng.integers(2, 400)))
    img = make_image(rng, dtype, nb, nr, nc, 5 if kind == 'big' else 12 if kind == 'bigtiled' else 40)
    nullv = None
    if rng.random() < 0.4:
        nullv = int(img.flat[int(rng.integers(0, img.size))]) if rng.random() < 0.5 else int(np.iinfo(dtype).max)
        if rng.random() < 0.5:
            r0 = int(rng.integers(0, nr))
            img[:, r0:r0 + int(rng.integers(1, 8)), :] = nullv
    k = int(rng.integers(2, 6 if kind == 'big' else 25))
    if kind == 'bigtiled':
        k = int(rng.integers(4, 16))
    four = bool(rng.integers(0, 2))
    minseg = int(rng.integers(1, 70))
    xs = shepseg._sample_rows(img, 100, nullv)
    if xs.shape[0] < k:
        return None
    init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
    centres, _l, _n = oracle.kmeans_fit(xs.astype(np.float64), init, max_iter=int(rng.integers(1, 30)))
    km = shepseg.KMeansModel(centres)
    msd = 'auto' if rng.random() < 0.6 else float(rng.choice([0.0, 1.0, 50.0, 1e9]))
    desc = '%s %s nb=%d %dx%d null=%s k=%d four=%d minseg=%d msd=%s' % (
        kind, np.dtype(dtype).name, nb, nr, nc, nullv, k, four, minseg, msd)
    if kind in ('tile', 'big'):
        got = shepseg.doShepherdSegmentation(img, numClusters=k, minSegmentSize=minseg, maxSpectralDiff=msd,
                                             imgNullVal=nullv, fourConnected=four, kmeansObj=km)
        want = oracle.segment_tile(img, centres, minseg, float(got.maxSpectralDiff), nullv, four)
        ok = (np.array_equal(got.segimg, want['segimg']) and
              got.singlePixelsEliminated == want['singlePixelsEliminated'] and
              got.smallSegmentsEliminated == want['smallSegmentsEliminated'])
        if kind == 'big':
            desc += ' clumps=%d' % int(want['numClumps'])
        return ok, desc
    if kind in ('tiled', 'bigtiled'):
        (tile, ov) = [(96, 32), (128, 48), (80, 24), (160, 64)][int(rng.integers(0, 4))]
        if kind == 'bigtiled':          # several cluster-map blocks, many tiles in flight
            (tile, ov) = [(512, 128), (640, 192), (1024, 256), (768, 64)][int(rng.integers(0, 4))]
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS,
                                                   numWorkers=int(rng.integers(1, 6 if kind == 'tiled' else 24)))
        r = tiling.doTiledShepherdSegmentation(img, None, tileSize=tile, overlapSize=ov, minSegmentSize=minseg,
                                               numClusters=k, maxSpectralDiff=msd, imgNullVal=nullv,
                                               fourConnected=four, kmeansObj=km, concurrencyCfg=cfg)
        tiles, ntc, ntr = oracle.get_tiles(nr, nc, tile, ov)
        local = {}
        for (c, rr), (x, y, xsz, ysz) in tiles.items():
            sub = np.ascontiguousarray(img[:, y:y + ysz, x:x + xsz])
            local[(c, rr)] = oracle.segment_tile(sub, centres, minseg, float(r.maxSpectralDiff), nullv, four)['segimg']
        want, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, nr, nc, ov)
        ok = np.array_equal(r.segimg, want) and int(r.maxSegId) == int(mx) and np.array_equal(np.asarray(r.hist), hist)
        return ok, desc + ' tile=%d/%d' % (tile, ov)
    # stats: a segmentation of the image, statistics of one band
    seg = oracle.segment_tile(img, centres, max(minseg, 2), 1e9, nullv, four)['segimg']
    band = np.ascontiguousarray(img[int(rng.integers(0, nb))])
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'), ('f', 'mode'),
           ('g', 'percentile', int(rng.integers(0, 101))), ('h', 'pixcount')]
    mx = int(seg.max())
    ic, fc, _fast = tilingstats.calcPerSegmentStats(seg, band, sel, imgNullVal=nullv, maxSegId=mx)
    wic, wfc = oracle.segstats(seg, band, sel, nullv, -9999, max_seg_id=mx)
    ok = np.array_equal(ic, wic) and np.array_equal(fc.view(np.uint32), wfc.view(np.uint32))
    return ok, desc


def label_raster(rng, nr, nc):
    """scrambled block ids with holes (first-seen order differs from id order)"""
    bh, bw = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    gr, gc = nr // bh + 1, nc // bw + 1
    base = rng.permutation(np.arange(1, gr * gc + 1)).reshape(gr, gc).astype(np.uint32)
    seg = np.kron(base, np.ones((bh, bw), dtype=np.uint32))[:nr, :nc].copy()
    seg[rng.random(seg.shape) < 0.02] = 0
    if rng.random() < 0.5:                       # leave some ids unused
        seg[seg % np.uint32(7) == 3] = 0
    return np.ascontiguousarray(seg)


def more_case(rng, kind, tmpdir):
    dtype = DTYPES[int(rng.integers(0, len(DTYPES)))]
    if kind == 'fit':
        nb = int(rng.integers(1, 11))
        if rng.random() < 0.15:                  # more than 16 bands: several accumulators in the M-step's chain
            nb = int(rng.integers(11, 70))
        n = int(rng.integers(50, 40000))
        k = int(rng.integers(2, 61))
        img = make_image(rng, dtype, nb, 1, n).reshape(nb, n).T.astype(np.float64)
        # real-valued jitter: on integer lattices with few levels exact distance ties decide the
        # trajectory (there the device, the oracle and sklearn's own thread counts
        # legitimately part ways); without ties the three must agree
        img = img + rng.random(img.shape) * float(rng.choice([0.5, 3.0, 40.0]))
        if len(np.unique(img, axis=0)) < k:
            return None
        init = shepseg.diagonalClusterCentres(img, k).astype(np.float64)
        if rng.random() < 0.5:                   # half of the cases on the lattice itself: exact ties, empty clusters
            img = np.floor(img)
            if len(np.unique(img, axis=0)) < k:
                return None
            init = shepseg.diagonalClusterCentres(img, k).astype(np.float64)
        # the reference's algorithm (Elkan's k-means as sklearn 0.24.2 evaluates it, row-order sums)
        want_c, want_l, want_n = oracle.kmeans_fit(img, init, algorithm='elkan')
        km = shepseg._fit(np.ascontiguousarray(img), init)
        # whichever path the tie guard chose (Lloyd iterations or the reference's own algorithm), the M-step
        # adds in the reference's row order: iteration count, labels and centres bit for bit
        ok = (km.n_iter_ == want_n and np.array_equal(km.labels_, want_l) and
              np.array_equal(km.cluster_centers_.view(np.uint64), want_c.view(np.uint64)))
        cdiff = float(np.max(np.abs(km.cluster_centers_ - want_c) / np.maximum(1.0, np.abs(want_c))))
        dump = os.environ.get('SHEPSEG_FUZZ_DUMP')     # keep (a few small) failing cases for a post-mortem on the CPU
        if not ok and dump and img.size < 100000 and len([f for f in os.listdir(dump) if f.startswith('fit_fail_')]) < 6:
            np.savez_compressed(os.path.join(dump, 'fit_fail_%d_%d.npz' % (n, k)), img=img,
                                init=init, dev_centres=km.cluster_centers_, dev_labels=km.labels_, dev_n_iter=km.n_iter_)
        return ok, 'fit %s n=%d nb=%d k=%d path=%s n_iter=%d/%d labels_equal=%s max_rel_centre_diff=%.2e' % (
            np.dtype(dtype).name, n, nb, k, km.fit_path_, km.n_iter_, want_n, np.array_equal(km.labels_, want_l), cdiff)
    if kind == 'sharded':
        from pyshepseg_amd import distributed
        from pyshepseg_amd import comm as shpcomm
        nb = int(rng.integers(1, 7))
        (nr, nc) = (int(rng.integers(150, 520)), int(rng.integers(150, 520)))
        img = make_image(rng, dtype, nb, nr, nc)
        k = int(rng.integers(2, 12))
        four = bool(rng.integers(0, 2))
        minseg = int(rng.integers(2, 60))
        xs = shepseg._sample_rows(img, 100, None)
        init = shepseg.diagonalClusterCentres(xs, k).astype(np.float64)
        centres, _l, _n = oracle.kmeans_fit(xs.astype(np.float64), init, max_iter=10)
        km = shepseg.KMeansModel(centres)
        (tile, ov) = [(96, 32), (64, 32), (80, 24), (160, 64), (48, 32)][int(rng.integers(0, 5))]
        tiles, ntc, ntr = oracle.get_tiles(nr, nc, tile, ov)
        res = {}
        for mode in ('sequential', 'parallel'):
            eng = distributed.HipEngine(lambda yLo, yHi: tiling.DeviceRaster.fromArray(np.ascontiguousarray(img[:, yLo:yHi])),
                                        numWorkers=int(rng.integers(1, 7)), keepOutput=True)
            r = distributed.runDistributed(eng, shpcomm.LocalComm(), nr, nc, tile, ov, minSegmentSize=minseg,
                                           fourConnected=four, kmeansObj=km, stitchMode=mode)
            out = eng.localOutput()
            eng.releaseOutput()
            eng.ras.free()
            res[mode] = (out, int(r.maxSegId), np.asarray(r.hist).copy(), r.stitchMode, float(r.maxSpectralDiff))
        local = {}
        for (c, rr), (x, y, xsz, ysz) in tiles.items():
            sub = np.ascontiguousarray(img[:, y:y + ysz, x:x + xsz])
            local[(c, rr)] = oracle.segment_tile(sub, centres, minseg, res['sequential'][4], None, four)['segimg']
        want, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, nr, nc, ov)
        ok = True
        for mode in ('sequential', 'parallel'):
            (out, m, h, _sm, _msd) = res[mode]
            ok = ok and np.array_equal(out, want) and m == int(mx) and np.array_equal(h, hist)
        return ok, 'sharded %s nb=%d %dx%d tile=%d/%d k=%d minseg=%d -> %s' % (
            np.dtype(dtype).name, nb, nr, nc, tile, ov, k, minseg, res['parallel'][3])
    (nr, nc) = (int(rng.integers(30, 700)), int(rng.integers(30, 700)))
    seg = label_raster(rng, nr, nc)
    S = int(seg.max())
    if kind == 'subset':
        xs, ys = int(rng.integers(1, nc + 1)), int(rng.integers(1, nr + 1))
        tlx, tly = int(rng.integers(0, nc - xs + 1)), int(rng.integers(0, nr - ys + 1))
        m = (rng.random((ys, xs)) > 0.3).astype(np.uint8) if rng.random() < 0.5 else None
        want, worig, whist = oracle.subset_recode(seg, tlx, tly, xs, ys, m, 1024)
        np.save(os.path.join(tmpdir, 'seg.npy'), seg)
        r = subset.subsetImage(os.path.join(tmpdir, 'seg.npy'), os.path.join(tmpdir, 'out.npy'), tlx, tly, xs, ys,
                               maskImage=m)
        ok = (np.array_equal(r.segimg, want) and np.array_equal(r.origSegIds, worig) and
              np.array_equal(r.hist, whist))
        return ok, 'subset %dx%d window (%d,%d,%d,%d) mask=%s' % (nr, nc, tlx, tly, xs, ys, m is not None)
    band = make_image(rng, dtype, 1, nr, nc)[0]
    nullv = int(band.flat[int(rng.integers(0, band.size))])
    if kind == 'spatial':
        ts = tilingstats
        R, I = ts.GFT_Real, ts.GFT_Integer
        which = int(rng.integers(0, 3))
        if which == 0:
            tr = np.array([float(rng.integers(0, 10 ** 6)), float(rng.integers(1, 30)), 0.0,
                           float(rng.integers(0, 10 ** 7)), 0.0, -float(rng.integers(1, 30))])
            _ic, fc = ts.calcPerSegmentSpatialStats(seg, band, [R, R], ts.userFuncMeanCoord, tr, nullv)
            _wi, wf = oracle.spatialstats(seg, band, 'meancoord', tr, nullv, 0, 2, max_seg_id=S)
            ok = bool(np.allclose(fc, wf, rtol=1e-6, atol=0))
        elif which == 1:
            four = bool(rng.integers(0, 2))
            ic, _fc = ts.calcPerSegmentSpatialStats(seg, band, [I, I], ts.userFuncNumEdgePixels, four, nullv)
            wi, _wf = oracle.spatialstats(seg, band, 'numedge', int(four), nullv, 2, 0, max_seg_id=S)
            ok = np.array_equal(ic, wi)
        else:
            md = int(rng.integers(1, 9))
            _ic, fc = ts.calcPerSegmentSpatialStats(seg, band, [R] * md, ts.userFuncVariogram, md, nullv)
            _wi, wf = oracle.spatialstats(seg, band, 'variogram', md, nullv, 0, md, max_seg_id=S)
            ok = np.array_equal(fc.view(np.uint32), wf.view(np.uint32))
        return ok, 'spatial func %d %s %dx%d null=%d' % (which, np.dtype(dtype).name, nr, nc, nullv)
    if kind == 'spectra':
        nb = int(rng.integers(1, 11))
        img = make_image(rng, dtype, nb, nr, nc)
        got = shepseg.buildSegmentSpectra(seg, img, S)
        want = oracle.build_segment_spectra(seg, img, S)
        ok = np.array_equal(np.asarray(got).view(np.uint32), np.asarray(want).view(np.uint32))
        segSize = np.bincount(seg.ravel(), minlength=S + 1).astype(np.uint32)
        loc = shepseg.makeSegmentLocations(seg, segSize)
        (woff, wrc) = oracle.segment_locations(seg, S)
        for sid in rng.integers(1, S + 1, size=min(S, 20)):
            a = np.asarray(loc[np.uint32(sid)].rowcols) if np.uint32(sid) in loc else np.zeros((0, 2), np.uint32)
            b = wrc[woff[sid]:woff[sid + 1]]
            ok = ok and np.array_equal(a.astype(np.int64), np.asarray(b).astype(np.int64))
        return ok, 'spectra %s nb=%d %dx%d S=%d' % (np.dtype(dtype).name, nb, nr, nc, S)
    # paged: the streaming form in small row chunks against the whole-raster statistics
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'), ('f', 'mode'),
           ('g', 'percentile', int(rng.integers(0, 101))), ('h', 'pixcount')]
    use_null = rng.random() < 0.5
    wic, wfc, fast = tilingstats.calcPerSegmentStats(seg, band, sel, imgNullVal=nullv if use_null else None, maxSegId=S)
    r = tilingstats.calcPerSegmentStatsTiled(band, 1, seg, sel, imgNullVal=nullv if use_null else None,
                                             chunkPixels=int(rng.integers(nc, nr * nc + 1)))
    ok = True
    present = np.bincount(seg.ravel(), minlength=S + 1) > 0
    for (i, row) in enumerate(fast):
        name = sel[i][0]
        col = np.asarray(r.columns[name])
        ref = (wic if int(row[2]) == 0 else wfc)[int(row[3])]
        ok = ok and np.array_equal(np.asarray(col)[present][1:] if False else np.asarray(col)[1:][present[1:]],
                                   np.asarray(ref)[1:][present[1:]].astype(col.dtype))
    return ok, 'paged %s %dx%d S=%d null=%s' % (np.dtype(dtype).name, nr, nc, S, use_null)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    more = len(sys.argv) > 3 and sys.argv[3] == 'more'
    import tempfile
    tmpdir = tempfile.mkdtemp()
    rng = np.random.default_rng(seed)
    counts = {'tile': [0, 0], 'tiled': [0, 0], 'stats': [0, 0], 'big': [0, 0], 'bigtiled': [0, 0], 'fit': [0, 0], 'subset': [0, 0],
              'spatial': [0, 0], 'spectra': [0, 0], 'paged': [0, 0], 'sharded': [0, 0]}
    t0 = time.time()
    for i in range(n):
        kind = ('big', 'bigtiled')[i % 2] if big else ('tile', 'tile', 'tiled', 'stats')[i % 4]
        if more:
            kind = ('fit', 'subset', 'spatial', 'spectra', 'paged', 'sharded')[i % 6]
        if len(sys.argv) > 4:
            kind = sys.argv[4]
        try:
            res = more_case(rng, kind, tmpdir) if more else one_case(rng, kind)
        except Exception as e:                      # a raised error is a failure of the case too
            res = (False, '%s raised %r' % (kind, e))
        if res is None:
            continue
        (ok, desc) = res
        counts[kind][0] += 1
        if not ok:
            counts[kind][1] += 1
            print('MISMATCH case %d: %s' % (i, desc), flush=True)
        if (i + 1) % (5 if big else 50) == 0:
            print('  ... %d cases, %.0f s' % (i + 1, time.time() - t0), flush=True)
    print('fuzz_gpu seed %d: ' % seed + ', '.join('%s %d cases / %d mismatches' % (k, v[0], v[1])
                                                  for (k, v) in counts.items() if v[0]) + ' (%.0f s)' % (time.time() - t0))
    return 1 if any(v[1] for v in counts.values()) else 0


if __name__ == '__main__':
    sys.exit(main())
