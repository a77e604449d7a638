"""Randomised GPU-vs-oracle parity run (a diagnostic, not a test: needs a GPU and minutes).
Every case draws a raster (dtype, shape, bands, nulls, noise level), a model and the segmentation
parameters, runs the HIP path and the C oracle, and compares bit for bit:
  tile   shepseg.doShepherdSegmentation           vs oracle.segment_tile
  tiled  tiling.doTiledShepherdSegmentation        vs oracle tiles + oracle.stitch_tiles
  stats  tilingstats.calcPerSegmentStats           vs oracle.segstats
  big    the same as tile on 1000-2600-pixel rasters with few value levels (components of 10^5-10^6
         pixels: the depth-first cut, its stack spills and the global-memory walk)
usage: python tools/fuzz_gpu.py [ncases] [seed] [big]     (prints one line per failure and a summary)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle                                   # noqa: E402
from pyshepseg_amd import shepseg, tiling, tilingstats      # noqa: E402

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
    elif kind == 'tiled':
        (nr, nc) = (int(rng.integers(150, 420)), int(rng.integers(150, 420)))
    elif shape_kind < 0.1:
        (nr, nc) = (1, int(rng.integers(1, 500)))
    elif shape_kind < 0.2:
        (nr, nc) = (int(rng.integers(1, 500)), 1)
    else:
        (nr, nc) = (int(rng.integers(2, 400)), int(rng.integers(2, 400)))
    img = make_image(rng, dtype, nb, nr, nc, 5 if kind == 'big' else 40)
    nullv = None
    if rng.random() < 0.4:
        nullv = int(img.flat[int(rng.integers(0, img.size))]) if rng.random() < 0.5 else int(np.iinfo(dtype).max)
        if rng.random() < 0.5:
            r0 = int(rng.integers(0, nr))
            img[:, r0:r0 + int(rng.integers(1, 8)), :] = nullv
    k = int(rng.integers(2, 6 if kind == 'big' else 25))
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
    if kind == 'tiled':
        (tile, ov) = [(96, 32), (128, 48), (80, 24), (160, 64)][int(rng.integers(0, 4))]
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS,
                                                   numWorkers=int(rng.integers(1, 6)))
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    rng = np.random.default_rng(seed)
    counts = {'tile': [0, 0], 'tiled': [0, 0], 'stats': [0, 0], 'big': [0, 0]}
    t0 = time.time()
    for i in range(n):
        kind = 'big' if big else ('tile', 'tile', 'tiled', 'stats')[i % 4]
        try:
            res = one_case(rng, kind)
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
