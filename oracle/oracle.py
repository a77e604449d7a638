"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is a plain-C CPU restatement of the pyshepseg hot path
(oracle/shepseg_oracle.c).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package
pyshepseg_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SHEPSEG_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle asan`)
_LIBPATH = os.environ.get('SHEPSEG_ORACLE_LIB') or os.path.join(_HERE, 'liboracle.so')

DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.uint16): 2,
          np.dtype(np.int32): 3, np.dtype(np.uint32): 4}


def build(force=False):
    src = os.path.join(_HERE, 'shepseg_oracle.c')
    if os.environ.get('SHEPSEG_ORACLE_LIB'):
        return
    if (force or not os.path.exists(_LIBPATH) or
            os.path.getmtime(_LIBPATH) < os.path.getmtime(src)):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            build()
        _lib = ctypes.CDLL(_LIBPATH)
        _lib.orc_clump.restype = ctypes.c_uint32
        _lib.orc_seg_max.restype = ctypes.c_uint32
        _lib.orc_eliminate_single_pixels.restype = ctypes.c_int64
        _lib.orc_eliminate_small_segments.restype = ctypes.c_int64
        _lib.orc_recode_tile.restype = ctypes.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _img(img):
    img = np.ascontiguousarray(img)
    if img.dtype not in DTYPES:
        raise TypeError('unsupported image dtype %s' % img.dtype)
    return img, DTYPES[img.dtype]


def synthimg(seed, nbands, rows, cols, y0=0, x0=0):
    out = np.empty((nbands, rows, cols), dtype=np.uint16)
    lib().orc_synthimg(ctypes.c_uint64(seed), nbands, ctypes.c_int64(y0), ctypes.c_int64(x0),
                       rows, cols, _p(out))
    return out


def kmeans_assign(img, centres, null_val=None):
    img, dt = _img(img)
    nb, nr, nc = img.shape
    centres = np.ascontiguousarray(centres, dtype=np.float64)
    out = np.empty((nr, nc), dtype=np.int32)
    rc = lib().orc_kmeans_assign(_p(img), dt, nb, nr, nc, _p(centres), centres.shape[0],
                                 int(null_val is not None),
                                 ctypes.c_int64(0 if null_val is None else int(null_val)), _p(out))
    assert rc == 0
    return out


def clump(clusters, ignore_val=0, four_connected=True, clump_id=1):
    clusters = np.ascontiguousarray(clusters, dtype=np.int32)
    nr, nc = clusters.shape
    out = np.empty((nr, nc), dtype=np.uint32)
    nxt = lib().orc_clump(_p(clusters), nr, nc, int(ignore_val), int(four_connected),
                          ctypes.c_uint32(clump_id), _p(out))
    return out, int(nxt)


def make_seg_size(seg):
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    m = int(seg.max()) if seg.size else 0
    out = np.empty(m + 1, dtype=np.uint32)
    lib().orc_make_seg_size(_p(seg), ctypes.c_size_t(seg.size), ctypes.c_uint32(m), _p(out))
    return out


def eliminate_single_pixels(img, seg, seg_size, min_seg_id, max_seg_id, four_connected):
    """In place on seg / seg_size, like shepseg.eliminateSinglePixels."""
    img, dt = _img(img)
    nb, nr, nc = img.shape
    assert seg.dtype == np.uint32 and seg.flags.c_contiguous
    assert seg_size.dtype == np.uint32 and seg_size.flags.c_contiguous
    return int(lib().orc_eliminate_single_pixels(_p(img), dt, nb, nr, nc, _p(seg), _p(seg_size),
                                                 ctypes.c_uint32(min_seg_id),
                                                 ctypes.c_uint32(max_seg_id), int(four_connected)))


def eliminate_small_segments(seg, img, max_seg_id, min_seg_size, max_spectral_diff,
                             four_connected, min_seg_id=1):
    """In place on seg, like shepseg.eliminateSmallSegments. Returns numElim."""
    img, dt = _img(img)
    nb, nr, nc = img.shape
    assert seg.dtype == np.uint32 and seg.flags.c_contiguous
    return int(lib().orc_eliminate_small_segments(_p(seg), _p(img), dt, nb, nr, nc,
                                                  ctypes.c_uint32(max_seg_id), int(min_seg_size),
                                                  ctypes.c_double(max_spectral_diff),
                                                  int(four_connected), ctypes.c_uint32(min_seg_id)))


def build_segment_spectra(seg, img, max_seg_id):
    """shepseg.buildSegmentSpectra: float32 (max_seg_id + 1, nBands)"""
    img, dt = _img(img)
    nb, nr, nc = img.shape
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    out = np.zeros((int(max_seg_id) + 1, nb), dtype=np.float32)
    lib().orc_build_segment_spectra(_p(seg), _p(img), dt, nb, nr, nc, ctypes.c_uint32(int(max_seg_id)), _p(out))
    return out


def segment_locations(seg, max_seg_id):
    """shepseg.makeSegmentLocations as (offsets (max_seg_id + 2,), rowcols (N, 2))"""
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    nr, nc = seg.shape
    off = np.zeros(int(max_seg_id) + 2, dtype=np.uint32)
    rc = np.zeros((int((seg != 0).sum()), 2), dtype=np.uint32)
    lib().orc_segment_locations(_p(seg), nr, nc, ctypes.c_uint32(int(max_seg_id)), _p(off), _p(rc))
    return off, rc


def segment_tile(img, centres, min_seg_size, max_spectral_diff, null_val=None, four_connected=True):
    img, dt = _img(img)
    nb, nr, nc = img.shape
    centres = np.ascontiguousarray(centres, dtype=np.float64)
    seg = np.empty((nr, nc), dtype=np.uint32)
    mx = ctypes.c_uint32(0)
    s1 = ctypes.c_int64(0)
    s2 = ctypes.c_int64(0)
    ncl = ctypes.c_uint32(0)
    rc = lib().orc_segment_tile(_p(img), dt, nb, nr, nc, _p(centres), centres.shape[0],
                                int(null_val is not None),
                                ctypes.c_int64(0 if null_val is None else int(null_val)),
                                int(four_connected), int(min_seg_size),
                                ctypes.c_double(max_spectral_diff), _p(seg), ctypes.byref(mx),
                                ctypes.byref(s1), ctypes.byref(s2), ctypes.byref(ncl))
    assert rc == 0
    return dict(segimg=seg, maxSegId=mx.value, singlePixelsEliminated=s1.value,
                smallSegmentsEliminated=s2.value, numClumps=ncl.value)


def kmeans_fit(xsample, init, max_iter=300, tol=1e-4, mstep='rows', algorithm='full'):
    """algorithm='full': Lloyd; 'elkan': what sklearn 0.24.2's KMeans(algorithm='auto') runs for k > 1.
    mstep='rows': sklearn's one-thread summation order (what the HIP fit uses on both of its paths);
    'device': chunked sums (min(256, 4096 // nBands) rows, groups of 64 chunks) -- the association an
    earlier HIP fit had, kept to show what a different order does to the centres"""
    x = np.ascontiguousarray(xsample, dtype=np.float64)
    init = np.ascontiguousarray(init, dtype=np.float64)
    n, nb = x.shape
    k = init.shape[0]
    centres = np.empty((k, nb), dtype=np.float64)
    labels = np.empty(n, dtype=np.int32)
    nit = ctypes.c_int(0)
    (chunk, group) = (min(256, 4096 // nb), 64) if mstep == 'device' else (0, 0)
    fn = lib().orc_kmeans_fit_elkan if (algorithm == 'elkan' and k > 1) else lib().orc_kmeans_fit_assoc
    rc = fn(_p(x), ctypes.c_int64(n), nb, k, _p(init), int(max_iter), ctypes.c_double(tol), chunk, group,
            _p(centres), _p(labels), ctypes.byref(nit))
    assert rc == 0
    return centres, labels, nit.value


def recode_tile(tile, overlap, top_b, left_b, max_seg_id, top, bottom, left, right):
    """tiling.recodeTile for one tile: returns the recoded copy.  top_b / left_b: the saved
    (recoded) bottom strip of the tile above / right strip of the tile to the left, or None."""
    tile = np.ascontiguousarray(tile, dtype=np.uint32)
    ys, xs = tile.shape
    out = np.empty_like(tile)
    tb = lb = None
    tp = lp = 0
    if top_b is not None:
        tb = np.ascontiguousarray(top_b, dtype=np.uint32); tp = tb.shape[1]
    if left_b is not None:
        lb = np.ascontiguousarray(left_b, dtype=np.uint32); lp = lb.shape[1]
    lib().orc_recode_tile(_p(tile), ys, xs, int(overlap),
                          _p(tb) if tb is not None else None, ctypes.c_size_t(tp),
                          _p(lb) if lb is not None else None, ctypes.c_size_t(lp),
                          ctypes.c_uint32(int(max_seg_id)), int(top), int(bottom), int(left),
                          int(right), _p(out))
    return out


def get_tiles(nrows, ncols, tile_size, overlap):
    """tiling.getTilesForFile (tiling.py:376-443): dict (col,row) -> (xpos, ypos, xsize, ysize)."""
    tiles = {}
    ypos = 0; ytile = 0; xtile = 0; ydone = False
    while not ydone:
        xdone = False; xpos = 0; xtile = 0; ysize = tile_size
        if ypos + ysize * 2 > nrows:
            ysize = nrows - ypos; ydone = True
            if ysize == 0:
                break
        while not xdone:
            xsize = tile_size
            if xpos + xsize * 2 > ncols:
                xsize = ncols - xpos; xdone = True
                if xsize == 0:
                    break
            tiles[(xtile, ytile)] = (xpos, ypos, xsize, ysize)
            xpos += tile_size - overlap; xtile += 1
        ypos += tile_size - overlap; ytile += 1
    return tiles, xtile, ytile


def stitch_tiles(tile_segs, tiles, ntcols, ntrows, nrows, ncols, overlap, simple=False):
    """tiling.stitchTiles (tiling.py:950-1064) on in-memory tiles.  tile_segs: dict
    (col,row) -> local label array.  Returns (mosaic, maxSegId, hist)."""
    margin = int(overlap / 2)
    out = np.zeros((nrows, ncols), dtype=np.uint32)
    cache = {}
    max_seg = 0
    for row in range(ntrows):
        for col in range(ntcols):
            xpos, ypos, xsize, ysize = tiles[(col, row)]
            t = tile_segs[(col, row)]
            top, bottom, left, right = margin, ysize - margin, margin, xsize - margin
            xout, yout = xpos + margin, ypos + margin
            if row == 0:
                top = 0; yout = ypos
            if row == ntrows - 1:
                bottom = ysize
            if col == 0:
                left = 0; xout = xpos
            if col == ntcols - 1:
                right = xsize
            if simple:
                t = np.where(t == 0, 0, t + np.uint32(max_seg)).astype(np.uint32)
            else:
                t = recode_tile(t, overlap, cache.get(('b', col, row - 1)) if row > 0 else None,
                                cache.get(('r', col - 1, row)) if col > 0 else None, max_seg,
                                top, bottom, left, right)
            trimmed = t[top:bottom, left:right]
            out[yout:yout + trimmed.shape[0], xout:xout + trimmed.shape[1]] = trimmed
            if col != ntcols - 1:
                cache[('r', col, row)] = t[:, -overlap:].copy()
            if row != ntrows - 1:
                cache[('b', col, row)] = t[-overlap:, :].copy()
            max_seg = max(max_seg, int(trimmed.max()))
    hist = np.bincount(out.ravel(), minlength=max_seg + 1).astype(np.uint32)
    hist[0] = 0
    return out, max_seg, hist


STAT_IDS = {'min': 0, 'max': 1, 'mean': 2, 'stddev': 3, 'median': 4, 'mode': 5, 'percentile': 6,
            'pixcount': 7}


def make_stats_sel(stats_selection):
    """tilingstats.makeFastStatsSelection (:798-863) with global column index = position."""
    sel = np.empty((len(stats_selection), 5), dtype=np.uint32)
    ni = nf = 0
    for i, st in enumerate(stats_selection):
        name = st[1]
        isf = name in ('mean', 'stddev')
        sel[i] = (i, STAT_IDS[name], int(isf), nf if isf else ni,
                  st[2] if name == 'percentile' else 0xFFFFFFFF)
        if isf:
            nf += 1
        else:
            ni += 1
    return sel, ni, nf


def segstats(seg, band, stats_selection, null_val=None, missing=-9999, max_seg_id=None):
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    band = np.ascontiguousarray(band)
    assert band.dtype in DTYPES and band.size == seg.size
    if max_seg_id is None:
        max_seg_id = int(seg.max()) if seg.size else 0
    sel, ni, nf = make_stats_sel(stats_selection)
    ic = np.zeros((ni, max_seg_id + 1), dtype=np.int64)
    fc = np.zeros((nf, max_seg_id + 1), dtype=np.float32)
    rc = lib().orc_segstats(_p(seg), _p(band), DTYPES[band.dtype], ctypes.c_int64(seg.size),
                            ctypes.c_uint32(max_seg_id), int(null_val is not None),
                            ctypes.c_int64(0 if null_val is None else int(null_val)), _p(sel),
                            len(stats_selection), ctypes.c_int64(int(missing)), _p(ic), _p(fc))
    assert rc == 0
    return ic, fc


def subset_recode(seg, tlx, tly, xs, ys, mask=None, tile_size=1024):
    """subset.subsetImage's recode: (out (ys, xs) uint32, orig ids per new id, histogram per new id)."""
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    assert 0 <= tlx and 0 <= tly and tlx + xs <= seg.shape[1] and tly + ys <= seg.shape[0]
    if mask is not None:
        mask = np.ascontiguousarray(mask != 0, dtype=np.uint8)
        assert mask.shape == (ys, xs)
    out = np.zeros((ys, xs), dtype=np.uint32)
    orig = np.zeros(xs * ys + 1, dtype=np.uint32)
    hist = np.zeros(xs * ys + 1, dtype=np.uint32)
    f = lib().orc_subset_recode
    f.restype = ctypes.c_uint32
    n = f(_p(seg), ctypes.c_int64(seg.shape[1]), ctypes.c_int64(tlx), ctypes.c_int64(tly),
          ctypes.c_int64(xs), ctypes.c_int64(ys), _p(mask) if mask is not None else None,
          ctypes.c_int64(tile_size), ctypes.c_uint32(int(seg.max()) if seg.size else 0), _p(out),
          _p(orig), _p(hist))
    return out, orig[:n + 1].copy(), hist[:n + 1].copy()


SPATIAL_FUNCS = {'meancoord': 0, 'numedge': 1, 'variogram': 2}


def spatialstats(seg, band, func, param, null_val, nint, nflt, missing=-9999, tile_size=1024,
                 max_seg_id=None):
    """calcPerSegmentSpatialStatsTiled with a built-in user function: (intcols (nint, S+1) int64,
    floatcols (nflt, S+1) float32)."""
    seg = np.ascontiguousarray(seg, dtype=np.uint32)
    band = np.ascontiguousarray(band)
    assert band.dtype in DTYPES and band.shape == seg.shape and seg.ndim == 2
    if max_seg_id is None:
        max_seg_id = int(seg.max()) if seg.size else 0
    params = np.zeros(6, dtype=np.float64)
    pv = np.atleast_1d(np.asarray(param, dtype=np.float64))
    params[:len(pv)] = pv
    ic = np.zeros((max(nint, 1), max_seg_id + 1), dtype=np.int64)
    fc = np.zeros((max(nflt, 1), max_seg_id + 1), dtype=np.float32)
    rc = lib().orc_spatialstats(_p(seg), _p(band), DTYPES[band.dtype], ctypes.c_int64(seg.shape[0]),
                                ctypes.c_int64(seg.shape[1]), ctypes.c_uint32(max_seg_id),
                                ctypes.c_int64(int(null_val)), SPATIAL_FUNCS[func], _p(params),
                                ctypes.c_int64(tile_size), ctypes.c_int64(int(missing)), nint, nflt,
                                _p(ic), _p(fc))
    assert rc == 0
    return ic[:nint], fc[:nflt]
