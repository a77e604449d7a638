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
_LIBPATH = os.path.join(_HERE, 'liboracle.so')

DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.uint16): 2,
          np.dtype(np.int32): 3, np.dtype(np.uint32): 4}


def build(force=False):
    src = os.path.join(_HERE, 'shepseg_oracle.c')
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


def kmeans_fit(xsample, init, max_iter=300, tol=1e-4):
    x = np.ascontiguousarray(xsample, dtype=np.float64)
    init = np.ascontiguousarray(init, dtype=np.float64)
    n, nb = x.shape
    k = init.shape[0]
    centres = np.empty((k, nb), dtype=np.float64)
    labels = np.empty(n, dtype=np.int32)
    nit = ctypes.c_int(0)
    rc = lib().orc_kmeans_fit(_p(x), ctypes.c_int64(n), nb, k, _p(init), int(max_iter),
                              ctypes.c_double(tol), _p(centres), _p(labels), ctypes.byref(nit))
    assert rc == 0
    return centres, labels, nit.value
