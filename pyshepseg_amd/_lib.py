"""ctypes binding of libshepseg_hip.so (the C-ABI declared in include/shepseg_hip.h).

There is no CPU fallback: if the HIP library is missing, or no MI355X-class device is usable,
every product entry point raises.  The library is built in-tree by
``pyshepseg_amd/csrc/Makefile`` (``__graft_entry__.build()`` runs it).
"""
import ctypes
import os
import threading

# more hardware queues than ROCm's default of 4, so that the per-tile worker streams do not
# queue behind each other's long-running kernels (must be set before the HIP runtime starts)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SHEPSEG_LIBPATH: another build of the same ABI, for A/B timing on one box)
LIBPATH = os.environ.get('SHEPSEG_LIBPATH') or os.path.join(_HERE, 'libshepseg_hip.so')

SHP_DTYPES = {numpy.dtype(numpy.uint8): 0, numpy.dtype(numpy.int16): 1,
              numpy.dtype(numpy.uint16): 2, numpy.dtype(numpy.int32): 3,
              numpy.dtype(numpy.uint32): 4}

SHP_ERR_NO_DEVICE = -1


class ShepsegHipError(RuntimeError):
    """Raised when the HIP library is missing, no device is usable, or a call fails."""


_c = ctypes
_vp = _c.c_void_p
_SIGS = {
    'shp_version': (_c.c_int, []),
    'shp_device_count': (_c.c_int, []),
    'shp_ctx_create': (_c.c_int, [_c.c_int, _c.POINTER(_vp)]),
    'shp_ctx_create_priority': (_c.c_int, [_c.c_int, _c.POINTER(_vp)]),
    'shp_ctx_create_shared': (_c.c_int, [_c.c_int, _c.POINTER(_vp)]),
    'shp_ctx_destroy': (None, [_vp]),
    'shp_last_error': (_c.c_char_p, [_vp]),
    'shp_last_timings': (_c.c_int, [_vp, _vp]),
    'shp_prof_get': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int]),
    'shp_kmeans_fit': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _c.c_int,
                                  _c.c_double, _vp, _vp, _c.POINTER(_c.c_int)]),
    'shp_kmeans_fit_typed': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _vp,
                                        _c.c_int, _c.c_double, _vp, _vp, _c.POINTER(_c.c_int)]),
    'shp_kmeans_fit_planar': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int64, _c.c_int,
                                         _vp, _c.c_int, _c.c_double, _vp, _vp, _c.POINTER(_c.c_int),
                                         _c.POINTER(_c.c_int64)]),
    'shp_kmeans_fit_planar_dist': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int64, _c.c_int,
                                         _vp, _c.c_int, _c.c_double, _vp, _vp, _c.POINTER(_c.c_int),
                                         _c.POINTER(_c.c_int64)]),
    'shp_last_fit_path': (_c.c_int, [_vp]),
    'shp_kmeans_assign': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                     _c.c_int, _c.c_int, _c.c_int64, _vp]),
    'shp_clump': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _vp,
                             _c.POINTER(_c.c_uint32)]),
    'shp_make_seg_size': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_uint32, _vp]),
    'shp_eliminate_single': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int, _vp, _c.POINTER(_c.c_uint32)]),
    'shp_eliminate_small': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                       _c.c_int, _c.c_double, _vp, _c.POINTER(_c.c_uint32),
                                       _c.POINTER(_c.c_int64)]),
    'shp_segment_locations': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_uint32, _vp, _vp]),
    'shp_build_segment_spectra': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                             _c.c_uint32, _vp]),
    'shp_segment_tile': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                    _c.c_int, _c.c_int, _c.c_int64, _c.c_int, _c.c_int,
                                    _c.c_double, _vp, _c.POINTER(_c.c_uint32),
                                    _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64),
                                    _c.POINTER(_c.c_uint32)]),
    'shp_synthimg': (_c.c_int, [_vp, _c.c_uint64, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int,
                                _c.c_int, _vp]),
    'shp_dev_alloc': (_c.c_int, [_vp, _c.c_size_t, _c.POINTER(_vp)]),
    'shp_dev_free': (_c.c_int, [_vp, _vp]),
    'shp_dev_upload': (_c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    'shp_dev_download': (_c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    'shp_dev_memset': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_size_t]),
    'shp_host_alloc': (_c.c_int, [_vp, _c.c_size_t, _c.POINTER(_vp)]),
    'shp_host_free': (_c.c_int, [_vp, _vp]),
    'shp_dev_copy': (_c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    'shp_sync': (_c.c_int, [_vp]),
    'shp_dev_synthimg': (_c.c_int, [_vp, _c.c_uint64, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int,
                                    _c.c_int, _vp]),
    'shp_dev_block_labels': (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                        _c.POINTER(_c.c_uint32)]),
    'shp_dev_subsample': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                     _c.c_int, _vp, _c.c_int, _vp]),
    'shp_segment_window_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_int,
                                          _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_double,
                                          _vp, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_int64),
                                          _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint32), _vp]),
    'shp_assign_rects_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                        _c.c_int, _vp, _c.c_int, _c.c_int, _c.c_int64, _vp]),
    'shp_segment_tile_to_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp,
                                           _c.c_int, _c.c_int, _c.c_int64, _c.c_int, _c.c_int,
                                           _c.c_double, _vp, _c.POINTER(_c.c_uint32),
                                           _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64),
                                           _c.POINTER(_c.c_uint32)]),
    'shp_stitch_tile_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_int64,
                                       _vp, _c.c_int64, _c.c_uint32, _c.c_int, _vp, _c.c_int,
                                       _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_int64, _c.c_int,
                                       _c.c_int]),
    'shp_stitch_prepare_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_uint32, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _vp, _vp]),
    'shp_stitch_counts_dev': (_c.c_int, [_vp, _vp, _c.c_uint32, _c.c_uint32, _vp]),
    'shp_renumber_dev': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_uint32, _vp, _c.c_int]),
    'shp_stitch_chain_dev': (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _c.c_int, _vp, _c.c_int64,
                                        _vp, _c.c_int64, _c.c_uint32, _c.c_int, _vp, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_int, _vp, _vp, _vp, _vp, _c.c_int64,
                                        _c.c_int, _c.c_int, _c.c_uint32, _c.c_uint32]),
    'shp_overview_window_dev': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _vp, _c.c_int, _c.c_int]),
    'shp_histogram_dev': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int64, _c.c_uint32, _vp]),
    'shp_hist_stats': (_c.c_int, [_vp, _c.c_int64, _vp]),
    'shp_segstats': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_uint32, _c.c_int,
                                _c.c_int64, _vp, _c.c_int, _c.c_int64, _vp, _vp]),
    'shp_segstats_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_uint32, _c.c_int,
                                    _c.c_int64, _vp, _c.c_int, _c.c_int64, _vp, _vp]),
    'shp_segstats2d_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_int64, _c.c_uint32, _c.c_int,
                                      _c.c_int64, _vp, _c.c_int, _c.c_int64, _vp, _vp]),
    'shp_ctx_reserve': (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int64]),
    'shp_ctx_reserve_query': (_c.c_int, [_vp, _c.c_int, _c.c_int, _c.c_int64, _c.POINTER(_c.c_int64),
                                         _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    'shp_subset_recode': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                     _c.c_int64, _c.c_int64, _vp, _c.c_int, _c.c_uint32, _vp, _vp,
                                     _vp, _c.c_int64, _vp]),
    'shp_subset_recode_dev': (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                         _c.c_int64, _c.c_int64, _vp, _c.c_int, _c.c_uint32, _vp, _vp,
                                         _vp, _c.c_int64, _vp]),
    'shp_spatialstats': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_int64, _c.c_uint32,
                                    _c.c_int64, _c.c_int, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp,
                                    _vp]),
    'shp_spatialstats_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_int64, _c.c_uint32,
                                        _c.c_int64, _c.c_int, _vp, _c.c_int64, _c.c_int, _c.c_int,
                                        _vp, _vp]),
    'shp_comm_unique_id': (_c.c_int, [_vp]),
    'shp_comm_create': (_c.c_int, [_vp, _c.c_int, _c.c_int, _vp, _c.POINTER(_vp)]),
    'shp_comm_destroy': (None, [_vp]),
    'shp_comm_send': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int]),
    'shp_comm_recv': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int]),
    'shp_comm_bcast': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int]),
    'shp_comm_allgather': (_c.c_int, [_vp, _vp, _vp, _c.c_size_t]),
    'shp_comm_allreduce': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int]),
    'shp_comm_count': (_c.c_int, [_vp, _c.POINTER(_c.c_int)]),
    'shp_comm_isend': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int, _vp]),
    'shp_comm_irecv': (_c.c_int, [_vp, _vp, _c.c_size_t, _c.c_int, _vp]),
    'shp_comm_group': (_c.c_int, [_vp, _c.c_int]),
    'shp_comm_wait': (_c.c_int, [_vp, _vp]),
    'shp_comm_drain': (_c.c_int, [_vp]),
    'shp_gather_flagged_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_uint32, _vp,
                                          _c.c_int64, _vp, _vp, _vp]),
    'shp_dstats_local_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int64, _c.c_int64, _c.c_uint32, _c.c_int,
                                        _c.c_int64, _vp, _c.c_int, _c.c_int64, _vp, _c.c_int, _vp,
                                        _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_c.c_int64),
                                        _c.POINTER(_c.c_int64)]),
    'shp_dstats_merge_dev': (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _c.c_int, _vp, _c.c_int, _c.c_uint32, _c.c_int,
                                        _c.c_int64, _vp, _c.c_int, _c.c_int64, _c.c_uint32, _c.c_uint32, _vp,
                                        _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
}

_lib = None
_lib_lock = threading.Lock()


def lib():
    """Load the shared library (no device needed for this step)."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIBPATH):
                raise ShepsegHipError(
                    "HIP library not built: %s is missing (run __graft_entry__.build() or "
                    "`make -C pyshepseg_amd/csrc`); there is no CPU fallback" % LIBPATH)
            L = ctypes.CDLL(LIBPATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(L, name)        # AttributeError here = ABI mismatch with the header
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


class Context(object):
    """One shp_ctx: a HIP stream plus device workspace.  Not shared between threads."""

    def __init__(self, device=None, highPriority=False, sharedStreams=False):
        L = lib()
        if device is None:
            device = int(os.environ.get('SHEPSEG_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        ndev = L.shp_device_count()
        if ndev <= 0:
            raise ShepsegHipError("no HIP device available: pyshepseg_amd runs only on a GPU "
                                  "(gfx950 / MI355X); there is no CPU fallback")
        device = device % ndev
        h = _vp()
        create = (L.shp_ctx_create_shared if sharedStreams else
                  L.shp_ctx_create_priority if highPriority else L.shp_ctx_create)
        rc = create(device, ctypes.byref(h))
        if rc != 0:
            raise ShepsegHipError("shp_ctx_create(device=%d) failed with code %d" % (device, rc))
        self.handle = h
        self.device = device
        self._L = L

    def check(self, rc):
        if rc != 0:
            msg = self._L.shp_last_error(self.handle)
            raise ShepsegHipError("libshepseg_hip error %d: %s" % (rc, (msg or b'').decode()))

    def timings(self):
        out = (ctypes.c_double * 8)()
        self._L.shp_last_timings(self.handle, out)
        names = ('assign', 'clump', 'single', 'small', 'h2d', 'd2h', 'total')
        return dict(zip(names, list(out)[:7]))

    def close(self):
        if self.handle is not None:
            self._L.shp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def ctx():
    """The calling thread's context (created on first use)."""
    c = getattr(_tls, 'ctx', None)
    if c is None or c.handle is None:
        c = Context()
        _tls.ctx = c
    return c


_pool = []
_pool_lock = threading.Lock()


class pooled_ctx(object):
    """Borrow a worker context (stream + grown device workspace) from a process-wide pool, so
    that worker threads of successive tiled runs reuse the same HIP allocations."""
    def __enter__(self):
        with _pool_lock:
            self.c = _pool.pop() if _pool else None
        if self.c is None:
            # worker contexts borrow their streams from the library's pool (common.h): the tiles in
            # flight are then not bounded by the hardware queues
            self.c = Context(sharedStreams=os.environ.get('SHEPSEG_SHARED_STREAMS', '1') != '0')
        return self.c

    def __exit__(self, *args):
        with _pool_lock:
            _pool.append(self.c)


def pool_contexts():
    with _pool_lock:
        return list(_pool)


_chain_ctx = {}


def chain_ctx():
    """A per-thread high-priority context for the sequential stitch chain."""
    c = getattr(_tls, 'chain', None)
    if c is None or c.handle is None:
        c = Context(highPriority=True)
        _tls.chain = c
    return c


def ptr(a):
    return a.ctypes.data_as(_vp)


def as_image(img):
    """C-contiguous (nBands, nRows, nCols) integer image + its dtype code."""
    img = numpy.ascontiguousarray(img)
    if img.ndim != 3:
        raise ValueError("img must have shape (nBands, nRows, nCols)")
    if img.dtype not in SHP_DTYPES:
        if img.dtype == numpy.int8:
            img = img.astype(numpy.int16)
        elif img.dtype in (numpy.dtype(numpy.int64), numpy.dtype(numpy.uint64)):
            lo, hi = (int(img.min()), int(img.max())) if img.size else (0, 0)
            if lo >= 0 and hi <= 0xFFFFFFFF:
                img = img.astype(numpy.uint32)
            elif lo >= -2**31 and hi < 2**31:
                img = img.astype(numpy.int32)
            else:
                raise TypeError("64-bit imagery outside the 32-bit range is not supported")
        else:
            raise TypeError("img must be an integer array (got %s)" % img.dtype)
    return img, SHP_DTYPES[img.dtype]
