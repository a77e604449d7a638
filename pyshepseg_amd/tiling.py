"""Drop-in for the segmentation side of ``pyshepseg.tiling``: tiled Shepherd segmentation of a
large raster with the per-tile work and the cross-tile stitch on the GPU.

Same entry point and arguments as the reference (``doTiledShepherdSegmentation``,
tiling.py:446-571), same tile grid (``getTilesForFile``, :376-443), same global k-means
subsample (``fitSpectralClustersWholeFile`` / ``readSubsampledImageBand``, :154-314) and the
same sequential stitch semantics (``stitchTiles``, :950-1064) -- but tiles are segmented by
HIP worker contexts (one stream each) and their labels never leave HBM until the stitched
raster is complete.  What is NOT here, on purpose: the CPU worker farms / AWS Fargate /
network channel (tiling.py:590-697, :1531-1912), overviews and colour tables.

Rasters.  GDAL is optional (imported lazily).  ``infile`` may be
  * a ``(nBands, nRows, nCols)`` numpy array or a ``.npy`` path (memory-mapped),
  * a :class:`DeviceRaster` (already resident in HBM, e.g. synthetic benchmark imagery),
  * anything GDAL opens, when ``osgeo`` is importable.
``outfile`` may be ``None`` (labels returned in ``result.segimg``), a ``.npy`` path, or a GDAL
path when ``osgeo`` is importable.
"""
import ctypes
import os
import queue
import sys
import threading
import time

import numpy

from . import _lib
from . import shepseg

TILESIZE = 1024                 # block size of the subsample reader (reference tiling.py:93)
DFLT_TILESIZE = 4096
DFLT_OVERLAPSIZE = 1024
DFLT_CHUNKSIZE = 100000

CONC_NONE = "CONC_NONE"
CONC_THREADS = "CONC_THREADS"
CONC_FARGATE = "CONC_FARGATE"
CONC_SUBPROC = "CONC_SUBPROC"


class PyShepSegTilingError(Exception):
    pass


class TiledSegmentationResult(object):
    """Result of tiled segmentation (reference tiling.py:112-151).  ``segimg`` (the stitched
    label array) and ``hist`` are additions for the in-memory / .npy path."""
    def __init__(self):
        self.maxSegId = None
        self.numTileRows = None
        self.numTileCols = None
        self.subsamplePcnt = None
        self.maxSpectralDiff = None
        self.kmeans = None
        self.hasEmptySegments = None
        self.timings = None
        self.outDs = None
        self.segimg = None
        self.hist = None
        self.overviews = None           # {level: array}: the output file's pyramid layers (tiling.py:1360-1404)
        self.bandStatistics = None      # [(item, value)]: the STATISTICS_* metadata (utils.py:47-95)


class SegmentationConcurrencyConfig(object):
    """Accepted for API compatibility (reference tiling.py:590-634).  Only ``numWorkers``
    matters here: it is the number of HIP worker contexts (streams) segmenting tiles
    concurrently on the GPU.  CONC_NONE means one worker; the CPU farm types
    (CONC_SUBPROC / CONC_FARGATE) are out of scope and are run as CONC_THREADS."""
    def __init__(self, concurrencyType=CONC_NONE, numWorkers=0, maxConcurrentReads=20,
                 tileCompletionTimeout=60, barrierTimeout=60, fargateCfg=None):
        if concurrencyType not in (CONC_NONE, CONC_THREADS, CONC_FARGATE, CONC_SUBPROC):
            raise ValueError("Unknown concurrencyType '{}'".format(concurrencyType))
        self.concurrencyType = concurrencyType
        self.numWorkers = numWorkers
        self.maxConcurrentReads = maxConcurrentReads
        self.tileCompletionTimeout = tileCompletionTimeout
        self.barrierTimeout = barrierTimeout
        self.fargateCfg = fargateCfg


class Timers(object):
    """Named interval timers, same interval names as the reference (timinghooks.py:18-160)."""
    def __init__(self):
        self.pairs = {}
        self.lock = threading.Lock()

    class _Interval(object):
        def __init__(self, timers, name):
            self.timers, self.name = timers, name

        def __enter__(self):
            self.t0 = time.time()

        def __exit__(self, *args):
            t1 = time.time()
            with self.timers.lock:
                self.timers.pairs.setdefault(self.name, []).append((self.t0, t1))

    def interval(self, name):
        return Timers._Interval(self, name)

    def makeSummaryDict(self):
        d = {}
        for name, pairs in self.pairs.items():
            iv = numpy.array([b - a for (a, b) in pairs])
            d[name] = {'total': float(iv.sum()), 'min': float(iv.min()), 'max': float(iv.max()),
                       'count': len(pairs)}
        return d


# ------------------------------------------------------------------------------------------
# raster sources
# ------------------------------------------------------------------------------------------
class DeviceRaster(object):
    """A band-planar raster resident in HBM (device 0 of the calling thread's context)."""
    def __init__(self, nBands, nRows, nCols, dtype=numpy.uint16, nullVal=None, cached=False):
        self.dtype = numpy.dtype(dtype)
        self.shape = (int(nBands), int(nRows), int(nCols))
        self.nullVal = nullVal
        self.RasterXSize, self.RasterYSize = self.shape[2], self.shape[1]
        self.nbytes = int(nBands) * int(nRows) * int(nCols) * self.dtype.itemsize
        c = _lib.ctx()
        self.cached = cached                # block taken from / returned to the scratch-raster cache
        if cached:
            p = _devAlloc(c, self.nbytes)
        else:
            p = ctypes.c_void_p()
            c.check(c._L.shp_dev_alloc(c.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self.device = c.device

    @classmethod
    def synth(cls, seed, nBands, nRows, nCols, y0=0, x0=0):
        """`synthimg v1` generated on the device."""
        r = cls(nBands, nRows, nCols, numpy.uint16)
        c = _lib.ctx()
        c.check(c._L.shp_dev_synthimg(c.handle, seed, nBands, y0, x0, nRows, nCols,
                                      ctypes.c_void_p(r.ptr)))
        return r

    @classmethod
    def fromArray(cls, img, nullVal=None):
        img, _dt = _lib.as_image(img)
        r = cls(img.shape[0], img.shape[1], img.shape[2], img.dtype, nullVal)
        c = _lib.ctx()
        c.check(c._L.shp_dev_upload(c.handle, ctypes.c_void_p(r.ptr), _lib.ptr(img), r.nbytes))
        return r

    def toArray(self):
        out = numpy.empty(self.shape, dtype=self.dtype)
        c = _lib.ctx()
        c.check(c._L.shp_dev_download(c.handle, _lib.ptr(out), ctypes.c_void_p(self.ptr),
                                      self.nbytes))
        return out

    def free(self):
        if self.ptr:
            c = _lib.ctx()
            if self.cached:
                _devRelease(c, ctypes.c_void_p(self.ptr), self.nbytes)
            else:
                c.check(c._L.shp_dev_free(c.handle, ctypes.c_void_p(self.ptr)))
            self.ptr = None


class _ArraySource(object):
    """Host ndarray / memmap source with the reader interface the driver needs."""
    def __init__(self, arr, nullVal=None):
        if arr.ndim != 3:
            raise PyShepSegTilingError("raster array must have shape (nBands, nRows, nCols)")
        self.arr = arr
        self.shape = arr.shape
        self.dtype = arr.dtype
        self.nullVal = nullVal
        self.RasterXSize, self.RasterYSize = arr.shape[2], arr.shape[1]

    def read(self, bands, xpos, ypos, xsize, ysize):
        return numpy.ascontiguousarray(self.arr[bands, ypos:ypos + ysize, xpos:xpos + xsize])

    def readRowsInto(self, bands, y0, y1, out):
        """rows [y0, y1) of the given bands into out (nBands, y1 - y0, nCols), band by band"""
        for (i, b) in enumerate(bands):
            numpy.copyto(out[i], self.arr[b, y0:y1, :], casting='unsafe')


class _GdalSource(object):
    def __init__(self, path):
        from osgeo import gdal
        gdal.UseExceptions()
        self.ds = gdal.Open(path)
        self.path = path
        self.RasterXSize, self.RasterYSize = self.ds.RasterXSize, self.ds.RasterYSize
        self.shape = (self.ds.RasterCount, self.RasterYSize, self.RasterXSize)
        self.nullVal = None
        self.local = threading.local()
        from osgeo import gdal_array
        self.dtype = numpy.dtype(gdal_array.GDALTypeCodeToNumericTypeCode(self.ds.GetRasterBand(1).DataType))

    def bandNull(self, bandNumbers):
        arr = numpy.array([self.ds.GetRasterBand(i).GetNoDataValue() for i in bandNumbers])
        if (arr != arr[0]).any():
            raise PyShepSegTilingError("Different null values in some bands")
        return arr[0]

    def read(self, bands, xpos, ypos, xsize, ysize):
        from osgeo import gdal
        ds = getattr(self.local, 'ds', None)
        if ds is None:
            ds = self.local.ds = gdal.Open(self.path)      # per-thread dataset (tiling.py:1565)
        return numpy.array([ds.GetRasterBand(int(b) + 1).ReadAsArray(xpos, ypos, xsize, ysize)
                            for b in bands])

    def readRowsInto(self, bands, y0, y1, out):
        from osgeo import gdal
        ds = getattr(self.local, 'ds', None)
        if ds is None:
            ds = self.local.ds = gdal.Open(self.path)
        for (i, b) in enumerate(bands):
            out[i] = ds.GetRasterBand(int(b) + 1).ReadAsArray(0, y0, self.RasterXSize, y1 - y0)


def _open_source(infile):
    if isinstance(infile, DeviceRaster):
        return infile
    if isinstance(infile, numpy.ndarray):
        return _ArraySource(infile)
    if isinstance(infile, str) and infile.endswith('.npy'):
        return _ArraySource(numpy.load(infile, mmap_mode='r'))
    try:
        import osgeo  # noqa: F401
    except ImportError:
        raise PyShepSegTilingError("cannot open %r: GDAL (osgeo) is not importable here; pass a "
                                   "numpy array, a .npy path or a DeviceRaster" % (infile,))
    return _GdalSource(infile)


# ------------------------------------------------------------------------------------------
# device scratch rasters (tile labels, stitched output) are cached between runs: hipMalloc /
# hipFree of multi-GB blocks are slow and synchronise the whole device
# ------------------------------------------------------------------------------------------
_devCache = {}
_devCacheLock = threading.Lock()
_devCacheBytes = [0]
_DEV_CACHE_CAP = int(os.environ.get('SHEPSEG_DEVCACHE_GB', '96')) << 30     # keep at most this much


def _devAlloc(c, nbytes):
    nbytes = max(int(nbytes), 16)
    with _devCacheLock:
        lst = _devCache.get((c.device, nbytes))
        if lst:
            _devCacheBytes[0] -= nbytes
            return ctypes.c_void_p(lst.pop())
    p = ctypes.c_void_p()
    c.check(c._L.shp_dev_alloc(c.handle, nbytes, ctypes.byref(p)))
    return p


def _devRelease(c, p, nbytes):
    if p is None or not p.value:
        return
    nbytes = max(int(nbytes), 16)
    with _devCacheLock:
        if _devCacheBytes[0] + nbytes <= _DEV_CACHE_CAP:
            _devCache.setdefault((c.device, nbytes), []).append(p.value)
            _devCacheBytes[0] += nbytes
            return
    c._L.shp_dev_free(c.handle, p)


def clearDeviceCache():
    """Free every cached device scratch raster."""
    c = _lib.ctx()
    with _devCacheLock:
        for (_dev, _n), lst in _devCache.items():
            for v in lst:
                c._L.shp_dev_free(c.handle, ctypes.c_void_p(v))
        _devCache.clear()
        _devCacheBytes[0] = 0


# ------------------------------------------------------------------------------------------
# tile grid (reference tiling.py:317-443)
# ------------------------------------------------------------------------------------------
class TileInfo(object):
    """Pixel coordinates of the tiles within an image (reference tiling.py:317-374)."""
    def __init__(self):
        self.tiles = {}
        self.ncols = None
        self.nrows = None

    def addTile(self, xpos, ypos, xsize, ysize, col, row):
        self.tiles[(col, row)] = (xpos, ypos, xsize, ysize)

    def getNumTiles(self):
        return len(self.tiles)

    def getTile(self, col, row):
        return self.tiles[(col, row)]


def _axisTiles(n, tileSize, step):
    """(start, size) of the tiles along one axis of n pixels: a tile starts every `step` pixels;
    one that could not be followed by another whole tile (start + 2 * tileSize > n) runs to the
    image edge and is the last one."""
    if n <= 0:
        return []
    nfull = 0 if n < 2 * tileSize else (n - 2 * tileSize) // step + 1
    return [(i * step, tileSize) for i in range(nfull)] + [(nfull * step, n - nfull * step)]


def getTilesForFile(ds, tileSize, overlapSize):
    """TileInfo for a raster (anything with RasterXSize / RasterYSize): the reference's grid
    (tiling.py:376-443) in closed form, axis by axis."""
    tileSize = int(tileSize)
    step = tileSize - int(overlapSize)
    if step <= 0:
        raise PyShepSegTilingError("overlapSize must be smaller than tileSize")
    xs = _axisTiles(ds.RasterXSize, tileSize, step)
    ys = _axisTiles(ds.RasterYSize, tileSize, step)
    tileInfo = TileInfo()
    for (row, (ypos, ysize)) in enumerate(ys):
        for (col, (xpos, xsize)) in enumerate(xs):
            tileInfo.addTile(xpos, ypos, xsize, ysize, col, row)
    tileInfo.nrows = len(ys)
    tileInfo.ncols = len(xs) if ys else 0
    return tileInfo


# ------------------------------------------------------------------------------------------
# whole-image k-means (reference tiling.py:154-314)
# ------------------------------------------------------------------------------------------
def _subsample_indices(n, skip, tileSize=TILESIZE):
    """Indices kept by readSubsampledImageBand along one axis: [::skip] restarted inside every
    1024-pixel block (reference tiling.py:287-311)."""
    idx = []
    for start in range(0, n, tileSize):
        size = min(tileSize, n - start)
        idx.extend(range(start, start + size, skip))
    return numpy.array(idx, dtype=numpy.uint32)


_sampleBuf = {}
_sampleLock = threading.Lock()      # held from the read-back to the end of the fit that consumes it


def readSubsampledImage(src, bandNumbers, subsampleProp):
    """Sub-sampled copy of the selected bands: (nBands, nRowsSub, nColsSub)."""
    with _sampleLock:
        out = _readSubsampledImage(src, bandNumbers, subsampleProp)
        return out.copy() if isinstance(src, DeviceRaster) else out


def _readSubsampledImage(src, bandNumbers, subsampleProp):
    """The same without the copy: for a DeviceRaster the result is a buffer kept between calls
    (the caller holds _sampleLock until it has consumed it)."""
    skip = int(round(1. / subsampleProp))
    (nb, nlines, npix) = src.shape
    ry = _subsample_indices(nlines, skip)
    rx = _subsample_indices(npix, skip)
    bands = [b - 1 for b in bandNumbers]
    if isinstance(src, DeviceRaster):
        # the read-back lands in a buffer kept between calls: a fresh 12-MB array costs its page
        # faults inside the device-to-host copy (1 ms vs 10-30 ms)
        key = (nb, len(ry), len(rx), numpy.dtype(src.dtype).str)
        out = _sampleBuf.get(key)
        if out is None:
            _sampleBuf.clear()
            out = _sampleBuf[key] = numpy.empty((nb, len(ry), len(rx)), dtype=src.dtype)
        c = _lib.ctx()
        c.check(c._L.shp_dev_subsample(c.handle, ctypes.c_void_p(src.ptr),
                                       _lib.SHP_DTYPES[src.dtype], nb, nlines, npix,
                                       _lib.ptr(ry), len(ry), _lib.ptr(rx), len(rx), _lib.ptr(out)))
        if bands == list(range(nb)):
            return out              # (consumed by fitSpectralClusters before the next call)
        return numpy.ascontiguousarray(out[bands])
    if isinstance(src, _ArraySource):
        # the kept rows only (a memmap then touches 1/skip of the file, not all of it), a band per
        # thread: the row gather of a 40000-column raster is 80 MB per band
        out = numpy.empty((len(bands), len(ry), len(rx)), dtype=src.arr.dtype)

        def one(i):
            out[i] = src.arr[bands[i]][ry][:, rx]
        ts = [threading.Thread(target=one, args=(i,)) for i in range(len(bands))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return out
    rows = []
    for ypos in range(0, nlines, TILESIZE):
        ysize = min(TILESIZE, nlines - ypos)
        cols = []
        for xpos in range(0, npix, TILESIZE):
            xsize = min(TILESIZE, npix - xpos)
            t = src.read(bands, xpos, ypos, xsize, ysize)
            cols.append(t[:, ::skip, ::skip])
        rows.append(numpy.concatenate(cols, axis=2))
    return numpy.ascontiguousarray(numpy.concatenate(rows, axis=1))


def fitSpectralClustersWholeFile(inDs, bandNumbers, numClusters=60, subsamplePcnt=None,
        imgNullVal=None, fixedKMeansInit=False):
    """Read a sub-sample of the whole raster and fit the spectral clusters on it
    (reference tiling.py:154-226).  Returns (kmeansObj, subsamplePcnt, imgNullVal)."""
    if subsamplePcnt is None:
        dfltTotalPixels = 1000000
        totalImagePixels = inDs.RasterXSize * inDs.RasterYSize
        subsampleProp = numpy.sqrt(dfltTotalPixels / totalImagePixels)
        subsampleProp = min(1, subsampleProp)
        subsamplePcnt = 100 * subsampleProp**2
    else:
        subsampleProp = numpy.sqrt(subsamplePcnt / 100.0)
    if imgNullVal is None:
        if isinstance(inDs, _GdalSource):
            imgNullVal = inDs.bandNull(bandNumbers)
        else:
            imgNullVal = inDs.nullVal
    with _sampleLock:
        img = _readSubsampledImage(inDs, bandNumbers, subsampleProp)
        kmeansObj = shepseg.fitSpectralClusters(img, numClusters=numClusters, subsamplePcnt=100,
                                                imgNullVal=imgNullVal, fixedKMeansInit=fixedKMeansInit)
    return (kmeansObj, subsamplePcnt, imgNullVal)


# ------------------------------------------------------------------------------------------
# raster I/O pipeline (BASELINE config 3: "async read / compute overlap on HIP streams")
# ------------------------------------------------------------------------------------------
class _PinnedBuffer(object):
    """Page-locked host memory with a numpy view (transfers from pageable memory are staged by the
    runtime in small synchronous pieces: 1-30 ms for a few MB, erratic)."""
    def __init__(self, c, nbytes):
        self.c = c
        self.nbytes = max(int(nbytes), 16)
        p = ctypes.c_void_p()
        c.check(c._L.shp_host_alloc(c.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self.u8 = numpy.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self.ptr))

    def view(self, dtype, shape):
        n = int(numpy.prod(shape)) * numpy.dtype(dtype).itemsize
        return self.u8[:n].view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.u8 = None
            self.c.check(self.c._L.shp_host_free(self.c.handle, ctypes.c_void_p(self.ptr)))
            self.ptr = None


_pinnedPool = {}
_pinnedLock = threading.Lock()


def _pinnedGet(c, nbytes):
    """Page-locking host memory is slow (~1 s per few GB): the staging buffers are kept between runs."""
    nbytes = max(int(nbytes), 16)
    with _pinnedLock:
        lst = _pinnedPool.get(nbytes)
        if lst:
            return lst.pop()
    return _PinnedBuffer(c, nbytes)


def _pinnedPut(buf):
    with _pinnedLock:
        _pinnedPool.setdefault(buf.nbytes, []).append(buf)


def clearPinnedPool():
    with _pinnedLock:
        for lst in _pinnedPool.values():
            for b in lst:
                b.c = _lib.ctx()
                b.free()
        _pinnedPool.clear()


STREAM_BLOCK_ROWS = int(os.environ.get('SHEPSEG_STREAM_ROWS', '256'))


class _RasterStreamer(object):
    """Brings a host raster (array, memmap, GDAL) into a band-planar device raster block of rows by
    block of rows, once per pixel: a reader thread fills page-locked buffers (the file read), an
    uploader thread sends them to HBM on its own stream, and tiles start as soon as their last
    row is resident (waitRows).  What the reference does per tile under a read semaphore
    (tiling.py:1436-1443, :1575-1583) -- but its overlapping tiles read every pixel 1.6 times,
    and a 288-GB device holds the whole raster."""
    def __init__(self, src, bands, ras, timings, nReaders=None):
        self.src, self.bands, self.ras, self.timings = src, list(bands), ras, timings
        (self.nb, self.nrows, self.ncols) = ras.shape
        self.blockRows = max(1, min(STREAM_BLOCK_ROWS, self.nrows))
        self.nblocks = (self.nrows + self.blockRows - 1) // self.blockRows
        if nReaders is None:
            nReaders = int(os.environ.get('SHEPSEG_STREAM_READERS', '6'))
        nReaders = max(1, min(nReaders, self.nblocks))
        self.cond = threading.Condition()
        self.rowsReady = 0
        self.blockDone = numpy.zeros(self.nblocks, dtype=bool)
        self.nextBlock = 0
        self.readersLeft = nReaders
        self.error = None
        self.stop = threading.Event()
        self.cUp = _lib.Context()
        self.t0 = time.time()
        blockBytes = self.nb * self.blockRows * self.ncols * ras.dtype.itemsize
        self.free = queue.Queue()
        self.full = queue.Queue()
        self.bufs = [_pinnedGet(self.cUp, blockBytes) for _ in range(nReaders + 2)]
        for b in self.bufs:
            self.free.put(b)
        # several readers: one thread copies ~15 GB/s out of the page cache, PCIe takes 50
        self.threads = [threading.Thread(target=self._guard, args=(self._reader,), daemon=True)
                        for _ in range(nReaders)]
        self.threads.append(threading.Thread(target=self._guard, args=(self._uploader,), daemon=True))
        for t in self.threads:
            t.start()

    def _guard(self, fn):
        try:
            fn()
        except Exception as e:
            with self.cond:
                self.error = e
                self.cond.notify_all()
            self.stop.set()
            self.full.put(None)

    def _reader(self):
        while not self.stop.is_set():
            with self.cond:
                k = self.nextBlock
                self.nextBlock += 1
            if k >= self.nblocks:
                break
            (y0, y1) = (k * self.blockRows, min(self.nrows, (k + 1) * self.blockRows))
            buf = None
            while buf is None and not self.stop.is_set():
                try:
                    buf = self.free.get(timeout=0.2)
                except queue.Empty:
                    pass
            if self.stop.is_set():
                return
            with self.timings.interval('reading'):
                v = buf.view(self.ras.dtype, (self.nb, y1 - y0, self.ncols))
                self.src.readRowsInto(self.bands, y0, y1, v)
            self.full.put((buf, k, y0, y1))
        with self.cond:
            self.readersLeft -= 1
            last = self.readersLeft == 0
        if last:
            self.full.put(None)

    def _uploader(self):
        c = self.cUp
        isz = self.ras.dtype.itemsize
        while True:
            item = self.full.get()
            if item is None or self.stop.is_set():
                return
            (buf, k, y0, y1) = item
            nbytes = (y1 - y0) * self.ncols * isz
            for b in range(self.nb):
                dst = self.ras.ptr + (b * self.nrows + y0) * self.ncols * isz
                c.check(c._L.shp_dev_upload(c.handle, ctypes.c_void_p(dst),
                                            ctypes.c_void_p(buf.ptr + b * nbytes), nbytes))
            self.free.put(buf)
            if os.environ.get('SHEPSEG_IO_TIMING') and y1 == self.nrows:
                sys.stderr.write('  [io] last block uploaded (t0 + %.3f s since the stream started)\n'
                                 % (time.time() - self.t0))
            with self.cond:
                self.blockDone[k] = True
                while self.rowsReady < self.nrows and self.blockDone[self.rowsReady // self.blockRows]:
                    self.rowsReady = min(self.nrows, (self.rowsReady // self.blockRows + 1) * self.blockRows)
                self.cond.notify_all()

    def waitRows(self, y1, forceExit=None):
        with self.cond:
            while self.rowsReady < y1:
                if self.error is not None:
                    raise PyShepSegTilingError("reading the raster failed: %s" % (self.error,))
                if forceExit is not None and forceExit.is_set():
                    raise PyShepSegTilingError("raster stream: another worker failed")
                self.cond.wait(timeout=0.5)

    def close(self):
        self.stop.set()
        self.full.put(None)
        for t in self.threads:
            t.join(timeout=60.0)
        if any(t.is_alive() for t in self.threads):
            # a reader still blocked in a slow file / GDAL read would later write into a buffer
            # that the pool had handed to the next run, or the uploader would use a destroyed
            # context: leak both (as the tiled driver does for a stuck worker)
            self.stuck = True
            self.bufs = []
            return
        self.cUp.check(self.cUp._L.shp_sync(self.cUp.handle))
        for b in self.bufs:
            _pinnedPut(b)
        self.bufs = []
        self.cUp.close()


class _NpyRowWriter(object):
    """A (nRows, nCols) uint32 .npy file written rows at a time with pwrite: storing into a fresh
    memory map costs a page fault per 4 KiB (1 GB/s and worse from several threads; populating the
    map ahead of the data with madvise(MADV_POPULATE_WRITE) from background threads was slower
    still: 3.2 s for 6.4 GB), a write into the page cache is one kernel copy.  Buffered writes to one
    file serialise on its inode lock, so the sink tops out near 5 GB/s whatever the thread count."""
    def __init__(self, path, nrows, ncols):
        (self.nrows, self.ncols) = (nrows, ncols)
        with open(path, 'wb') as f:
            numpy.lib.format.write_array_header_1_0(
                f, {'descr': numpy.lib.format.dtype_to_descr(numpy.dtype(numpy.uint32)),
                    'fortran_order': False, 'shape': (nrows, ncols)})
            self.offset = f.tell()
        self.fd = os.open(path, os.O_WRONLY)
        os.ftruncate(self.fd, self.offset + nrows * ncols * 4)

    def writeRows(self, y0, y1, v):
        mv = memoryview(numpy.ascontiguousarray(v)).cast('B')
        pos = self.offset + y0 * self.ncols * 4
        done = 0
        while done < len(mv):
            done += os.pwrite(self.fd, mv[done:], pos + done)

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


class _OutputWriter(object):
    """Finished rows of the stitched raster leave the device while later tiles are still being
    segmented: rows [yLo, yHi) are downloaded through page-locked staging on the writer's own
    stream and handed to the sink (an ndarray / .npy memmap slice assignment, or GDAL WriteArray) --
    instead of one blocking download of the whole raster at the end (tiling.py:1032-1034 writes
    each trimmed tile as it is stitched)."""
    CHUNK_BYTES = 64 << 20

    def __init__(self, d_out, nrows, ncols, sink, timings, nCopiers=None):
        (self.d_out, self.nrows, self.ncols, self.sink, self.timings) = (d_out, nrows, ncols, sink, timings)
        self.c = _lib.Context()
        self.rowsPer = max(1, min(nrows, self.CHUNK_BYTES // max(ncols * 4, 1)))
        if nCopiers is None:
            nCopiers = int(os.environ.get('SHEPSEG_STREAM_WRITERS', '3'))
        self.free = queue.Queue()
        self.bufs = [_pinnedGet(self.c, self.rowsPer * ncols * 4) for _ in range(nCopiers + 1)]
        for b in self.bufs:
            self.free.put(b)
        self.q = queue.Queue()
        self.cq = queue.Queue()
        self.error = None
        # one thread downloads (PCIe), several hand the rows to the sink: a fresh file's pages are
        # first touched there, a few GB/s per thread
        self.copiers = [threading.Thread(target=self._copy, daemon=True) for _ in range(max(1, nCopiers))]
        self.thread = threading.Thread(target=self._run, daemon=True)
        for t in self.copiers + [self.thread]:
            t.start()

    def _copy(self):
        while True:
            item = self.cq.get()
            if item is None:
                return
            (buf, y0, y1) = item
            try:
                with self.timings.interval('writing'):
                    self.sink(y0, y1, buf.view(numpy.uint32, (y1 - y0, self.ncols)))
            except Exception as e:
                self.error = e
            self.free.put(buf)

    def _run(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    return
                (yLo, yHi) = item
                for y0 in range(yLo, yHi, self.rowsPer):
                    y1 = min(yHi, y0 + self.rowsPer)
                    buf = self.free.get()
                    self.c.check(self.c._L.shp_dev_download(
                        self.c.handle, ctypes.c_void_p(buf.ptr),
                        ctypes.c_void_p(self.d_out.value + 4 * y0 * self.ncols), (y1 - y0) * self.ncols * 4))
                    self.cq.put((buf, y0, y1))
        except Exception as e:
            self.error = e

    def rowsFinal(self, yLo, yHi):
        if yHi > yLo:
            self.q.put((yLo, yHi))

    def finish(self):
        self.q.put(None)
        self.thread.join()
        for _ in self.copiers:
            self.cq.put(None)
        for t in self.copiers:
            t.join()
        for b in self.bufs:
            _pinnedPut(b)
        self.c.close()
        if self.error is not None:
            raise self.error


# ------------------------------------------------------------------------------------------
# the tiled driver
# ------------------------------------------------------------------------------------------
class _TileJob(object):
    __slots__ = ('col', 'row', 'xpos', 'ypos', 'xsize', 'ysize', 'offset', 'maxLocal', 'done',
                 'error', 'meta', 'rightOff', 'bottomOff', 'crossPx')


class _MetaArena(object):
    """Bump allocator over one device block for the per-tile stitch tables (4 uint32 arrays of
    maxLocal+1 entries each); falls back to individual allocations when the block is full."""
    def __init__(self, c, nbytes):
        self.c = c
        self.nbytes = max(int(nbytes), 1 << 20)
        self.base = _devAlloc(c, self.nbytes)
        self.used = 0
        self.extra = []
        self.lock = threading.Lock()

    def alloc(self, nseg, c=None):
        """c: the calling thread's own context (contexts are not shared between threads)."""
        need = (16 * int(nseg) + 255) & ~255
        with self.lock:
            if self.used + need <= self.nbytes:
                p = self.base.value + self.used
                self.used += need
                return p
        d = _devAlloc(c if c is not None else self.c, need)
        with self.lock:
            self.extra.append((d, need))
        return d.value

    def release(self):
        _devRelease(self.c, self.base, self.nbytes)
        for (d, n) in self.extra:
            _devRelease(self.c, d, n)
        self.extra = []

    def free(self):
        """After a failed run: hipFree instead of returning the blocks to the cache."""
        for (d, _n) in [(self.base, self.nbytes)] + self.extra:
            if d is not None and d.value:
                self.c._L.shp_dev_free(self.c.handle, d)
        self.extra = []


def layoutStrips(jobs, overlapSize):
    """Offsets (in uint32 elements) of every tile's dense recoded right strip (ysize x overlap)
    and bottom strip (overlap x xsize) inside one block; returns the block's element count."""
    total = 0
    for j in jobs:
        j.rightOff = total
        total += j.ysize * min(overlapSize, j.xsize)
        j.bottomOff = total
        total += min(overlapSize, j.ysize) * j.xsize
    return total


def trimmedWindow(tileInfo, col, row, xpos, ypos, xsize, ysize, overlapSize):
    """(top, bottom, left, right, xout, yout) of a tile: the part of it that is written to the
    output, i.e. the tile minus half the overlap on interior sides (reference tiling.py:997-1022)."""
    marginSize = int(overlapSize / 2)
    (top, bottom, left, right) = (marginSize, ysize - marginSize, marginSize, xsize - marginSize)
    (xout, yout) = (xpos + marginSize, ypos + marginSize)
    if row == 0:
        top = 0
        yout = ypos
    if row == tileInfo.nrows - 1:
        bottom = ysize
    if col == 0:
        left = 0
        xout = xpos
    if col == tileInfo.ncols - 1:
        right = xsize
    return (top, bottom, left, right, xout, yout)


def makeTileJobs(tileInfo, rows=None, tiles=None):
    """Row-major list of tile jobs (optionally only the given tile rows, or only the given set of
    (col, row) tiles) with their offsets in one contiguous label block; returns (jobs, total
    pixels)."""
    jobs = []
    total = 0
    for (col, row) in sorted(tileInfo.tiles.keys(), key=lambda x: (x[1], x[0])):
        if rows is not None and row not in rows:
            continue
        if tiles is not None and (col, row) not in tiles:
            continue
        j = _TileJob()
        (j.col, j.row) = (col, row)
        (j.xpos, j.ypos, j.xsize, j.ysize) = tileInfo.getTile(col, row)
        j.offset = total
        j.maxLocal = 0
        j.done = threading.Event()
        j.error = None
        j.crossPx = (0xFFFFFFFF, 0xFFFFFFFF)
        total += j.xsize * j.ysize
        jobs.append(j)
    return jobs, total


class _ClusterMap(object):
    """Raster-wide map of k-means clusters (uint16, device) filled block by block on demand.  The
    model is global, so a pixel's cluster does not depend on the tile: the overlapping tiles of
    the reference each repeat km.predict (shepseg.py:211) for the 1.6x pixels they share; here
    every block of the raster is assigned once, by the first worker whose window needs it, and
    the tiles copy their windows.  Blocks are claimed under a lock; a worker assigns the blocks it
    claimed (one call, one rectangle per run of blocks in a block row), then waits for the blocks
    of its window that others claimed."""
    BLOCK = 1024

    def __init__(self, nRows, nCols, numWorkers):
        self.c = _lib.ctx()
        self.nRows, self.nCols = nRows, nCols
        self.nbytes = max(nRows * nCols, 1) * 2
        self.dptr = _devAlloc(self.c, self.nbytes)
        B = self.BLOCK
        self.state = numpy.zeros(((nRows + B - 1) // B, (nCols + B - 1) // B), dtype=numpy.int8)
        self.cond = threading.Condition()
        self.workersLeft = numWorkers

    def ensureWindow(self, c, x, y, xs, ys, forceExit, assignRects):
        """Returns the device address of the map once every block under the window is assigned."""
        B = self.BLOCK
        (by0, by1, bx0, bx1) = (y // B, (y + ys + B - 1) // B, x // B, (x + xs + B - 1) // B)
        rects = []
        with self.cond:
            sub = self.state[by0:by1, bx0:bx1]
            for r in range(sub.shape[0]):
                free = numpy.flatnonzero(sub[r] == 0)
                if free.size == 0:
                    continue
                sub[r, free] = 1                                   # claimed
                runs = numpy.split(free, numpy.flatnonzero(numpy.diff(free) != 1) + 1)
                for run in runs:
                    (rx, ry) = ((bx0 + int(run[0])) * B, (by0 + r) * B)
                    rects.append((rx, ry, min(int(run.size) * B, self.nCols - rx), min(B, self.nRows - ry)))
        if rects:
            arr = numpy.array(rects, dtype=numpy.int32)
            try:
                assignRects(arr)
            except Exception:
                forceExit.set()
                with self.cond:
                    self.cond.notify_all()
                raise
            with self.cond:
                for (rx, ry, w, h) in rects:
                    self.state[ry // B, rx // B:(rx + w + B - 1) // B] = 2
                self.cond.notify_all()
        with self.cond:
            while not (self.state[by0:by1, bx0:bx1] == 2).all():
                if forceExit.is_set():
                    raise PyShepSegTilingError("cluster map: another worker failed")
                self.cond.wait(timeout=1.0)
        return self.dptr

    def workerExit(self):
        with self.cond:
            self.workersLeft -= 1
            last = self.workersLeft == 0
        if last and self.dptr is not None:
            _devRelease(self.c, self.dptr, self.nbytes)
            self.dptr = None


def _workersThatFit(numWorkers, dtcode, nBands, maxTilePx, verbose=False):
    """How many of `numWorkers` worker contexts can hold the workspace of the job's largest tile in
    the device memory that is free now (the pooled contexts' present workspaces count: they are
    reused in pool order).  At least 1: a single worker that does not fit fails loudly later."""
    pool = list(reversed(_lib.pool_contexts()))          # pooled_ctx pops from the end
    extra = ctypes.c_int64(0)
    needs = []
    fresh = None
    for i in range(numWorkers):
        if i < len(pool):
            c = pool[i]
            c.check(c._L.shp_ctx_reserve_query(c.handle, dtcode, nBands, maxTilePx, ctypes.byref(extra),
                                               None, None))
            needs.append(extra.value)
        else:
            if fresh is None:                            # what a context without a workspace needs
                c = _lib.Context()
                try:
                    c.check(c._L.shp_ctx_reserve_query(c.handle, dtcode, nBands, maxTilePx,
                                                       ctypes.byref(extra), None, None))
                    fresh = extra.value
                finally:
                    c.close()
            needs.append(fresh)
    if not any(needs):
        return numWorkers                                # every workspace is already large enough
    free = ctypes.c_int64(0)
    c = _lib.ctx()
    c.check(c._L.shp_ctx_reserve_query(c.handle, dtcode, nBands, 0, None, ctypes.byref(free), None))
    budget = int(free.value * 0.92)
    fit = 0
    for need in needs:
        if need > budget:
            break
        budget -= need
        fit += 1
    fit = max(fit, 1)
    if fit < numWorkers:
        msg = ("pyshepseg_amd: %d of %d worker streams fit the free device memory for tiles of %.0f Mpx"
               % (fit, numWorkers, maxTilePx / 1e6))
        if verbose:
            print(msg)
        sys.stderr.write(msg + "\n")
    return fit


def startSegmentationWorkers(src, jobs, d_tiles, centres, msd, imgNullVal, fourConnected,
        minSegmentSize, numWorkers, timings, bands=None, yOrigin=0, maxConcurrentReads=20,
        verbose=False, stitchPrep=None, rowGate=None):
    """Start `numWorkers` threads, each with a pooled HIP context (one stream), that segment
    the jobs' windows of `src` (tile window rows are relative to yOrigin when src holds only a
    slice of the raster) into the device label block d_tiles.  Mirrors SegThreadsMgr.worker
    (reference tiling.py:1560-1600).  Returns (threads, forceExit event)."""
    L = _lib.lib()
    nullFlag = int(imgNullVal is not None)
    nullV = 0 if imgNullVal is None else int(imgNullVal)
    onDevice = isinstance(src, DeviceRaster)
    nBandsAll = src.shape[0]
    if bands is None:
        bands = list(range(nBandsAll))
    if onDevice and list(bands) != list(range(nBandsAll)):
        raise PyShepSegTilingError("band selection on a DeviceRaster is not supported")
    dtcode = _lib.SHP_DTYPES[numpy.dtype(src.dtype)] if onDevice else None
    (srcYsize, srcXsize) = (src.shape[1], src.shape[2])
    readSem = threading.BoundedSemaphore(max(1, maxConcurrentReads))
    # longest-processing-time-first: the edge tiles are up to 2.3x larger; starting them first
    # avoids a tail where a few workers finish the big ones alone (the stitch consumes in
    # row-major order regardless and its sequential part is thin)
    order = sorted(range(len(jobs)), key=lambda i: (-(jobs[i].xsize * jobs[i].ysize), i))
    if os.environ.get('SHEPSEG_TILE_ORDER', 'lpt') != 'lpt' or rowGate is not None:
        order = list(range(len(jobs)))           # (a raster still streaming in: tiles in arrival order)
    inQue = queue.Queue()
    for i in order:
        inQue.put(jobs[i])
    forceExit = threading.Event()
    clusMap = None
    if onDevice and jobs and os.environ.get('SHEPSEG_CLUSTER_MAP', '1') != '0':
        clusMap = _ClusterMap(srcYsize, srcXsize, max(1, numWorkers))

    def worker():
        try:
            with _lib.pooled_ctx() as c:
                worker_loop(c)
        except Exception as e:          # no GPU, library missing ...
            startLine.abort()
            forceExit.set()
            for jj in jobs:
                if jj.error is None and not jj.done.is_set():
                    jj.error = e
                jj.done.set()
        finally:
            if clusMap is not None:
                clusMap.workerExit()

    maxTilePx = max([jj.xsize * jj.ysize for jj in jobs] or [0])

    def worker_loop(c):
        if onDevice and maxTilePx > 0:
            # size the pooled context for the job's largest tile now: a grow-only workspace that
            # regrows when it first meets that tile stalls the device in the middle of the run
            c.check(L.shp_ctx_reserve(c.handle, dtcode, nBandsAll, maxTilePx))
        try:
            startLine.wait(timeout=300.0)
        except threading.BrokenBarrierError:
            pass                            # (a worker failed to start: forceExit is set, or the wait timed out)
        while not forceExit.is_set():
            try:
                j = inQue.get_nowait()
            except queue.Empty:
                break
            try:
                mx = ctypes.c_uint32(0)
                s1 = ctypes.c_int64(0)
                s2 = ctypes.c_int64(0)
                ncl = ctypes.c_uint32(0)
                dseg = ctypes.c_void_p(d_tiles.value + 4 * j.offset)
                if onDevice:
                    if rowGate is not None:
                        need = j.ypos - yOrigin + j.ysize
                        if clusMap is not None:         # the cluster map is assigned in whole blocks
                            B = clusMap.BLOCK
                            need = min(srcYsize, ((need + B - 1) // B) * B)
                        rowGate.waitRows(need, forceExit)
                    with timings.interval('segmentation'):
                        dclus = None
                        if clusMap is not None:
                            dclus = clusMap.ensureWindow(
                                c, j.xpos, j.ypos - yOrigin, j.xsize, j.ysize, forceExit,
                                lambda rects: c.check(L.shp_assign_rects_dev(
                                    c.handle, ctypes.c_void_p(src.ptr), dtcode, nBandsAll, srcYsize,
                                    srcXsize, _lib.ptr(rects), rects.shape[0], _lib.ptr(centres),
                                    centres.shape[0], nullFlag, nullV, clusMap.dptr)))
                        c.check(L.shp_segment_window_dev(
                            c.handle, ctypes.c_void_p(src.ptr), dtcode, nBandsAll, srcYsize,
                            srcXsize, j.xpos, j.ypos - yOrigin, j.xsize, j.ysize, _lib.ptr(centres),
                            centres.shape[0], nullFlag, nullV, int(bool(fourConnected)),
                            int(minSegmentSize), float(msd), dseg, ctypes.byref(mx),
                            ctypes.byref(s1), ctypes.byref(s2), ctypes.byref(ncl), dclus))
                else:
                    with timings.interval('reading'):
                        with readSem:
                            img = src.read(bands, j.xpos, j.ypos - yOrigin, j.xsize, j.ysize)
                    img, dt = _lib.as_image(img)
                    with timings.interval('segmentation'):
                        c.check(L.shp_segment_tile_to_dev(
                            c.handle, _lib.ptr(img), dt, img.shape[0], j.ysize, j.xsize,
                            _lib.ptr(centres), centres.shape[0], nullFlag, nullV,
                            int(bool(fourConnected)), int(minSegmentSize), float(msd), dseg,
                            ctypes.byref(mx), ctypes.byref(s1), ctypes.byref(s2),
                            ctypes.byref(ncl)))
                j.maxLocal = mx.value
                if stitchPrep is not None:
                    # the tile-local part of the stitch, off the sequential chain
                    (tileInfo, overlapSize, arena, simple) = stitchPrep
                    (top, bottom, left, right, _x, _y) = trimmedWindow(
                        tileInfo, j.col, j.row, j.xpos, j.ypos, j.xsize, j.ysize, overlapSize)
                    j.meta = arena.alloc(mx.value + 1, c)
                    cross = (ctypes.c_uint32 * 2)(0, 0)
                    c.check(L.shp_stitch_prepare_dev(
                        c.handle, dseg, j.ysize, j.xsize, overlapSize,
                        int(j.row > 0 and not simple), int(j.col > 0 and not simple), mx.value,
                        top, bottom, left, right, ctypes.c_void_p(j.meta), cross))
                    j.crossPx = (int(cross[0]), int(cross[1]))
                if verbose:
                    print("Tile ({}, {}): {} segments".format(j.col, j.row, mx.value))
            except Exception as e:
                j.error = e
                forceExit.set()
            j.done.set()

    nThreads = max(1, numWorkers)
    if onDevice and maxTilePx > 0:
        nThreads = _workersThatFit(nThreads, dtcode, nBandsAll, maxTilePx, verbose)
        if clusMap is not None:
            clusMap.workersLeft = nThreads
    # every worker sizes its workspace BEFORE any of them launches a kernel: device allocations and frees
    # (a pooled context regrowing for a larger tile) then never run beside the tiles' kernels
    startLine = threading.Barrier(nThreads)
    with timings.interval('startworkers'):
        threads = [threading.Thread(target=worker, daemon=True) for _ in range(nThreads)]
        for t in threads:
            t.start()
    return threads, forceExit


def waitForTile(j, jobs, threads, forceExit, timeout):
    """Block until tile job j is segmented; raise like the reference on failure / timeout
    (tiling.py:1045-1053, :918-928)."""
    t0 = time.time()
    timedOut = False
    while not j.done.wait(timeout=0.2):         # short slices: a failed worker is noticed at once
        if forceExit.is_set():
            break
        if not any(t.is_alive() for t in threads):
            break
        if timeout is not None and time.time() - t0 > timeout:
            timedOut = True
            break
    if j.error is not None or not j.done.is_set():
        forceExit.set()
        if timedOut and j.error is None:
            raise PyShepSegTilingError("Timeout waiting for tile ({}, {}) after {} seconds".format(
                j.col, j.row, timeout))
        err = j.error
        for jj in jobs:
            err = err or jj.error
        if isinstance(err, _lib.ShepsegHipError):
            raise err
        raise PyShepSegTilingError("Tile ({}, {}) failed: {}".format(j.col, j.row, err))


def doTiledShepherdSegmentation(infile, outfile, tileSize=DFLT_TILESIZE,
        overlapSize=DFLT_OVERLAPSIZE, minSegmentSize=50, numClusters=60,
        bandNumbers=None, subsamplePcnt=None, maxSpectralDiff='auto',
        imgNullVal=None, fixedKMeansInit=False, fourConnected=True, verbose=False,
        simpleTileRecode=False, outputDriver='KEA', creationOptions=[],
        spectDistPcntile=50, kmeansObj=None, tempfilesDriver='KEA',
        tempfilesExt='kea', tempfilesCreationOptions=[], writeHistogram=True,
        returnGDALDS=False, concurrencyCfg=None):
    """
    Run the Shepherd segmentation algorithm in a memory-efficient manner suitable for large
    rasters: one global k-means, overlapping tiles segmented independently, tiles stitched by
    matching segments across the overlap midline.  Arguments as the reference
    (tiling.py:446-571); the temp-file arguments are accepted and ignored (tile labels stay in
    GPU memory).  Returns a :class:`TiledSegmentationResult`.
    """
    if concurrencyCfg is None:
        concurrencyCfg = SegmentationConcurrencyConfig()
    if (overlapSize % 2) != 0:
        raise PyShepSegTilingError("Overlap size must be an even number")     # tiling.py:746
    timings = Timers()
    ioT0 = time.time()

    def ioMark(what):
        if os.environ.get('SHEPSEG_IO_TIMING'):
            sys.stderr.write('  [io] %-34s %.3f s\n' % (what, time.time() - ioT0))

    with timings.interval('walltime'):
        src = _open_source(infile)
        nBandsAll = src.shape[0]
        if bandNumbers is None:
            bandNumbers = list(range(1, nBandsAll + 1))
        bands = [b - 1 for b in bandNumbers]
        (inYsize, inXsize) = (src.RasterYSize, src.RasterXSize)
        # a host raster is streamed into HBM once, row block by row block, while the model is fitted
        # and the first tiles run (see _RasterStreamer); the per-tile read + upload path remains for
        # pixel types that need a conversion and for rasters that do not fit the device
        (streamer, devRas) = (None, None)
        workSrc = src
        if (not isinstance(src, DeviceRaster) and os.environ.get('SHEPSEG_STREAM_INPUT', '1') != '0' and
                numpy.dtype(src.dtype) in _lib.SHP_DTYPES and inYsize * inXsize > 0 and
                _rasterFitsDevice(len(bands), inYsize, inXsize, numpy.dtype(src.dtype).itemsize)):
            devRas = DeviceRaster(len(bands), inYsize, inXsize, src.dtype,
                                  imgNullVal if imgNullVal is not None else getattr(src, 'nullVal', None),
                                  cached=True)
            streamer = _RasterStreamer(src, bands, devRas, timings)
            workSrc = devRas
            ioMark('stream started')

        with timings.interval('spectralclusters'):
            if kmeansObj is None:
                (kmeansObj, subsamplePcnt, imgNullVal) = fitSpectralClustersWholeFile(
                    src, bandNumbers, numClusters, subsamplePcnt, imgNullVal, fixedKMeansInit)
            elif imgNullVal is None:
                imgNullVal = (src.bandNull(bandNumbers) if isinstance(src, _GdalSource)
                              else src.nullVal)
        ioMark('model fitted')
        centres = numpy.ascontiguousarray(kmeansObj.cluster_centers_, dtype=numpy.float64)
        msd = shepseg.autoMaxSpectralDiff(kmeansObj, maxSpectralDiff, spectDistPcntile)
        if verbose:
            print("KMeans of whole raster", kmeansObj.n_clusters, "clusters; maxSpectralDiff", msd)

        tileInfo = getTilesForFile(src, tileSize, overlapSize)
        if verbose:
            print("Found {} tiles, with {} rows and {} cols".format(
                tileInfo.getNumTiles(), tileInfo.nrows, tileInfo.ncols))

        main = _lib.chain_ctx()          # high-priority stream: the stitch chain must not lag
        L = main._L
        # one device block for every tile's labels, one for the stitched raster
        jobs, total = makeTileJobs(tileInfo)
        jobmap = {(j.col, j.row): j for j in jobs}
        nbTiles, nbOut = max(total, 1) * 4, max(inYsize * inXsize, 1) * 4
        d_tiles = _devAlloc(main, nbTiles)
        d_out = _devAlloc(main, nbOut)
        d_scal = _devAlloc(main, 256)
        main.check(L.shp_dev_memset(main.handle, d_scal, 0, 256))
        nbStrips = max(layoutStrips(jobs, overlapSize), 1) * 4
        d_strips = _devAlloc(main, nbStrips)
        arena = _MetaArena(main, 16 * (total // 8 + 1024))
        forceExit = None
        threads = []
        ok = False
        writer = None
        try:
            numWorkers = 1
            if concurrencyCfg.concurrencyType != CONC_NONE:
                numWorkers = max(1, int(concurrencyCfg.numWorkers))
            threads, forceExit = startSegmentationWorkers(
                workSrc, jobs, d_tiles, centres, msd, imgNullVal, fourConnected, minSegmentSize,
                numWorkers, timings, bands=(None if streamer is not None else bands),
                maxConcurrentReads=concurrencyCfg.maxConcurrentReads,
                verbose=verbose, stitchPrep=(tileInfo, overlapSize, arena, bool(simpleTileRecode)),
                rowGate=streamer)
            # finished rows of the stitched raster stream out while the rest is still in the making
            writer = None
            segimg = None
            gdalOut = None
            if outfile is not _KEEP_ON_DEVICE and inYsize * inXsize > 0:
                if outfile is None:
                    segimg = numpy.empty((inYsize, inXsize), dtype=shepseg.SegIdType)
                    dest = segimg
                elif isinstance(outfile, str) and outfile.endswith('.npy'):
                    dest = _NpyRowWriter(outfile, inYsize, inXsize)
                else:
                    gdalOut = _createGdalOutput(outfile, inYsize, inXsize, infile, outputDriver, creationOptions)
                    dest = None
                if isinstance(dest, _NpyRowWriter):
                    sink = dest.writeRows
                elif dest is not None:
                    def sink(y0, y1, v, dest=dest):
                        dest[y0:y1] = v
                else:
                    # one GDAL dataset handle is not safe for concurrent calls (the SWIG layer drops
                    # the GIL, the KEA / GTiff drivers behind it keep unguarded state): the row
                    # blocks go to the band one at a time, as the reference's single WriteArray
                    # per tile does (tiling.py:1032-1034)
                    gdalLock = threading.Lock()

                    def sink(y0, y1, v, band=gdalOut[1], lock=gdalLock):
                        with lock:
                            band.WriteArray(v, 0, y0)
                writer = _OutputWriter(d_out, inYsize, inXsize, sink, timings,
                                       nCopiers=1 if gdalOut is not None else None)
            rowsWritten = 0
            ovLevels = overviewLevels(inXsize, inYsize) if outfile is not _KEEP_ON_DEVICE else []
            ovDev = []
            for lvl in ovLevels:
                (oh, ow) = ((inYsize + lvl - 1) // lvl, (inXsize + lvl - 1) // lvl)
                d = _devAlloc(main, oh * ow * 4)
                main.check(L.shp_dev_memset(main.handle, d, 0, oh * ow * 4))
                ovDev.append((lvl, d, oh, ow))

            if os.environ.get('SHEPSEG_CHAIN_TIMING'):
                # diagnostic: let every tile finish first, so that 'stitchtiles' times the bare chain
                for t in threads:
                    t.join()
            # ---- stitchTiles (tiling.py:950-1064): sequential, concurrent with the workers ----
            with timings.interval('stitchtiles'):
                for j in jobs:
                    waitForTile(j, jobs, threads, forceExit, concurrencyCfg.tileCompletionTimeout)
                    (top, bottom, left, right, xout, yout) = trimmedWindow(
                        tileInfo, j.col, j.row, j.xpos, j.ypos, j.xsize, j.ysize, overlapSize)
                    topB = leftB = None
                    (topPitch, leftPitch) = (0, 0)
                    if not simpleTileRecode:
                        if j.row > 0:
                            a = jobmap[(j.col, j.row - 1)]
                            topB = ctypes.c_void_p(d_strips.value + 4 * a.bottomOff)
                            topPitch = a.xsize
                        if j.col > 0:
                            a = jobmap[(j.col - 1, j.row)]
                            leftB = ctypes.c_void_p(d_strips.value + 4 * a.rightOff)
                            leftPitch = min(overlapSize, a.xsize)
                    rightOut = bottomOut = None
                    if j.col != tileInfo.ncols - 1:
                        rightOut = ctypes.c_void_p(d_strips.value + 4 * j.rightOff)
                    if j.row != tileInfo.nrows - 1:
                        bottomOut = ctypes.c_void_p(d_strips.value + 4 * j.bottomOff)
                    main.check(L.shp_stitch_chain_dev(
                        main.handle, ctypes.c_void_p(d_tiles.value + 4 * j.offset), j.ysize, j.xsize,
                        overlapSize, topB, topPitch, leftB, leftPitch, j.maxLocal,
                        int(bool(simpleTileRecode)), d_scal, top, bottom, left, right,
                        ctypes.c_void_p(j.meta), rightOut, bottomOut, d_out, inXsize, xout, yout,
                        j.crossPx[0], j.crossPx[1]))
                    for (lvl, d, oh, ow) in ovDev:
                        main.check(L.shp_overview_window_dev(main.handle, d_out, inXsize, xout, yout,
                                                             right - left, bottom - top, lvl, d, ow, oh))
                    if writer is not None and j.col == tileInfo.ncols - 1:
                        # the tile row is stitched: its output rows are final once the device is done
                        main.check(L.shp_sync(main.handle))
                        writer.rowsFinal(rowsWritten, yout + (bottom - top))
                        rowsWritten = yout + (bottom - top)
                main.check(L.shp_sync(main.handle))
            ioMark('last tile stitched')
            for t in threads:
                t.join()
            if writer is not None:
                writer.rowsFinal(rowsWritten, inYsize)
                writer.finish()
                writer = None
            ioMark('output written')

            scal = numpy.zeros(1, dtype=numpy.uint32)
            main.check(L.shp_dev_download(main.handle, _lib.ptr(scal), d_scal, 4))
            maxSegId = int(scal[0])
            hist = numpy.zeros(maxSegId + 1, dtype=numpy.uint32)
            ioMark('maxSegId read')
            main.check(L.shp_histogram_dev(main.handle, d_out, inYsize * inXsize, inXsize, maxSegId,
                                           _lib.ptr(hist)))
            ioMark('histogram')
            hasEmpty = bool((hist[1:] == 0).any())
            if hasEmpty:
                _warnEmptySegments(hist, overlapSize)

            result = TiledSegmentationResult()
            result.bandStatistics = estimateStatsFromHisto(hist) if hist.sum() > 0 else []
            ioMark('band statistics')
            result.overviews = {}
            for (lvl, d, oh, ow) in ovDev:
                a = numpy.empty((oh, ow), dtype=shepseg.SegIdType)
                main.check(L.shp_dev_download(main.handle, _lib.ptr(a), d, a.nbytes))
                result.overviews[lvl] = a
            if outfile is _KEEP_ON_DEVICE:
                result.outDev = (d_out.value, inYsize, inXsize, nbOut)
                d_out = None                        # ownership moves to the caller
            elif outfile is None:
                result.segimg = segimg if segimg is not None else numpy.zeros((inYsize, inXsize), shepseg.SegIdType)
            elif isinstance(outfile, str) and outfile.endswith('.npy'):
                if inYsize * inXsize == 0:
                    numpy.save(outfile, numpy.zeros((inYsize, inXsize), shepseg.SegIdType))
                else:
                    dest.close()
                if writeHistogram:
                    numpy.save(outfile[:-4] + '_hist.npy', hist)
                for (lvl, a) in result.overviews.items():
                    numpy.save(outfile[:-4] + '_ov%d.npy' % lvl, a)
            elif gdalOut is not None:
                _finishGdalOutput(gdalOut, hist, writeHistogram, result.overviews, result.bandStatistics)
            ok = True
        finally:
            # no buffer goes back to the cache (or to hipFree) while a worker may still write to
            # it: workers only test forceExit between tiles, so wait for them and for the chain
            if forceExit is not None:
                forceExit.set()
            if streamer is not None:
                streamer.close()
            if writer is not None:              # (failure path: the queue is abandoned)
                try:
                    writer.finish()
                except Exception:
                    pass
            stuck = bool(streamer is not None and getattr(streamer, 'stuck', False))
            for t in threads:
                t.join(timeout=None if ok else 120.0)
                stuck = stuck or t.is_alive()
            L.shp_sync(main.handle)
            if stuck:
                sys.stderr.write("pyshepseg_amd: a worker did not stop after a failure; its device "
                                 "buffers are leaked rather than reused\n")
            elif ok:
                _devRelease(main, d_tiles, nbTiles)
                _devRelease(main, d_out, nbOut)
                _devRelease(main, d_scal, 256)
                _devRelease(main, d_strips, nbStrips)
                arena.release()
            else:           # failed run: free, do not recycle
                for (p_, n_) in ((d_tiles, nbTiles), (d_out, nbOut), (d_scal, 256), (d_strips, nbStrips)):
                    if p_ is not None and p_.value:
                        L.shp_dev_free(main.handle, p_)
                arena.free()
            if devRas is not None and not stuck:
                devRas.free()
            if not stuck:
                for (_lvl, d, oh, ow) in (ovDev if 'ovDev' in locals() else []):
                    _devRelease(main, d, oh * ow * 4)

    result.maxSegId = maxSegId
    result.numTileRows = tileInfo.nrows
    result.numTileCols = tileInfo.ncols
    result.subsamplePcnt = subsamplePcnt
    result.maxSpectralDiff = msd
    result.kmeans = kmeansObj
    result.hasEmptySegments = hasEmpty
    result.hist = hist
    result.timings = timings
    return result


def overviewLevels(inXsize, inYsize):
    """The overview (pyramid) levels of an output raster: 4, 8, 16 ... the last one being the first
    whose layer is smaller than 1024 pixels on the raster's larger side; none below 4096 pixels
    (the rule of the reference's setupOverviews, tiling.py:1385-1404)."""
    outSize = max(int(inXsize), int(inYsize))
    levels = []
    lvl = 4
    if outSize // lvl >= 1024:
        levels.append(lvl)
        while outSize // lvl >= 1024:
            lvl *= 2
            levels.append(lvl)
    return levels


def estimateStatsFromHisto(hist):
    """The band statistics the reference derives from the segment histogram and stores as GDAL
    metadata (utils.estimateStatsFromHisto, utils.py:47-95): list of (item name, string value) in
    the reference's order, values formatted as it formats them (ints for the thematic band)."""
    hist = numpy.asarray(hist)
    if hist.dtype == numpy.uint32 and hist.ndim == 1 and hist.size > 0:
        # the same evaluation in one pass of compiled code (numpy's sums and their order restated, see
        # shp_hist_stats): a dozen numpy passes over a few million bins were 22 ms at the very end of a run
        h = numpy.ascontiguousarray(hist)
        out = numpy.zeros(6, dtype=numpy.float64)
        rc = _lib.lib().shp_hist_stats(_lib.ptr(h), h.size, _lib.ptr(out))
        if rc != 0:
            raise RuntimeError("shp_hist_stats failed (%d)" % rc)
        (minVal, maxVal, meanVal, stdDevVal, modeVal, medianVal) = out
        return [("STATISTICS_MINIMUM", repr(int(minVal))), ("STATISTICS_MAXIMUM", repr(int(maxVal))),
                ("STATISTICS_MEAN", repr(float(meanVal))), ("STATISTICS_STDDEV", repr(float(stdDevVal))),
                ("STATISTICS_MODE", repr(int(modeVal))), ("STATISTICS_MEDIAN", repr(int(medianVal))),
                ("STATISTICS_SKIPFACTORX", "1"), ("STATISTICS_SKIPFACTORY", "1"),
                ("STATISTICS_HISTOBINFUNCTION", "direct")]
    mask = hist > 0
    nVals = hist.sum()
    minVal = mask.argmax()
    maxVal = hist.shape[0] - numpy.flip(mask).argmax() - 1
    values = numpy.arange(hist.shape[0])
    meanVal = (values * hist).sum() / nVals
    stdDevVal = numpy.sqrt((hist * numpy.power(values - meanVal, 2)).sum() / nVals)
    modeVal = numpy.argmax(hist)
    medianVal = (hist.cumsum() >= hist.sum() / 2).nonzero()[0][0]
    return [("STATISTICS_MINIMUM", repr(int(minVal))), ("STATISTICS_MAXIMUM", repr(int(maxVal))),
            ("STATISTICS_MEAN", repr(float(meanVal))), ("STATISTICS_STDDEV", repr(float(stdDevVal))),
            ("STATISTICS_MODE", repr(int(modeVal))), ("STATISTICS_MEDIAN", repr(int(medianVal))),
            ("STATISTICS_SKIPFACTORX", "1"), ("STATISTICS_SKIPFACTORY", "1"),
            ("STATISTICS_HISTOBINFUNCTION", "direct")]


_KEEP_ON_DEVICE = object()      # outfile sentinel used by bench.py: labels stay in HBM


def freeDeviceOutput(result):
    """Release the device raster kept by outfile=_KEEP_ON_DEVICE."""
    od = getattr(result, 'outDev', None)
    if od and od[0]:
        _devRelease(_lib.ctx(), ctypes.c_void_p(od[0]), od[3])
        result.outDev = None


def _warnEmptySegments(hist, overlapSize):
    """Same warning as reference checkForEmptySegments (tiling.py:1308-1341).  NB the reference
    method has no return statement, so its hasEmptySegments is always None; here the flag is the
    real boolean (a knowing fix)."""
    emptySegIds = numpy.where(hist[1:] == 0)[0] + 1
    msg = [
        "",
        "WARNING: Found {} segments with zero pixels".format(len(emptySegIds)),
        "    Segment IDs: {}".format(emptySegIds),
        "    This is caused by inconsistent joining of segmentation",
        "    tiles, and will probably cause trouble later on.",
        "    It is highly recommended to re-run with a larger overlap",
        "    size (currently {}), and if necessary a larger tile size".format(overlapSize),
        ""
    ]
    print('\n'.join(msg), file=sys.stderr)


def _rasterFitsDevice(nBands, nRows, nCols, itemsize):
    """Is there room in HBM for the raster beside what a tiled run needs (cluster map, tile labels,
    strips, stitched output: ~20 B per pixel) and a few worker workspaces?"""
    c = _lib.ctx()
    free = ctypes.c_int64(0)
    c.check(c._L.shp_ctx_reserve_query(c.handle, 2, 1, 0, None, ctypes.byref(free), None))
    npx = nRows * nCols
    cached = _devCacheBytes[0]
    return (free.value + cached) * 0.9 > npx * (nBands * itemsize + 20) + (8 << 30)


def _createGdalOutput(outfile, ys, xs, infile, outputDriver, creationOptions):
    """The output dataset (thematic uint32, georeferenced like the input; tiling.py:961-975);
    returns (dataset, band)."""
    try:
        from osgeo import gdal
    except ImportError:
        raise PyShepSegTilingError("cannot write %r: GDAL (osgeo) is not importable here; use "
                                   "outfile=None or a .npy path" % (outfile,))
    drvr = gdal.GetDriverByName(outputDriver)
    if drvr is None:
        raise PyShepSegTilingError("This GDAL does not support driver '{}'".format(outputDriver))
    if os.path.exists(outfile):                    # (tiling.py:960-962)
        gdal.IdentifyDriver(outfile).Delete(outfile)
    ds = drvr.Create(outfile, xs, ys, 1, gdal.GDT_UInt32, creationOptions)
    if isinstance(infile, str):
        inDs = gdal.Open(infile)
        ds.SetProjection(inDs.GetProjection())
        ds.SetGeoTransform(inDs.GetGeoTransform())
    band = ds.GetRasterBand(1)
    band.SetMetadataItem('LAYER_TYPE', 'thematic')
    band.SetNoDataValue(shepseg.SEGNULLVAL)
    return (ds, band)


def _finishGdalOutput(gdalOut, hist, writeHistogram, overviews, bandStatistics):
    """Overview layers (tiling.py:1360-1404), RAT 'Histogram' column (tiling.py:1343-1358), band
    statistics metadata (utils.py:47-95) and flush."""
    from osgeo import gdal
    (ds, band) = gdalOut
    if overviews:
        levels = sorted(overviews)
        ds.BuildOverviews("NEAREST", levels)
        for (j, lvl) in enumerate(levels):
            band.GetOverview(j).WriteArray(overviews[lvl], 0, 0)
    for (k, v) in bandStatistics:
        band.SetMetadataItem(k, v)
    if writeHistogram:                              # writeHistogramToFile, tiling.py:1343-1358
        rat = band.GetDefaultRAT()
        if rat.GetRowCount() != len(hist):
            rat.SetRowCount(len(hist))
        colNum = rat.GetColOfUsage(gdal.GFU_PixelCount)
        if colNum == -1:
            rat.CreateColumn('Histogram', gdal.GFT_Real, gdal.GFU_PixelCount)
            colNum = rat.GetColumnCount() - 1
        rat.WriteArray(hist.astype(numpy.float64), colNum)
    ds.FlushCache()
