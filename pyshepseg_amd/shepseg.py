"""Drop-in for ``pyshepseg.shepseg`` with the hot path on MI355X HIP kernels.

Same names, arguments and results as the reference module (pyshepseg/shepseg.py); the
arithmetic that the reference runs in numba ``@njit`` functions and in scikit-learn's
``KMeans`` runs in ``libshepseg_hip.so`` through the ctypes C-ABI of
``include/shepseg_hip.h``.  Host code here is plain Python + numpy: argument handling, the
sample selection and initial centres of ``fitSpectralClusters`` and ``autoMaxSpectralDiff``
(tiny, reference shepseg.py:283-310, :364-449).  No numba, scikit-learn, torch or GDAL import.
There is no CPU fallback: without the HIP library or a GPU every compute call raises.
"""
import ctypes
import os
import time

import numpy

from . import _lib
from ._lib import ShepsegHipError  # noqa: F401  (re-exported)

# A symbol for the data type used as a segment ID number (reference shepseg.py:97-101)
SegIdType = numpy.uint32
SEGNULLVAL = 0
MINSEGID = SEGNULLVAL + 1


class SegmentationResult(object):
    """Results of the segmentation process (reference shepseg.py:104-127)."""
    def __init__(self):
        self.segimg = None
        self.kmeans = None
        self.maxSpectralDiff = None
        self.singlePixelsEliminated = None
        self.smallSegmentsEliminated = None


class KMeansModel(object):
    """Fitted k-means model; stands where the reference has a ``sklearn.cluster.KMeans``.

    Exposes what the reference and its callers read: ``cluster_centers_`` (shepseg.py:434,
    cmdline/run_seg.py:210), ``predict`` (shepseg.py:350), ``n_iter_``, ``labels_``,
    ``inertia_``.  Plain attributes only, so it pickles (tiling.py:825).
    """
    def __init__(self, centres, n_iter=None, labels=None, inertia=None):
        self.cluster_centers_ = numpy.ascontiguousarray(centres, dtype=numpy.float64)
        self.n_clusters = self.cluster_centers_.shape[0]
        self.n_iter_ = n_iter
        self.labels_ = labels
        self.inertia_ = inertia

    def predict(self, x):
        """Nearest-centre index (0-based) of every row of integer array x (N, nBands)."""
        x = numpy.asarray(x)
        img = numpy.ascontiguousarray(x.T).reshape(x.shape[1], x.shape[0], 1)
        return _assign(self.cluster_centers_, img, None).reshape(-1) - 1


def _centres_of(kmeansObj):
    c = getattr(kmeansObj, 'cluster_centers_', None)
    if c is None:
        raise ValueError("kmeansObj must be fitted (have cluster_centers_)")
    return numpy.ascontiguousarray(c, dtype=numpy.float64)


def doShepherdSegmentation(img, numClusters=60, clusterSubsamplePcnt=1,
        minSegmentSize=50, maxSpectralDiff='auto', imgNullVal=None,
        fourConnected=True, verbose=False, fixedKMeansInit=False,
        kmeansObj=None, spectDistPcntile=50):
    """
    Perform Shepherd segmentation in memory, on the given multi-band img array
    (nBands, nRows, nCols).  Signature and result as reference shepseg.py:130-249;
    the stages after the k-means fit run fused on the GPU (shp_segment_tile).
    """
    t0 = time.time()
    if kmeansObj is not None:
        km = kmeansObj
    else:
        km = fitSpectralClusters(img, numClusters, clusterSubsamplePcnt, imgNullVal,
                                 fixedKMeansInit)
    centres = _centres_of(km)
    maxSpectralDiff = autoMaxSpectralDiff(km, maxSpectralDiff, spectDistPcntile)
    if verbose:
        print("Kmeans, in", round(time.time() - t0, 1), "seconds")

    t0 = time.time()
    img_c, dt = _lib.as_image(img)
    (nBands, nRows, nCols) = img_c.shape
    if centres.shape[1] != nBands:
        raise ValueError("k-means centres have %d bands, image has %d" % (centres.shape[1], nBands))
    seg = numpy.empty((nRows, nCols), dtype=SegIdType)
    c = _lib.ctx()
    maxSegId = ctypes.c_uint32(0)
    nSingle = ctypes.c_int64(0)
    nSmall = ctypes.c_int64(0)
    nClumps = ctypes.c_uint32(0)
    c.check(c._L.shp_segment_tile(
        c.handle, _lib.ptr(img_c), dt, nBands, nRows, nCols, _lib.ptr(centres),
        centres.shape[0], int(imgNullVal is not None),
        0 if imgNullVal is None else int(imgNullVal), int(bool(fourConnected)),
        int(minSegmentSize), float(maxSpectralDiff), _lib.ptr(seg), ctypes.byref(maxSegId),
        ctypes.byref(nSingle), ctypes.byref(nSmall), ctypes.byref(nClumps)))
    if verbose:
        print("Found", nClumps.value, "clumps")
        print("Eliminated", nSingle.value, "single pixels")
        print("Eliminated", nSmall.value, "segments, in", round(time.time() - t0, 1), "seconds")
        print("Final result has", maxSegId.value, "segments")

    segResult = SegmentationResult()
    segResult.segimg = seg
    segResult.kmeans = km
    segResult.maxSpectralDiff = maxSpectralDiff
    segResult.singlePixelsEliminated = nSingle.value
    segResult.smallSegmentsEliminated = nSmall.value
    segResult.timings = c.timings()
    return segResult


def _sample_rows(img, subsamplePcnt, imgNullVal, wantMinMax=False):
    """The rows sklearn is fitted on in the reference (shepseg.py:283-299): non-null pixels
    in raster order, every skip-th one.  With wantMinMax also the per-band (min, max) of those
    rows, taken on the band-planar form where the reduction runs over contiguous memory."""
    (nBands, nRows, nCols) = img.shape
    flat = img.reshape(nBands, nRows * nCols)
    skip = int(round(100. / subsamplePcnt))
    if imgNullVal is not None:
        nonNull = (flat != imgNullVal).all(axis=0)
        idx = numpy.flatnonzero(nonNull)[::skip]
        planar = flat[:, idx]
    elif skip == 1:
        planar = flat
    else:
        planar = flat[:, ::skip]
    xSample = numpy.ascontiguousarray(planar.T)
    if wantMinMax:
        if planar.shape[1] == 0:
            return xSample, None
        return xSample, (planar.min(axis=1), planar.max(axis=1))
    return xSample


def _kmeans_plusplus(x, k, rng):
    """k-means++ seeding with sklearn's greedy local trials (no parity definition: the
    reference leaves it randomly seeded, shepseg.py:305-311)."""
    n = x.shape[0]
    centres = numpy.empty((k, x.shape[1]), dtype=numpy.float64)
    ntrials = 2 + int(numpy.log(k))
    centres[0] = x[rng.randint(n)]
    closest = ((x - centres[0]) ** 2).sum(axis=1)
    pot = closest.sum()
    for c in range(1, k):
        r = rng.random_sample(ntrials) * pot
        cand = numpy.searchsorted(numpy.cumsum(closest), r)
        numpy.clip(cand, None, n - 1, out=cand)
        best = None
        for ci in cand:
            d = numpy.minimum(closest, ((x - x[ci]) ** 2).sum(axis=1))
            p = d.sum()
            if best is None or p < best[0]:
                best = (p, ci, d)
        pot, ci, closest = best
        centres[c] = x[ci]
    return centres


def _fit(xSample, init, max_iter=300, tol=1e-4, wantInertia=False):
    xSample = numpy.asarray(xSample)
    typed = xSample.dtype in _lib.SHP_DTYPES         # pixel types go down as they are
    x = numpy.ascontiguousarray(xSample, dtype=None if typed else numpy.float64)
    init = numpy.ascontiguousarray(init, dtype=numpy.float64)
    (n, nb) = x.shape
    k = init.shape[0]
    centres = numpy.empty((k, nb), dtype=numpy.float64)
    labels = numpy.empty(n, dtype=numpy.int32)
    nit = ctypes.c_int(0)
    c = _lib.ctx()
    if typed:
        c.check(c._L.shp_kmeans_fit_typed(c.handle, _lib.ptr(x), _lib.SHP_DTYPES[x.dtype], n, nb, k,
                                          _lib.ptr(init), int(max_iter), float(tol),
                                          _lib.ptr(centres), _lib.ptr(labels), ctypes.byref(nit)))
    else:
        c.check(c._L.shp_kmeans_fit(c.handle, _lib.ptr(x), n, nb, k, _lib.ptr(init), int(max_iter),
                                    float(tol), _lib.ptr(centres), _lib.ptr(labels),
                                    ctypes.byref(nit)))
    inertia = float(((x - centres[labels]) ** 2).sum()) if wantInertia else None
    km = KMeansModel(centres, nit.value, labels, inertia)
    km.fit_path_ = ('lloyd', 'elkan')[c._L.shp_last_fit_path(c.handle)]
    return km


def _fit_planar(img, numClusters, imgNullVal, init=None, max_iter=300, tol=1e-4, commHandle=None):
    """commHandle: an RCCL communicator of the C-ABI (comm.RcclComm.h) -- the E-step is then sharded by sample
    rows over its ranks, every one of which must make this call with the same sample (shp_kmeans_fit_planar_dist)"""
    img = numpy.ascontiguousarray(img)
    nb = img.shape[0]
    npix = int(img.size // max(nb, 1))
    centres = numpy.empty((numClusters, nb), dtype=numpy.float64)
    labels = numpy.empty(max(npix, 1), dtype=numpy.int32)
    nit = ctypes.c_int(0)
    nrows = ctypes.c_int64(0)
    if init is not None:
        init = numpy.ascontiguousarray(init, dtype=numpy.float64)
    c = _lib.ctx()
    args = (_lib.ptr(img), _lib.SHP_DTYPES[img.dtype], npix, nb, int(imgNullVal is not None),
            0 if imgNullVal is None else int(imgNullVal), int(numClusters),
            None if init is None else _lib.ptr(init), int(max_iter), float(tol), _lib.ptr(centres),
            _lib.ptr(labels), ctypes.byref(nit), ctypes.byref(nrows))
    if commHandle is not None:
        c.check(c._L.shp_kmeans_fit_planar_dist(c.handle, commHandle, *args))
    else:
        c.check(c._L.shp_kmeans_fit_planar(c.handle, *args))
    km = KMeansModel(centres, nit.value, labels[:nrows.value], None)
    km.fit_path_ = ('lloyd', 'elkan')[c._L.shp_last_fit_path(c.handle)]
    return km


def fitSpectralClusters(img, numClusters, subsamplePcnt, imgNullVal, fixedKMeansInit, _commHandle=None):
    """First step of Shepherd segmentation: k-means on a subsample of the pixels
    (reference shepseg.py:252-314).  Lloyd iterations run on the GPU (shp_kmeans_fit).
    Returns a fitted :class:`KMeansModel`.  (_commHandle: the sharded driver's RCCL communicator, see _fit_planar.)"""
    img = numpy.asarray(img)
    if (fixedKMeansInit and img.ndim == 3 and img.dtype in _lib.SHP_DTYPES and
            int(round(100. / subsamplePcnt)) == 1 and os.environ.get('SHEPSEG_FIT_PLANAR', '1') != '0'):
        # every pixel of img is a sample: the band-planar array goes down as it is (null rows are
        # dropped, the diagonal initial centres taken and the sample centred by one host thread per
        # band inside the library: the same arithmetic as the row form below, no transposition)
        return _fit_planar(img, numClusters, imgNullVal, commHandle=_commHandle)
    xSample, minmax = _sample_rows(img, subsamplePcnt, imgNullVal, wantMinMax=True)
    if fixedKMeansInit:
        init = diagonalClusterCentres(xSample, numClusters, minmax)
        return _fit(xSample, init)
    best = None
    rng = numpy.random.RandomState()
    xs = xSample.astype(numpy.float64)
    for _trial in range(5):                    # numKmeansTrials (shepseg.py:305)
        km = _fit(xSample, _kmeans_plusplus(xs, numClusters, rng), wantInertia=True)
        if best is None or km.inertia_ < best.inertia_:
            best = km
    return best


def _assign(centres, img, imgNullVal):
    img_c, dt = _lib.as_image(img)
    (nBands, nRows, nCols) = img_c.shape
    centres = numpy.ascontiguousarray(centres, dtype=numpy.float64)
    if centres.shape[1] != nBands:
        raise ValueError("k-means centres have %d bands, image has %d" % (centres.shape[1], nBands))
    out = numpy.empty((nRows, nCols), dtype=numpy.int32)
    c = _lib.ctx()
    c.check(c._L.shp_kmeans_assign(c.handle, _lib.ptr(img_c), dt, nBands, nRows, nCols,
                                   _lib.ptr(centres), centres.shape[0],
                                   int(imgNullVal is not None),
                                   0 if imgNullVal is None else int(imgNullVal), _lib.ptr(out)))
    return out


def applySpectralClusters(kmeansObj, img, imgNullVal):
    """Cluster id (1..k, 0 = null) of every pixel (reference shepseg.py:317-361)."""
    return _assign(_centres_of(kmeansObj), img, imgNullVal)


def diagonalClusterCentres(xSample, numClusters, _minmax=None):
    """Initial centres evenly spaced along the diagonal of the data's bounding box, cast to
    the sample's integer dtype (reference shepseg.py:364-397).  _minmax: per-band (min, max)
    of xSample when the caller already has them."""
    if _minmax is not None:
        (bandMin, bandMax) = _minmax
    else:
        bandMin = xSample.min(axis=0)
        bandMax = xSample.max(axis=0)
    # centre i = min + (i + 1) * (max - min) / (k + 1) per band, truncated to the sample's dtype
    step = (bandMax - bandMin) / (numClusters + 1)
    ramp = numpy.arange(1, numClusters + 1, dtype=numpy.float64)
    return (bandMin + numpy.multiply.outer(ramp, step)).astype(xSample.dtype)


def _percentile_linear_f64(a32, q):
    """numpy.percentile(float32 array, q) as numpy 1.26 evaluates it: 'linear' method,
    float64 virtual index and gamma, lerp in float64 on a float32 difference, float64 result.
    (numpy >= 2 returns float32 here; the oracle stack is pinned to 1.26)"""
    arr = numpy.sort(numpy.asarray(a32, dtype=numpy.float32))
    n = arr.shape[0]
    vidx = (n - 1) * (numpy.float64(q) / numpy.float64(100))
    prev = int(numpy.floor(vidx))
    nxt = min(prev + 1, n - 1)
    gamma = numpy.float64(vidx - prev)
    a = arr[prev]
    b = arr[nxt]
    diff = numpy.float64(numpy.float32(b - a))
    if gamma >= 0.5:
        return numpy.float64(b) - diff * (1 - gamma)
    return numpy.float64(a) + diff * gamma


def autoMaxSpectralDiff(km, maxSpectralDiff, distPcntile):
    """maxSpectralDiff to use: 'auto' = percentile of the pairwise centre distances, None =
    10 x the largest, a number = itself (reference shepseg.py:400-449)."""
    if not (maxSpectralDiff is None or (isinstance(maxSpectralDiff, str) and maxSpectralDiff == 'auto')):
        return maxSpectralDiff          # a number: the centre distances are not needed
    centres = _centres_of(km)
    numClusters = centres.shape[0]
    numPairs = numClusters * (numClusters - 1) // 2
    clusterDist = numpy.full(numPairs, -1, dtype=numpy.float32)
    k = 0
    for i in range(numClusters - 1):
        for j in range(i + 1, numClusters):
            clusterDist[k] = numpy.sqrt(((centres[i] - centres[j]) ** 2).sum())
            k += 1
    if isinstance(maxSpectralDiff, str) and maxSpectralDiff == 'auto':
        maxSpectralDiff = _percentile_linear_f64(clusterDist, distPcntile)
    elif maxSpectralDiff is None:
        # numpy 1.26 (oracle stack) evaluates int * float32-scalar in float64
        maxSpectralDiff = 10 * numpy.float64(clusterDist.max())
    return maxSpectralDiff


def clump(img, ignoreVal, fourConnected=True, clumpId=1):
    """Connected components of equal values with the reference's 10000-pixel depth-first cut
    (reference shepseg.py:452-541).  Returns (clumpimg uint32, next id)."""
    img = numpy.asarray(img)
    (nRows, nCols) = img.shape
    codes = img
    if ignoreVal != 0 or img.size == 0 or int(img.min()) < 0 or int(img.max()) > 65535:
        vals, inv = numpy.unique(img, return_inverse=True)
        if vals.shape[0] > 65535:
            raise ValueError("clump supports at most 65535 distinct values")
        codes = (inv.reshape(img.shape) + 1).astype(numpy.int32)
        codes[img == ignoreVal] = 0
    codes = numpy.ascontiguousarray(codes, dtype=numpy.int32)
    out = numpy.empty((nRows, nCols), dtype=SegIdType)
    mx = ctypes.c_uint32(0)
    c = _lib.ctx()
    c.check(c._L.shp_clump(c.handle, _lib.ptr(codes), nRows, nCols, int(bool(fourConnected)),
                           _lib.ptr(out), ctypes.byref(mx)))
    if clumpId != 1:
        out[out != 0] += SegIdType(clumpId - 1)
    return (out, int(mx.value) + int(clumpId))


def makeSegSize(seg):
    """Histogram of segment ids, length seg.max()+1 (reference shepseg.py:544-569)."""
    seg = numpy.ascontiguousarray(seg, dtype=SegIdType)
    maxSegId = int(seg.max()) if seg.size else 0
    out = numpy.empty(maxSegId + 1, dtype=numpy.uint32)
    c = _lib.ctx()
    c.check(c._L.shp_make_seg_size(c.handle, _lib.ptr(seg), seg.size, maxSegId, _lib.ptr(out)))
    return out


def eliminateSinglePixels(img, seg, segSize, minSegId, maxSegId, fourConnected):
    """Merge single-pixel segments into their spectrally nearest neighbouring pixel's
    segment and relabel; seg is modified in place (reference shepseg.py:572-615).
    segSize is recomputed on the device from seg (the reference's copy is stale on return)."""
    img_c, dt = _lib.as_image(img)
    (nBands, nRows, nCols) = img_c.shape
    if seg.dtype != SegIdType or not seg.flags.c_contiguous:
        raise TypeError("seg must be a C-contiguous uint32 array")
    mx = ctypes.c_uint32(int(maxSegId))
    c = _lib.ctx()
    c.check(c._L.shp_eliminate_single(c.handle, _lib.ptr(img_c), dt, nBands, nRows, nCols,
                                      int(bool(fourConnected)), _lib.ptr(seg), ctypes.byref(mx)))


def eliminateSmallSegments(seg, img, maxSegId, minSegSize, maxSpectralDiff, fourConnected,
        minSegId):
    """Iteratively merge segments smaller than minSegSize into their spectrally closest
    larger neighbour; seg modified in place; returns the number eliminated
    (reference shepseg.py:918-1000)."""
    img_c, dt = _lib.as_image(img)
    (nBands, nRows, nCols) = img_c.shape
    if seg.dtype != SegIdType or not seg.flags.c_contiguous:
        raise TypeError("seg must be a C-contiguous uint32 array")
    mx = ctypes.c_uint32(int(maxSegId))
    ne = ctypes.c_int64(0)
    c = _lib.ctx()
    c.check(c._L.shp_eliminate_small(c.handle, _lib.ptr(img_c), dt, nBands, nRows, nCols,
                                     int(bool(fourConnected)), int(minSegSize),
                                     float(maxSpectralDiff), _lib.ptr(seg), ctypes.byref(mx),
                                     ctypes.byref(ne)))
    return int(ne.value)


class RowColArray(object):
    """Pixel coordinates of one segment, raster order (reference shepseg.py:816-870)."""
    def __init__(self, rowcols):
        self.rowcols = rowcols
        self.idx = rowcols.shape[0]

    def getSegmentIndices(self):
        return (self.rowcols[:, 0], self.rowcols[:, 1])


def makeSegmentLocations(seg, segSize):
    """dict: segment id -> RowColArray of its pixels in raster order (reference shepseg.py:880-915),
    read back from the device CSR the elimination stage builds (shp_segment_locations)."""
    seg = numpy.ascontiguousarray(seg, dtype=SegIdType)
    (nRows, nCols) = seg.shape
    maxSegId = len(segSize) - 1
    offs = numpy.zeros(maxSegId + 2, dtype=numpy.uint32)
    pix = numpy.empty(seg.size, dtype=numpy.uint32)
    c = _lib.ctx()
    c.check(c._L.shp_segment_locations(c.handle, _lib.ptr(seg), nRows, nCols, maxSegId,
                                       _lib.ptr(offs), _lib.ptr(pix)))
    rc = numpy.empty((seg.size, 2), dtype=numpy.uint32)
    numpy.floor_divide(pix, numpy.uint32(max(nCols, 1)), out=rc[:, 0])
    numpy.remainder(pix, numpy.uint32(max(nCols, 1)), out=rc[:, 1])
    d = {}
    for segid in range(MINSEGID, maxSegId + 1):
        d[SegIdType(segid)] = RowColArray(rc[offs[segid]:offs[segid + 1]])
    return d


def buildSegmentSpectra(seg, img, maxSegId):
    """float32 per-segment per-band sums accumulated in raster order (reference shepseg.py:780-813),
    computed by the device kernels of the elimination stage (shp_build_segment_spectra)."""
    img_c, dt = _lib.as_image(img)
    (nBands, nRows, nCols) = img_c.shape
    seg = numpy.ascontiguousarray(seg, dtype=SegIdType)
    spectSum = numpy.zeros((int(maxSegId) + 1, nBands), dtype=numpy.float32)
    c = _lib.ctx()
    c.check(c._L.shp_build_segment_spectra(c.handle, _lib.ptr(seg), _lib.ptr(img_c), dt, nBands, nRows,
                                           nCols, int(maxSegId), _lib.ptr(spectSum)))
    return spectSum
