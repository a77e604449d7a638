"""Drop-in for the non-spatial part of ``pyshepseg.tilingstats``: per-segment statistics of an
image band against a segmentation raster, computed on the GPU.

``calcPerSegmentStatsTiled(imgfile, imgbandnum, segfile, statsSelection, missingStatsValue)``
keeps the reference signature (tilingstats.py:85-216).  The reference streams 1024² tiles
through numba dict-of-dict histograms and pages the results into the GDAL RAT; here the whole
band and label raster go to HBM once and the exact per-segment value multisets come out of two
stable radix sorts (``shp_segstats``).  Results are identical for the integer statistics and
bit-identical for mean / stddev on the reference's golden vectors.

Rasters: numpy arrays or ``.npy`` paths (GDAL optional, imported lazily).  Without GDAL the
columns are returned in ``result.columns`` (name -> array indexed by segment id) instead of
being written to the RAT.  Out of scope: the spatial statistics with user njit callbacks
(tilingstats.py:1262-1390) and the RIOS variants.
"""
import ctypes

import numpy

from . import _lib
from . import shepseg
from .tiling import Timers

STATID_MIN = 0
STATID_MAX = 1
STATID_MEAN = 2
STATID_STDDEV = 3
STATID_MEDIAN = 4
STATID_MODE = 5
STATID_PERCENTILE = 6
STATID_PIXCOUNT = 7
statIDdict = {'min': STATID_MIN, 'max': STATID_MAX, 'mean': STATID_MEAN, 'stddev': STATID_STDDEV,
              'median': STATID_MEDIAN, 'mode': STATID_MODE, 'percentile': STATID_PERCENTILE,
              'pixcount': STATID_PIXCOUNT}
STATSSELFAST_DTYPE = numpy.uint32
NOPARAM = numpy.iinfo(STATSSELFAST_DTYPE).max
STAT_DTYPE_INT = 0
STAT_DTYPE_FLOAT = 1
(STATSEL_GLOBALCOLINDEX, STATSEL_STATID, STATSEL_COLTYPE, STATSEL_COLARRAYINDEX,
 STATSEL_PARAM) = range(5)
RAT_PAGE_SIZE = 100000


class PyShepSegStatsError(Exception):
    pass


class TiledStatsResult(object):
    """Result of calcPerSegmentStatsTiled (reference tilingstats.py:219-232) plus the computed
    columns (name -> ndarray indexed by segment id; int64 or float32)."""
    def __init__(self):
        self.timings = None
        self.columns = None


def makeFastStatsSelection(colIndexList, statsSelection):
    """(statsSelection_fast, numIntCols, numFloatCols) exactly as the reference builds them
    (tilingstats.py:798-863)."""
    numStats = len(colIndexList)
    fast = numpy.empty((numStats, 5), dtype=STATSSELFAST_DTYPE)
    intCount = 0
    floatCount = 0
    for i in range(numStats):
        fast[i, STATSEL_GLOBALCOLINDEX] = colIndexList[i]
        statName = statsSelection[i][1]
        if statName not in statIDdict:
            raise PyShepSegStatsError("Unknown statistic '{}'".format(statName))
        fast[i, STATSEL_STATID] = statIDdict[statName]
        statType = STAT_DTYPE_FLOAT if statName in ('mean', 'stddev') else STAT_DTYPE_INT
        fast[i, STATSEL_COLTYPE] = statType
        if statType == STAT_DTYPE_INT:
            fast[i, STATSEL_COLARRAYINDEX] = intCount
            intCount += 1
        else:
            fast[i, STATSEL_COLARRAYINDEX] = floatCount
            floatCount += 1
        fast[i, STATSEL_PARAM] = NOPARAM
        if statName == 'percentile':
            fast[i, STATSEL_PARAM] = statsSelection[i][2]
    return (fast, intCount, floatCount)


def _loadArray(obj, band=None):
    if isinstance(obj, numpy.ndarray):
        arr = obj
    elif isinstance(obj, str) and obj.endswith('.npy'):
        arr = numpy.load(obj, mmap_mode='r')
    else:
        return None
    if band is not None and arr.ndim == 3:
        arr = arr[band - 1]
    return arr


def calcPerSegmentStats(seg, band, statsSelection, imgNullVal=None, missingStatsValue=-9999,
                        maxSegId=None):
    """The compute step on arrays: returns (intcols int64 (nInt, maxSegId+1), floatcols float32
    (nFloat, maxSegId+1), statsSelection_fast)."""
    seg = numpy.ascontiguousarray(seg, dtype=shepseg.SegIdType)
    band = numpy.ascontiguousarray(band)
    if band.dtype.kind == 'f':
        raise PyShepSegStatsError("Float image types not supported")      # tilingstats.py:450-452
    if band.dtype not in _lib.SHP_DTYPES:
        b3, _dt = _lib.as_image(band.reshape((1,) + band.shape))
        band = b3[0]
    if band.shape != seg.shape:
        raise PyShepSegStatsError("Images are different sizes")           # tilingstats.py:453-455
    if maxSegId is None:
        maxSegId = int(seg.max()) if seg.size else 0
    (fast, nInt, nFloat) = makeFastStatsSelection(list(range(len(statsSelection))), statsSelection)
    intcols = numpy.zeros((nInt, maxSegId + 1), dtype=numpy.int64)
    floatcols = numpy.zeros((nFloat, maxSegId + 1), dtype=numpy.float32)
    c = _lib.ctx()
    c.check(c._L.shp_segstats(c.handle, _lib.ptr(seg), _lib.ptr(band), _lib.SHP_DTYPES[band.dtype],
                              seg.size, maxSegId, int(imgNullVal is not None),
                              0 if imgNullVal is None else int(imgNullVal), _lib.ptr(fast),
                              len(statsSelection), int(missingStatsValue), _lib.ptr(intcols),
                              _lib.ptr(floatcols)))
    return intcols, floatcols, fast


def calcPerSegmentStatsTiled(imgfile, imgbandnum, segfile, statsSelection,
        missingStatsValue=-9999, imgNullVal=None):
    """
    Calculate selected per-segment statistics for the given band of imgfile against the
    segment raster segfile (reference tilingstats.py:85-216).  statsSelection is a list of
    (columnName, statName[, parameter]) with statName in 'min', 'max', 'mean', 'stddev',
    'median', 'mode', 'percentile', 'pixcount'.  Returns a TiledStatsResult; with GDAL files the
    columns are also written to the segfile's RAT.

    Both rasters may already live in HBM: ``imgfile`` a ``tiling.DeviceRaster`` and ``segfile``
    the result of ``doTiledShepherdSegmentation(..., outfile=tiling._KEEP_ON_DEVICE)``; nothing
    is copied then but the result columns.
    """
    timings = Timers()
    gdalSeg = None
    from . import tiling as _tiling
    if isinstance(imgfile, _tiling.DeviceRaster) and getattr(segfile, 'outDev', None):
        (dptr, nrows, ncols, _nbytes) = segfile.outDev
        (nb, ir, ic) = imgfile.shape
        if (ir, ic) != (nrows, ncols):
            raise PyShepSegStatsError("Images are different sizes")
        if not (1 <= imgbandnum <= nb):
            raise PyShepSegStatsError("band %d not in image" % imgbandnum)
        if imgNullVal is None:
            imgNullVal = imgfile.nullVal
        maxSegId = int(segfile.maxSegId)
        (fast, nInt, nFloat) = makeFastStatsSelection(list(range(len(statsSelection))), statsSelection)
        intcols = numpy.zeros((max(nInt, 1), maxSegId + 1), dtype=numpy.int64)
        floatcols = numpy.zeros((max(nFloat, 1), maxSegId + 1), dtype=numpy.float32)
        c = _lib.ctx()
        band = imgfile.ptr + (imgbandnum - 1) * nrows * ncols * imgfile.dtype.itemsize
        with timings.interval('accumulation'):
            c.check(c._L.shp_segstats_dev(
                c.handle, ctypes.c_void_p(dptr), ctypes.c_void_p(band), _lib.SHP_DTYPES[imgfile.dtype],
                nrows * ncols, maxSegId, int(imgNullVal is not None),
                0 if imgNullVal is None else int(imgNullVal), _lib.ptr(fast), len(statsSelection),
                int(missingStatsValue), _lib.ptr(intcols), _lib.ptr(floatcols)))
        cols = {}
        for i, sel in enumerate(statsSelection):
            src = floatcols if fast[i, STATSEL_COLTYPE] == STAT_DTYPE_FLOAT else intcols
            cols[sel[0]] = src[fast[i, STATSEL_COLARRAYINDEX]]
        rtn = TiledStatsResult()
        rtn.timings = timings
        rtn.columns = cols
        return rtn
    with timings.interval('reading'):
        seg = _loadArray(segfile)
        img = _loadArray(imgfile, imgbandnum)
        if seg is None or img is None:
            (seg, img, imgNullVal, gdalSeg) = _readGdal(imgfile, imgbandnum, segfile, imgNullVal)
    with timings.interval('accumulation'):
        (intcols, floatcols, fast) = calcPerSegmentStats(seg, img, statsSelection, imgNullVal,
                                                         missingStatsValue)
    cols = {}
    for i, sel in enumerate(statsSelection):
        src = floatcols if fast[i, STATSEL_COLTYPE] == STAT_DTYPE_FLOAT else intcols
        cols[sel[0]] = src[fast[i, STATSEL_COLARRAYINDEX]]
    if gdalSeg is not None:
        with timings.interval('writing'):
            _writeRat(gdalSeg, statsSelection, cols)
    rtn = TiledStatsResult()
    rtn.timings = timings
    rtn.columns = cols
    return rtn


def _readGdal(imgfile, imgbandnum, segfile, imgNullVal):
    try:
        from osgeo import gdal
    except ImportError:
        raise PyShepSegStatsError("GDAL (osgeo) is not importable here: pass numpy arrays or "
                                  ".npy paths")
    gdal.UseExceptions()
    segds = segfile if isinstance(segfile, gdal.Dataset) else gdal.Open(segfile, gdal.GA_Update)
    imgds = gdal.Open(imgfile)
    if (segds.RasterXSize != imgds.RasterXSize) or (segds.RasterYSize != imgds.RasterYSize):
        raise PyShepSegStatsError("Images are different sizes")
    if segds.GetGeoTransform() != imgds.GetGeoTransform():
        raise PyShepSegStatsError("Images have different spatial extents or pixel sizes")
    imgband = imgds.GetRasterBand(imgbandnum)
    if imgNullVal is None:
        imgNullVal = imgband.GetNoDataValue()
    attrTbl = segds.GetRasterBand(1).GetDefaultRAT()
    names = [attrTbl.GetNameOfCol(i) for i in range(attrTbl.GetColumnCount())]
    if 'Histogram' not in names:
        raise PyShepSegStatsError("Histogram column must exist before calculating per-segment stats")
    return (segds.GetRasterBand(1).ReadAsArray(), imgband.ReadAsArray(), imgNullVal, segds)


def _writeRat(segds, statsSelection, cols):
    from osgeo import gdal
    attrTbl = segds.GetRasterBand(1).GetDefaultRAT()
    names = [attrTbl.GetNameOfCol(i) for i in range(attrTbl.GetColumnCount())]
    for sel in statsSelection:
        (colName, statName) = sel[:2]
        if colName not in names:
            colType = gdal.GFT_Real if statName in ('mean', 'stddev') else gdal.GFT_Integer
            attrTbl.CreateColumn(colName, colType, gdal.GFU_Generic)
            names.append(colName)
        attrTbl.WriteArray(cols[colName], names.index(colName))
    segds.FlushCache()


# ------------------------------------------------------------------------------------------
# spatial statistics with the reference's built-in user functions
# ------------------------------------------------------------------------------------------
GFT_Integer, GFT_Real = 0, 1        # gdal.GFT_* values, so callers need not import GDAL


class _BuiltinSpatialFunc(object):
    """Stands for one of the reference's njit user functions; on the GPU they are fixed
    reductions (pyshepseg_amd/csrc/spatial.h), so the object only carries an id."""
    def __init__(self, funcId, name):
        self.funcId, self.__name__ = funcId, name

    def __call__(self, *args):
        raise PyShepSegStatsError("%s is evaluated on the GPU; it cannot be called" % self.__name__)


userFuncMeanCoord = _BuiltinSpatialFunc(0, 'userFuncMeanCoord')             # tilingstats.py:1098
userFuncNumEdgePixels = _BuiltinSpatialFunc(1, 'userFuncNumEdgePixels')     # tilingstats.py:1146
userFuncVariogram = _BuiltinSpatialFunc(2, 'userFuncVariogram')             # tilingstats.py:1037


def calcPerSegmentSpatialStats(seg, band, colTypes, userFunc, userParam, imgNullVal,
                               missingStatsValue=-9999, maxSegId=None):
    """The compute step on arrays: colTypes = list of GFT_Integer / GFT_Real in column order.
    Returns (intcols int64 (nInt, maxSegId+1), floatcols float32 (nFloat, maxSegId+1))."""
    if not isinstance(userFunc, _BuiltinSpatialFunc):
        raise PyShepSegStatsError(
            "only the built-in user functions (userFuncMeanCoord, userFuncNumEdgePixels, "
            "userFuncVariogram) are supported on the GPU")
    seg = numpy.ascontiguousarray(seg, dtype=shepseg.SegIdType)
    band = numpy.ascontiguousarray(band)
    if band.dtype.kind == 'f':
        raise PyShepSegStatsError("Float image types not supported")
    if band.dtype not in _lib.SHP_DTYPES:
        b3, _dt = _lib.as_image(band.reshape((1,) + band.shape))
        band = b3[0]
    if band.shape != seg.shape or seg.ndim != 2:
        raise PyShepSegStatsError("Images are different sizes")
    if maxSegId is None:
        maxSegId = int(seg.max()) if seg.size else 0
    nInt = sum(1 for t in colTypes if t == GFT_Integer)
    nFloat = sum(1 for t in colTypes if t == GFT_Real)
    if nInt + nFloat != len(colTypes):
        raise PyShepSegStatsError("column types must be GFT_Integer or GFT_Real")
    params = numpy.zeros(6, dtype=numpy.float64)
    pv = numpy.atleast_1d(numpy.asarray(0 if userParam is None else userParam, dtype=numpy.float64))
    params[:min(len(pv), 6)] = pv[:6]
    intcols = numpy.zeros((max(nInt, 1), maxSegId + 1), dtype=numpy.int64)
    floatcols = numpy.zeros((max(nFloat, 1), maxSegId + 1), dtype=numpy.float32)
    c = _lib.ctx()
    c.check(c._L.shp_spatialstats(c.handle, _lib.ptr(seg), _lib.ptr(band), _lib.SHP_DTYPES[band.dtype],
                                  seg.shape[0], seg.shape[1], maxSegId, int(imgNullVal),
                                  userFunc.funcId, _lib.ptr(params), int(missingStatsValue), nInt,
                                  nFloat, _lib.ptr(intcols), _lib.ptr(floatcols)))
    return intcols[:nInt], floatcols[:nFloat]


def calcPerSegmentSpatialStatsTiled(imgfile, imgbandnum, segfile, colNamesAndTypes, userFunc,
        userParam=None, missingStatsValue=-9999, imgNullVal=None):
    """
    Spatial per-segment statistics (reference tilingstats.py:1262-1390) for the reference's
    built-in user functions: pass this module's ``userFuncMeanCoord`` (userParam = the six
    geotransform numbers; two Real columns), ``userFuncNumEdgePixels`` (userParam =
    fourConnected; one Integer column) or ``userFuncVariogram`` (userParam = maxDist; maxDist Real
    columns).  ``colNamesAndTypes`` is the reference's list of (name, GFT_Integer | GFT_Real); the
    order of the integer / real columns is the order of the function's intArr / floatArr.
    ``imgNullVal`` stands for the image band's nodata value, which must be set (the reference
    raises the same error, tilingstats.py:1325-1333).  Arbitrary njit callbacks are not supported.
    Returns a TiledStatsResult whose ``columns`` maps column name -> array (one row per id).
    """
    timings = Timers()
    if imgNullVal is None:
        raise PyShepSegStatsError("NoData value must be set on imgfile")
    if len(colNamesAndTypes) == 0:
        raise PyShepSegStatsError("Must specify one or more columns")
    with timings.interval('reading'):
        seg = _loadArray(segfile)
        img = _loadArray(imgfile, imgbandnum)
        if seg is None or img is None:
            raise PyShepSegStatsError("GDAL is not available here: pass numpy arrays or .npy paths")
    with timings.interval('accumulation'):
        (intcols, floatcols) = calcPerSegmentSpatialStats(
            seg, img, [t for (_n, t) in colNamesAndTypes], userFunc, userParam, imgNullVal,
            missingStatsValue)
    cols = {}
    ni = nf = 0
    for (name, t) in colNamesAndTypes:
        if t == GFT_Integer:
            cols[name] = intcols[ni]
            ni += 1
        else:
            cols[name] = floatcols[nf]
            nf += 1
    rtn = TiledStatsResult()
    rtn.timings = timings
    rtn.columns = cols
    return rtn
