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


# ------------------------------------------------------------------------------------------
# paged RAT (reference tilingstats.py:1935-2045, :723-764)
# ------------------------------------------------------------------------------------------
def getRatPageId(segId):
    """The page a segment id lives in = the id of the page's first row (tilingstats.py:1950-1956)."""
    return (int(segId) // RAT_PAGE_SIZE) * RAT_PAGE_SIZE


class RatPage(object):
    """One page of the paged RAT: RAT_PAGE_SIZE consecutive segment ids (fewer in the last page),
    int columns int64, float columns float32, a completion flag per row; the null segment's row
    is born complete, all zero (reference tilingstats.py:1972-2045)."""
    def __init__(self, numIntCols, numFloatCols, startSegId, numSeg):
        self.startSegId = int(startSegId)
        self.intcols = numpy.empty((numIntCols, numSeg), dtype=numpy.int64)
        self.floatcols = numpy.empty((numFloatCols, numSeg), dtype=numpy.float32)
        self.complete = numpy.zeros(numSeg, dtype=bool)
        if self.startSegId == shepseg.SEGNULLVAL:
            self.complete[0] = True
            self.intcols[:, 0] = 0
            self.floatcols[:, 0] = 0

    def getIndexInPage(self, segId):
        return segId - self.startSegId

    def setRatVal(self, segId, colType, colArrayNdx, val):
        if colType == STAT_DTYPE_INT:
            self.intcols[colArrayNdx, segId - self.startSegId] = val
        elif colType == STAT_DTYPE_FLOAT:
            self.floatcols[colArrayNdx, segId - self.startSegId] = val

    def getRatVal(self, segId, colType, colArrayNdx):
        if colType == STAT_DTYPE_INT:
            return self.intcols[colArrayNdx, segId - self.startSegId]
        return self.floatcols[colArrayNdx, segId - self.startSegId]

    def setSegmentComplete(self, segId):
        self.complete[segId - self.startSegId] = True

    def getSegmentComplete(self, segId):
        return bool(self.complete[segId - self.startSegId])

    def pageComplete(self):
        return bool(self.complete.all())


def createPagedRat():
    """page id -> RatPage, initially empty (reference tilingstats.py:1935-1946)."""
    return {}


class MemoryRat(object):
    """Stands where the reference has the GDAL RasterAttributeTable when the rasters are arrays:
    the same ``WriteArray(column, colNumber, start=row)`` call fills whole-table column arrays
    (``columns[colNumber]``, int64 or float32, numRows rows).  ``pagesWritten`` records the
    (startSegId, numRows) of every page in the order it was written."""
    def __init__(self, numRows, colTypes):
        self.columns = [numpy.zeros(numRows, dtype=numpy.float32 if t == STAT_DTYPE_FLOAT else numpy.int64)
                        for t in colTypes]
        self.pagesWritten = []

    def WriteArray(self, colArr, colNumber, start=0):
        self.columns[colNumber][start:start + len(colArr)] = colArr

    def notePage(self, startSegId, numRows):
        self.pagesWritten.append((int(startSegId), int(numRows)))


def writeCompletePages(pagedRat, attrTbl, statsSelection_fast):
    """Write every completed page to the attribute table, column by column at its start row, and
    drop it from the paged RAT (reference tilingstats.py:723-764)."""
    for pageId in sorted(pagedRat.keys()):
        ratPage = pagedRat[pageId]
        if not ratPage.pageComplete():
            continue
        for statSel in statsSelection_fast:
            colNumber = int(statSel[STATSEL_GLOBALCOLINDEX])
            if statSel[STATSEL_COLTYPE] == STAT_DTYPE_INT:
                colArr = ratPage.intcols[statSel[STATSEL_COLARRAYINDEX]]
            else:
                colArr = ratPage.floatcols[statSel[STATSEL_COLARRAYINDEX]]
            attrTbl.WriteArray(colArr, colNumber, start=ratPage.startSegId)
        if hasattr(attrTbl, 'notePage'):
            attrTbl.notePage(ratPage.startSegId, len(ratPage.complete))
        pagedRat.pop(pageId)


def _pageRows(pagedRat, segIds, intRows, floatRows, segSize, numIntCols, numFloatCols, emptyRow):
    """Store finished rows (columns x ids) in their pages and flag them complete.  A page is created
    on first touch; ids of that page which have no pixels at all (segSize == 0: the stitch can
    leave such ids, tiling.py:1308-1341) are completed there and then with `emptyRow`, so that the
    page can finish -- the reference never sees them and fails with 'Not all pixels found'."""
    if len(segIds) == 0:
        return
    numRows = len(segSize)
    pages = (segIds // RAT_PAGE_SIZE) * RAT_PAGE_SIZE
    order = numpy.argsort(pages, kind='stable')
    (upages, first) = numpy.unique(pages[order], return_index=True)
    bounds = list(first) + [len(order)]
    for (k, pageId) in enumerate(upages):
        sel = order[bounds[k]:bounds[k + 1]]
        pageId = int(pageId)
        page = pagedRat.get(pageId)
        if page is None:
            numSeg = min(RAT_PAGE_SIZE, numRows - pageId)
            page = pagedRat[pageId] = RatPage(numIntCols, numFloatCols, pageId, numSeg)
            empty = numpy.flatnonzero(segSize[pageId:pageId + numSeg] == 0)
            if pageId == shepseg.SEGNULLVAL:
                empty = empty[empty != 0]                     # row 0 is the null segment: zeros
            if len(empty):
                page.intcols[:, empty] = emptyRow[0][:, None]
                page.floatcols[:, empty] = emptyRow[1][:, None]
                page.complete[empty] = True
        idx = segIds[sel] - pageId
        page.intcols[:, idx] = intRows[:, sel]
        page.floatcols[:, idx] = floatRows[:, sel]
        page.complete[idx] = True


class _ChunkSource(object):
    """Row blocks of (label raster, image band) as device pointers: resident rasters are addressed
    in place, host arrays / memmaps go up block by block into two reusable device buffers."""
    def __init__(self, c, seg, band, devSeg=None, devBand=None, bandDtype=None, shape=None):
        self.c = c
        (self.seg, self.band, self.devSeg, self.devBand) = (seg, band, devSeg, devBand)
        (self.nrows, self.ncols) = shape
        self.bandDtype = numpy.dtype(bandDtype)
        self.bufs = []

    def _buf(self, i, nbytes):
        while len(self.bufs) <= i:
            self.bufs.append([None, 0])
        if self.bufs[i][1] < nbytes:
            if self.bufs[i][0] is not None:
                self.c.check(self.c._L.shp_dev_free(self.c.handle, self.bufs[i][0]))
            p = ctypes.c_void_p()
            self.c.check(self.c._L.shp_dev_alloc(self.c.handle, nbytes, ctypes.byref(p)))
            self.bufs[i] = [p, nbytes]
        return self.bufs[i][0]

    def chunk(self, y0, y1):
        n = (y1 - y0) * self.ncols
        if self.devSeg is not None:
            return (ctypes.c_void_p(self.devSeg + 4 * y0 * self.ncols),
                    ctypes.c_void_p(self.devBand + self.bandDtype.itemsize * y0 * self.ncols))
        s = numpy.ascontiguousarray(self.seg[y0:y1], dtype=shepseg.SegIdType)
        b = numpy.ascontiguousarray(self.band[y0:y1], dtype=self.bandDtype)
        ds = self._buf(0, n * 4)
        db = self._buf(1, n * self.bandDtype.itemsize)
        self.c.check(self.c._L.shp_dev_upload(self.c.handle, ds, _lib.ptr(s), s.nbytes))
        self.c.check(self.c._L.shp_dev_upload(self.c.handle, db, _lib.ptr(b), b.nbytes))
        return (ds, db)

    def scratch(self, i, nbytes):
        return self._buf(2 + i, nbytes)

    def close(self):
        for (p, _n) in self.bufs:
            if p is not None:
                self.c.check(self.c._L.shp_dev_free(self.c.handle, p))
        self.bufs = []


STATS_CHUNK_PIXELS = 1 << 28        # pixels per streamed block (the kernels index a block with 32 bits)


def _streamStats(src, segSize, statsSelection_fast, numIntCols, numFloatCols, imgNullVal,
                 missingStatsValue, attrTbl, timings, chunkPixels):
    """The tile loop of calcPerSegmentStatsTiled (tilingstats.py:183-206) over row blocks.  Per block:
    the labels are renumbered 1..m in first-seen order on the device (shp_subset_recode_dev, which
    also counts their pixels), the block's statistics are computed for those m ids, the ids whose
    block count equals segSize are complete (checkSegComplete, :518-553) and go to their RAT page;
    the (id, value) pairs of the others are set aside and reduced once at the end."""
    c = src.c
    L = c._L
    (nrows, ncols) = (src.nrows, src.ncols)
    S = len(segSize) - 1
    dt = _lib.SHP_DTYPES[src.bandDtype]
    nstats = len(statsSelection_fast)
    fast = numpy.ascontiguousarray(statsSelection_fast, dtype=numpy.uint32)
    # statistics of a segment without pixels (see _pageRows)
    emptyInt = numpy.full(numIntCols, int(missingStatsValue), dtype=numpy.int64)
    emptyFloat = numpy.full(numFloatCols, float(missingStatsValue), dtype=numpy.float32)
    for sel in fast:
        if sel[STATSEL_STATID] == STATID_PIXCOUNT:
            emptyInt[sel[STATSEL_COLARRAYINDEX]] = 0
    pagedRat = createPagedRat()
    written = set()

    def flush():
        written.update(pid for (pid, pg) in pagedRat.items() if pg.pageComplete())
        writeCompletePages(pagedRat, attrTbl, fast)

    rowsPerChunk = max(1, min(nrows, int(chunkPixels) // max(ncols, 1)))
    carryIds = []
    carryVals = []
    nullFlag = int(imgNullVal is not None)
    nullV = 0 if imgNullVal is None else int(imgNullVal)
    for y0 in range(0, nrows, rowsPerChunk):
        y1 = min(nrows, y0 + rowsPerChunk)
        n = (y1 - y0) * ncols
        with timings.interval('reading'):
            (dseg, dband) = src.chunk(y0, y1)
        with timings.interval('accumulation'):
            cap = min(S, n) + 1
            drec = src.scratch(0, n * 4)
            orig = numpy.zeros(cap, dtype=numpy.uint32)
            lhist = numpy.zeros(cap, dtype=numpy.uint32)
            nnew = ctypes.c_uint32(0)
            c.check(L.shp_subset_recode_dev(c.handle, dseg, y1 - y0, ncols, 0, 0, ncols, y1 - y0, None,
                                            1 << 30, S, drec, _lib.ptr(orig), _lib.ptr(lhist), cap,
                                            ctypes.byref(nnew)))
            m = nnew.value
            if m == 0:
                continue
            ic = numpy.zeros((max(numIntCols, 1), m + 1), dtype=numpy.int64)
            fc = numpy.zeros((max(numFloatCols, 1), m + 1), dtype=numpy.float32)
            # (the block's shape goes along: with small segments the library works patch by patch)
            c.check(L.shp_segstats2d_dev(c.handle, drec, dband, dt, y1 - y0, ncols, m, nullFlag, nullV,
                                         _lib.ptr(fast), nstats, int(missingStatsValue), _lib.ptr(ic),
                                         _lib.ptr(fc)))
        with timings.interval('statscompletion'):
            ids = orig[1:m + 1].astype(numpy.int64)
            done = lhist[1:m + 1] == segSize[ids]
            sel = numpy.flatnonzero(done)
            _pageRows(pagedRat, ids[sel], ic[:numIntCols, 1:][:, sel], fc[:numFloatCols, 1:][:, sel],
                      segSize, numIntCols, numFloatCols, (emptyInt, emptyFloat))
            rest = numpy.flatnonzero(~done)
            if len(rest):
                flags = numpy.zeros(m + 1, dtype=numpy.uint8)
                flags[rest + 1] = 1
                npairs = int(lhist[1:m + 1][rest].sum())
                so = numpy.empty(npairs, dtype=numpy.uint32)
                vo = numpy.empty(npairs, dtype=numpy.int64)
                cnt = ctypes.c_int64(0)
                c.check(L.shp_gather_flagged_dev(c.handle, drec, dband, dt, n, m, _lib.ptr(flags), npairs,
                                                 _lib.ptr(so), _lib.ptr(vo), ctypes.byref(cnt)))
                if cnt.value != npairs:
                    raise PyShepSegStatsError("internal: %d pixels of unfinished segments, expected %d"
                                              % (cnt.value, npairs))
                carryIds.append(orig[so])
                carryVals.append(vo.astype(src.bandDtype))
        with timings.interval('writing'):
            flush()
    if carryIds:
        # the segments that straddle block boundaries: all their pixels are here now
        with timings.interval('statscompletion'):
            allIds = numpy.concatenate(carryIds)
            allVals = numpy.concatenate(carryVals)
            (uids, compact) = numpy.unique(allIds, return_inverse=True)
            counts = numpy.bincount(compact, minlength=len(uids))
            if not numpy.array_equal(counts, segSize[uids]):
                raise PyShepSegStatsError('Not all pixels found during processing')     # tilingstats.py:211
            m = len(uids)
            ic = numpy.zeros((max(numIntCols, 1), m + 1), dtype=numpy.int64)
            fc = numpy.zeros((max(numFloatCols, 1), m + 1), dtype=numpy.float32)
            seg1 = numpy.ascontiguousarray(compact + 1, dtype=numpy.uint32)
            c.check(L.shp_segstats(c.handle, _lib.ptr(seg1), _lib.ptr(allVals), dt, len(seg1), m, nullFlag,
                                   nullV, _lib.ptr(fast), nstats, int(missingStatsValue), _lib.ptr(ic),
                                   _lib.ptr(fc)))
            _pageRows(pagedRat, uids.astype(numpy.int64), ic[:numIntCols, 1:], fc[:numFloatCols, 1:], segSize,
                      numIntCols, numFloatCols, (emptyInt, emptyFloat))
    # pages no block touched hold only ids without pixels
    with timings.interval('writing'):
        for pageId in range(0, S + 1, RAT_PAGE_SIZE):
            if pageId in written or pageId in pagedRat:
                continue
            if (segSize[max(pageId, 1):pageId + RAT_PAGE_SIZE] != 0).any():
                raise PyShepSegStatsError('Not all pixels found during processing')      # tilingstats.py:211
            numSeg = min(RAT_PAGE_SIZE, S + 1 - pageId)
            page = pagedRat[pageId] = RatPage(numIntCols, numFloatCols, pageId, numSeg)
            first = 1 if pageId == shepseg.SEGNULLVAL else 0
            page.intcols[:, first:] = emptyInt[:, None]
            page.floatcols[:, first:] = emptyFloat[:, None]
            page.complete[:] = True
        flush()
    if len(pagedRat) > 0:
        raise PyShepSegStatsError('Not all pixels found during processing')              # tilingstats.py:211


def calcPerSegmentStatsTiled(imgfile, imgbandnum, segfile, statsSelection,
        missingStatsValue=-9999, imgNullVal=None, segSize=None, chunkPixels=None):
    """
    Calculate selected per-segment statistics for the given band of imgfile against the
    segment raster segfile (reference tilingstats.py:85-216).  statsSelection is a list of
    (columnName, statName[, parameter]) with statName in 'min', 'max', 'mean', 'stddev',
    'median', 'mode', 'percentile', 'pixcount'.  Returns a TiledStatsResult whose ``columns``
    maps column name -> whole-table array; with GDAL files the columns are written to the
    segfile's RAT page by page (RAT_PAGE_SIZE rows) as the reference does.

    The rasters are streamed through the GPU in row blocks of ``chunkPixels`` pixels (default
    STATS_CHUNK_PIXELS), so they may be larger than the kernels' 32-bit pixel index and than HBM;
    a finished page leaves for the attribute table as soon as all its segments are complete.
    ``segSize`` stands for the reference's RAT 'Histogram' column (pixels per segment id): taken
    from the RAT for GDAL files, from ``segfile.hist`` for a tiled-segmentation result, counted
    here otherwise.  Both rasters may already live in HBM: ``imgfile`` a ``tiling.DeviceRaster``
    and ``segfile`` the result of ``doTiledShepherdSegmentation(..., outfile=tiling._KEEP_ON_DEVICE)``.
    """
    timings = Timers()
    from . import tiling as _tiling
    c = _lib.ctx()
    gdalSeg = None
    if chunkPixels is None:
        chunkPixels = STATS_CHUNK_PIXELS
    with timings.interval('reading'):
        if isinstance(imgfile, _tiling.DeviceRaster) and getattr(segfile, 'outDev', None):
            (dptr, nrows, ncols, _nbytes) = segfile.outDev
            (nb, ir, ic_) = imgfile.shape
            if (ir, ic_) != (nrows, ncols):
                raise PyShepSegStatsError("Images are different sizes")
            if not (1 <= imgbandnum <= nb):
                raise PyShepSegStatsError("band %d not in image" % imgbandnum)
            if imgNullVal is None:
                imgNullVal = imgfile.nullVal
            if segSize is None:
                segSize = getattr(segfile, 'hist', None)
            band = imgfile.ptr + (imgbandnum - 1) * nrows * ncols * imgfile.dtype.itemsize
            src = _ChunkSource(c, None, None, devSeg=dptr, devBand=band, bandDtype=imgfile.dtype,
                               shape=(nrows, ncols))
            maxSegId = int(segfile.maxSegId)
        else:
            seg = _loadArray(segfile)
            img = _loadArray(imgfile, imgbandnum)
            if seg is None or img is None:
                (seg, img, imgNullVal, gdalSeg, segSize) = _readGdal(imgfile, imgbandnum, segfile, imgNullVal)
            if img.dtype.kind == 'f':
                raise PyShepSegStatsError("Float image types not supported")        # tilingstats.py:450-452
            if img.shape != seg.shape:
                raise PyShepSegStatsError("Images are different sizes")             # tilingstats.py:453-455
            bdt = img.dtype
            if bdt not in _lib.SHP_DTYPES:
                bdt = _lib.as_image(numpy.zeros((1, 1, 1), dtype=img.dtype))[0].dtype
            src = _ChunkSource(c, seg, img, bandDtype=bdt, shape=seg.shape)
            maxSegId = None
    try:
        if segSize is None:
            with timings.interval('reading'):
                segSize = _countSegments(src, chunkPixels)
        segSize = numpy.ascontiguousarray(segSize).astype(numpy.int64)
        if maxSegId is not None and len(segSize) < maxSegId + 1:
            raise PyShepSegStatsError("segSize has %d rows, segment id %d needs more" % (len(segSize), maxSegId))
        (fast, nInt, nFloat) = makeFastStatsSelection(list(range(len(statsSelection))), statsSelection)
        if gdalSeg is not None:
            attrTbl = _GdalRat(gdalSeg, statsSelection, fast)
        else:
            attrTbl = MemoryRat(len(segSize), [int(f[STATSEL_COLTYPE]) for f in fast])
        _streamStats(src, segSize, fast, nInt, nFloat, imgNullVal, missingStatsValue, attrTbl, timings,
                     chunkPixels)
    finally:
        src.close()
    rtn = TiledStatsResult()
    rtn.timings = timings
    if isinstance(attrTbl, MemoryRat):
        rtn.columns = {sel[0]: attrTbl.columns[i] for (i, sel) in enumerate(statsSelection)}
        rtn.pagesWritten = attrTbl.pagesWritten
    else:
        attrTbl.flush()
        rtn.columns = None
    return rtn


def _countSegments(src, chunkPixels):
    """Pixels per segment id of the label raster (what the reference reads from the RAT's
    'Histogram' column, tilingstats.py:165-166), block by block on the device."""
    c = src.c
    (nrows, ncols) = (src.nrows, src.ncols)
    rowsPerChunk = max(1, min(max(nrows, 1), int(chunkPixels) // max(ncols, 1)))
    maxId = 0
    parts = []
    for y0 in range(0, nrows, rowsPerChunk):
        y1 = min(nrows, y0 + rowsPerChunk)
        if src.devSeg is None:
            blk = numpy.asarray(src.seg[y0:y1])
            parts.append(numpy.bincount(blk.reshape(-1)))
        else:
            raise PyShepSegStatsError("segSize (the label histogram) is needed for a device-resident label raster")
    n = max([len(p) for p in parts] or [1])
    out = numpy.zeros(n, dtype=numpy.int64)
    for p in parts:
        out[:len(p)] += p
    return out


class _GdalRat(object):
    """The segfile's GDAL attribute table behind the WriteArray interface writeCompletePages uses;
    creates the requested columns like the reference's createStatColumns (tilingstats.py:682-720):
    Real for mean / stddev, Integer otherwise."""
    def __init__(self, segds, statsSelection, fast):
        from osgeo import gdal
        self.segds = segds
        self.tbl = segds.GetRasterBand(1).GetDefaultRAT()
        names = [self.tbl.GetNameOfCol(i) for i in range(self.tbl.GetColumnCount())]
        self.colNdx = []
        for sel in statsSelection:
            (colName, statName) = sel[:2]
            if colName not in names:
                colType = gdal.GFT_Real if statName in ('mean', 'stddev') else gdal.GFT_Integer
                self.tbl.CreateColumn(colName, colType, gdal.GFU_Generic)
                names.append(colName)
            else:
                print('Column {} already exists'.format(colName))
            self.colNdx.append(names.index(colName))

    def WriteArray(self, colArr, colNumber, start=0):
        self.tbl.WriteArray(colArr, self.colNdx[colNumber], start=start)

    def flush(self):
        self.segds.FlushCache()


def _readGdal(imgfile, imgbandnum, segfile, imgNullVal):
    """The reference's doImageAlignmentChecks + Histogram column read (tilingstats.py:151-166,
    :409-461); returns (seg, band, nodata, segment dataset, segSize)."""
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
    segSize = attrTbl.ReadAsArray(names.index('Histogram')).astype(numpy.uint32)
    return (segds.GetRasterBand(1).ReadAsArray(), imgband.ReadAsArray(), imgNullVal, segds, segSize)


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
