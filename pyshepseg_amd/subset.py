"""
Subset a segmentation so that only the segments present in the subset remain, renumbered in
first-seen order -- the compute step of the reference's ``pyshepseg.subset.subsetImage``
(subset.py:39-227) on the GPU (``shp_subset_recode``).

Rasters are numpy arrays / ``.npy`` paths (GDAL is optional and absent on the build machines);
the RAT of the reference becomes a dict of column arrays (``ratColumns``), every column row-
gathered to the new ids exactly as ``copySubsettedSegmentsToNew`` (subset.py:232-266) does.
"""
import ctypes

import numpy

from . import _lib
from . import shepseg
from . import tiling


class PyShepSegSubsetError(Exception):
    "Same name as the reference's exception (subset.py:450)"


class SubsetResult(object):
    """segimg: recoded window (newYsize, newXsize) uint32; origSegIds[new id] = old id (row 0 = 0);
    hist[new id] = pixel count (the output 'Histogram' column); columns: the input RAT columns
    gathered to the new ids (plus origSegIdColName when asked for)."""
    def __init__(self):
        self.segimg = None
        self.origSegIds = None
        self.hist = None
        self.columns = {}


def _load(x):
    if isinstance(x, str):
        if not x.endswith('.npy'):
            raise PyShepSegSubsetError("GDAL is not available here: pass a numpy array or a .npy path")
        return numpy.load(x, mmap_mode='r')
    return x


def subsetImage(inname, outname, tlx, tly, newXsize, newYsize, outformat=None, creationOptions=[],
                origSegIdColName=None, maskImage=None, ratColumns=None, tileSize=None):
    """
    Same arguments as the reference (subset.py:39-40) plus ``ratColumns`` (dict name -> array
    with one row per input segment id, the input RAT) and ``tileSize`` (visiting-order tile,
    default tiling.TILESIZE as in the reference).  ``inname`` is a (nRows, nCols) uint32 array
    or a ``.npy`` path; ``outname`` None (result only) or a ``.npy`` path; ``maskImage`` an
    array / ``.npy`` path of shape (newYsize, newXsize): only non-zero pixels are included.
    Returns a :class:`SubsetResult`.
    """
    seg = _load(inname)
    if seg.ndim != 2:
        raise PyShepSegSubsetError("input must be a single-band label raster")
    (tlx, tly, newXsize, newYsize) = (int(tlx), int(tly), int(newXsize), int(newYsize))
    if (tlx + newXsize) > seg.shape[1] or (tly + newYsize) > seg.shape[0] or tlx < 0 or tly < 0:
        raise PyShepSegSubsetError('Requested subset is not within input image')
    mask = None
    if maskImage is not None:
        mask = numpy.asarray(_load(maskImage))
        if mask.shape != (newYsize, newXsize):
            raise PyShepSegSubsetError('mask should match requested subset size if supplied')
        mask = numpy.ascontiguousarray(mask != 0, dtype=numpy.uint8)
    if tileSize is None:
        tileSize = tiling.TILESIZE
    # only the window is needed on the device
    win = numpy.ascontiguousarray(seg[tly:tly + newYsize, tlx:tlx + newXsize], dtype=shepseg.SegIdType)
    valid = win if mask is None else win[mask != 0]
    valid = valid[valid != shepseg.SEGNULLVAL]
    if valid.size == 0:
        raise PyShepSegSubsetError('No valid data found in subset')
    maxId = int(valid.max())
    cap = min(maxId, win.size) + 1
    out = numpy.empty((newYsize, newXsize), dtype=shepseg.SegIdType)
    orig = numpy.zeros(cap, dtype=numpy.uint32)
    hist = numpy.zeros(cap, dtype=numpy.uint32)
    nnew = ctypes.c_uint32(0)
    c = _lib.ctx()
    c.check(c._L.shp_subset_recode(
        c.handle, _lib.ptr(win), newYsize, newXsize, 0, 0, newXsize, newYsize,
        _lib.ptr(mask) if mask is not None else None, int(tileSize), maxId, _lib.ptr(out),
        _lib.ptr(orig), _lib.ptr(hist), cap, ctypes.byref(nnew)))
    n = nnew.value
    res = SubsetResult()
    res.segimg = out
    res.origSegIds = orig[:n + 1].copy()
    res.hist = hist[:n + 1].copy()
    if ratColumns:
        for name, col in ratColumns.items():
            col = numpy.asarray(col)
            if col.shape[0] <= maxId:
                raise PyShepSegSubsetError("RAT column %r has %d rows, segment id %d needs more"
                                           % (name, col.shape[0], maxId))
            new = col[res.origSegIds]
            new[0] = 0
            res.columns[name] = new
    res.columns['Histogram'] = res.hist.astype(numpy.float64)          # subset.py:196-205
    if origSegIdColName is not None:
        res.columns[origSegIdColName] = res.origSegIds.astype(numpy.int32)     # subset.py:207-226
    if outname is not None:
        if not (isinstance(outname, str) and outname.endswith('.npy')):
            raise PyShepSegSubsetError("GDAL is not available here: outname must be None or a .npy path")
        numpy.save(outname, out)
    return res
