"""In-memory gdal: Open / GetDriverByName / IdentifyDriver, Dataset, Band, RasterAttributeTable."""
import threading

import numpy

GA_ReadOnly, GA_Update = 0, 1
(GDT_Unknown, GDT_Byte, GDT_UInt16, GDT_Int16, GDT_UInt32, GDT_Int32, GDT_Float32, GDT_Float64) = range(8)
GFT_Integer, GFT_Real, GFT_String = 0, 1, 2
GFU_Generic, GFU_PixelCount, GFU_Name = 0, 1, 2
_NP = {GDT_Byte: numpy.uint8, GDT_UInt16: numpy.uint16, GDT_Int16: numpy.int16, GDT_UInt32: numpy.uint32,
       GDT_Int32: numpy.int32, GDT_Float32: numpy.float32, GDT_Float64: numpy.float64}

REGISTRY = {}          # path -> Dataset
CALLS = []             # (what, ...) in call order
DRIVERS = ('KEA', 'GTiff', 'HFA')
CONCURRENT_BAND_CALLS = [0]      # times two threads were inside one band's ReadAsArray / WriteArray at once


def reset():
    REGISTRY.clear()
    del CALLS[:]
    CONCURRENT_BAND_CALLS[0] = 0


def UseExceptions():
    CALLS.append(('UseExceptions',))


class RasterAttributeTable(object):
    def __init__(self):
        self.names, self.types, self.usages, self.cols, self.nrows = [], [], [], [], 0

    def GetColumnCount(self):
        return len(self.names)

    def GetNameOfCol(self, i):
        return self.names[i]

    def GetTypeOfCol(self, i):
        return self.types[i]

    def GetUsageOfCol(self, i):
        return self.usages[i]

    def GetRowCount(self):
        return self.nrows

    def SetRowCount(self, n):
        CALLS.append(('RAT.SetRowCount', int(n)))
        self.nrows = int(n)
        for i in range(len(self.cols)):
            c = numpy.zeros(self.nrows, dtype=self.cols[i].dtype)
            m = min(len(self.cols[i]), self.nrows)
            c[:m] = self.cols[i][:m]
            self.cols[i] = c

    def GetColOfUsage(self, usage):
        return self.usages.index(usage) if usage in self.usages else -1

    def CreateColumn(self, name, ftype, usage):
        CALLS.append(('RAT.CreateColumn', name, ftype, usage))
        self.names.append(name)
        self.types.append(ftype)
        self.usages.append(usage)
        self.cols.append(numpy.zeros(self.nrows, dtype=numpy.int64 if ftype == GFT_Integer else numpy.float64))
        return 0

    def WriteArray(self, arr, col, start=0):
        arr = numpy.asarray(arr)
        CALLS.append(('RAT.WriteArray', int(col), int(start), int(len(arr))))
        if start + len(arr) > self.nrows:
            self.SetRowCount(start + len(arr))           # (GDAL grows the table)
        self.cols[col][start:start + len(arr)] = arr
        return 0

    def ReadAsArray(self, col, start=0, length=None):
        c = self.cols[col]
        return c[start:(None if length is None else start + length)].copy()


class Band(object):
    def __init__(self, ds, arr, dtype_code):
        (self.ds, self.arr, self.DataType) = (ds, arr, dtype_code)
        (self.YSize, self.XSize) = arr.shape
        self.nodata = None
        self.meta = {}
        self.rat = RasterAttributeTable()
        self.overviews = []
        self.busy = threading.Lock()

    def _enter(self):
        if not self.busy.acquire(False):
            CONCURRENT_BAND_CALLS[0] += 1
            self.busy.acquire()

    def ReadAsArray(self, xoff=0, yoff=0, win_xsize=None, win_ysize=None):
        self._enter()
        try:
            xs = self.XSize - xoff if win_xsize is None else win_xsize
            ys = self.YSize - yoff if win_ysize is None else win_ysize
            if xoff < 0 or yoff < 0 or xoff + xs > self.XSize or yoff + ys > self.YSize:
                raise RuntimeError('Access window out of range in RasterIO()')
            return self.arr[yoff:yoff + ys, xoff:xoff + xs].copy()
        finally:
            self.busy.release()

    def WriteArray(self, array, xoff=0, yoff=0):
        self._enter()
        try:
            a = numpy.asarray(array)
            CALLS.append(('Band.WriteArray', self.ds.path, a.shape, int(xoff), int(yoff), threading.get_ident()))
            if xoff < 0 or yoff < 0 or xoff + a.shape[1] > self.XSize or yoff + a.shape[0] > self.YSize:
                raise RuntimeError('Access window out of range in RasterIO()')
            self.arr[yoff:yoff + a.shape[0], xoff:xoff + a.shape[1]] = a
            return 0
        finally:
            self.busy.release()

    def GetNoDataValue(self):
        return self.nodata

    def SetNoDataValue(self, v):
        CALLS.append(('Band.SetNoDataValue', v))
        self.nodata = float(v)
        base = getattr(self.ds, 'base', None)
        if base is not None:
            base.bands[self.ds.bands.index(self)].nodata = float(v)

    def SetMetadataItem(self, k, v):
        CALLS.append(('Band.SetMetadataItem', k, v))
        self.meta[k] = v

    def GetMetadataItem(self, k):
        return self.meta.get(k)

    def GetDefaultRAT(self):
        return self.rat

    def GetOverviewCount(self):
        return len(self.overviews)

    def GetOverview(self, j):
        return self.overviews[j]

    def FlushCache(self):
        pass


class Dataset(object):
    def __init__(self, path, xs, ys, nbands, etype):
        (self.path, self.RasterXSize, self.RasterYSize, self.RasterCount) = (path, xs, ys, nbands)
        self.bands = [Band(self, numpy.zeros((ys, xs), dtype=_NP[etype]), etype) for _ in range(nbands)]
        self.proj = ''
        self.gt = (0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
        self.flushed = 0
        self.etype = etype

    def GetRasterBand(self, i):
        return self.bands[i - 1]

    def SetProjection(self, p):
        CALLS.append(('Dataset.SetProjection', p))
        self.proj = p

    def GetProjection(self):
        return self.proj

    def SetGeoTransform(self, g):
        CALLS.append(('Dataset.SetGeoTransform', tuple(g)))
        self.gt = tuple(g)

    def GetGeoTransform(self):
        return self.gt

    def BuildOverviews(self, resampling, levels):
        CALLS.append(('Dataset.BuildOverviews', resampling, tuple(levels)))
        for b in self.bands:
            b.overviews = []
            for lvl in levels:
                (oy, ox) = ((self.RasterYSize + lvl - 1) // lvl, (self.RasterXSize + lvl - 1) // lvl)
                b.overviews.append(Band(self, numpy.zeros((oy, ox), dtype=b.arr.dtype), b.DataType))
        return 0

    def FlushCache(self):
        CALLS.append(('Dataset.FlushCache', self.path))
        self.flushed += 1
        base = getattr(self, 'base', None)
        if base is not None:
            base.flushed += 1


class Driver(object):
    def __init__(self, name):
        self.ShortName = name

    def Create(self, path, xs, ys, nbands=1, etype=GDT_Byte, options=None):
        CALLS.append(('Driver.Create', self.ShortName, path, int(xs), int(ys), int(nbands), etype, list(options or [])))
        ds = Dataset(path, int(xs), int(ys), int(nbands), etype)
        ds.driver = self
        REGISTRY[path] = ds
        return ds

    def Delete(self, path):
        CALLS.append(('Driver.Delete', path))
        REGISTRY.pop(path, None)
        return 0


def GetDriverByName(name):
    return Driver(name) if name in DRIVERS else None


def IdentifyDriver(path):
    ds = REGISTRY.get(path)
    return getattr(ds, 'driver', Driver('KEA')) if ds is not None else None


def Open(path, access=GA_ReadOnly):
    """a NEW handle on the stored dataset (pixels, attribute table and metadata are shared; the handle's
    bands have their own busy flags: concurrent calls are only an error on ONE handle)"""
    CALLS.append(('Open', path, access))
    if path not in REGISTRY:
        raise RuntimeError('%s: No such file or directory' % path)
    base = REGISTRY[path]
    h = Dataset.__new__(Dataset)
    h.__dict__.update(base.__dict__)
    h.bands = []
    for b in base.bands:
        nb = Band.__new__(Band)
        nb.__dict__.update(b.__dict__)
        nb.ds = h
        nb.busy = threading.Lock()
        h.bands.append(nb)
    h.base = base
    return h
