"""In-memory gdal_array: the two type-code translations pyshepseg touches."""
import numpy

from . import gdal


def GDALTypeCodeToNumericTypeCode(code):
    return gdal._NP[code]


def NumericTypeCodeToGDALTypeCode(t):
    t = numpy.dtype(t).type
    for (code, np_t) in gdal._NP.items():
        if np_t == t:
            return code
    return None
