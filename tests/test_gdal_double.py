"""The GDAL-facing branches of pyshepseg_amd (tiling._GdalSource, _createGdalOutput, _finishGdalOutput;
tilingstats._readGdal, _GdalRat), executed against the in-memory ``osgeo`` stand-in of tests/fake_osgeo
and compared with what the reference issues (tiling.py:961-975, :1343-1404; tilingstats.py:151-166,
:409-461, :682-764).  No GPU: these run wherever the tests run.  The end-to-end runs through the same
branches are in test_gpu_gdal_double.py."""
import os
import sys

import numpy as np
import pytest

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fake_osgeo')


@pytest.fixture
def gdal(monkeypatch):
    """the stand-in as ``osgeo`` for one test; modules dropped afterwards"""
    monkeypatch.syspath_prepend(FAKE)
    for m in [m for m in sys.modules if m == 'osgeo' or m.startswith('osgeo.')]:
        monkeypatch.delitem(sys.modules, m)
    from osgeo import gdal as g
    g.reset()
    yield g
    for m in [m for m in sys.modules if m == 'osgeo' or m.startswith('osgeo.')]:
        sys.modules.pop(m, None)


def make_image(gdal, path, img, nodata, gt=(500000.0, 30.0, 0.0, 6500000.0, 0.0, -30.0), proj='PROJCS["fake"]'):
    from osgeo import gdal_array
    (nb, ys, xs) = img.shape
    ds = gdal.GetDriverByName('KEA').Create(path, xs, ys, nb, gdal_array.NumericTypeCodeToGDALTypeCode(img.dtype))
    ds.SetProjection(proj)
    ds.SetGeoTransform(gt)
    for b in range(nb):
        band = ds.GetRasterBand(b + 1)
        band.WriteArray(img[b])
        if nodata is not None:
            band.SetNoDataValue(nodata[b] if isinstance(nodata, (list, tuple)) else nodata)
    del gdal.CALLS[:]
    return ds


def test_gdal_source(gdal):
    from pyshepseg_amd import tiling
    rng = np.random.RandomState(1)
    img = rng.randint(0, 5000, size=(3, 40, 50)).astype(np.uint16)
    make_image(gdal, 'in.kea', img, 65535)
    src = tiling._open_source('in.kea')
    assert isinstance(src, tiling._GdalSource)
    assert src.shape == (3, 40, 50) and (src.RasterXSize, src.RasterYSize) == (50, 40) and src.dtype == np.uint16
    assert src.bandNull([1, 2, 3]) == 65535
    assert np.array_equal(src.read([0, 2], 7, 5, 20, 11), img[[0, 2], 5:16, 7:27])
    out = np.zeros((2, 8, 50), dtype=np.uint16)
    src.readRowsInto([1, 2], 30, 38, out)
    assert np.array_equal(out, img[1:3, 30:38])
    make_image(gdal, 'mixed.kea', img, [65535, 0, 65535])
    with pytest.raises(tiling.PyShepSegTilingError, match='Different null values'):      # tiling.py:233-236
        tiling._open_source('mixed.kea').bandNull([1, 2, 3])
    with pytest.raises(RuntimeError):
        tiling._open_source('missing.kea')


def test_gdal_output_calls(gdal):
    """_createGdalOutput / _finishGdalOutput leave the dataset as stitchTiles does (tiling.py:961-975:
    Create uint32, projection and geotransform of the input, LAYER_TYPE thematic, nodata 0; :1343-1358:
    row count, a Real 'Histogram' column of usage PixelCount; :1360-1404: NEAREST overviews)"""
    from pyshepseg_amd import tiling, shepseg
    img = np.zeros((1, 30, 20), dtype=np.uint16)
    make_image(gdal, 'in.kea', img, 0, gt=(1.0, 2.0, 0.0, 3.0, 0.0, -2.0), proj='PROJ-X')
    (ds, band) = tiling._createGdalOutput('out.kea', 30, 20, 'in.kea', 'KEA', ['OPT=1'])
    assert ('Driver.Create', 'KEA', 'out.kea', 20, 30, 1, gdal.GDT_UInt32, ['OPT=1']) in gdal.CALLS
    assert ds.GetProjection() == 'PROJ-X' and ds.GetGeoTransform() == (1.0, 2.0, 0.0, 3.0, 0.0, -2.0)
    assert band.GetMetadataItem('LAYER_TYPE') == 'thematic' and band.GetNoDataValue() == shepseg.SEGNULLVAL
    hist = np.array([0, 5, 7, 0, 3], dtype=np.uint32)
    ov = {4: np.arange(8 * 5, dtype=np.uint32).reshape(8, 5), 8: np.ones((4, 3), dtype=np.uint32)}
    tiling._finishGdalOutput((ds, band), hist, True, ov, [('STATISTICS_MINIMUM', '1'), ('STATISTICS_MEAN', '2.5')])
    assert ('Dataset.BuildOverviews', 'NEAREST', (4, 8)) in gdal.CALLS
    assert np.array_equal(band.GetOverview(0).ReadAsArray(), ov[4]) and np.array_equal(band.GetOverview(1).ReadAsArray(), ov[8])
    rat = band.GetDefaultRAT()
    assert rat.GetRowCount() == 5 and rat.GetColumnCount() == 1 and rat.GetNameOfCol(0) == 'Histogram'
    assert rat.GetTypeOfCol(0) == gdal.GFT_Real and rat.GetUsageOfCol(0) == gdal.GFU_PixelCount
    assert np.array_equal(rat.ReadAsArray(0), hist.astype(np.float64))
    assert band.GetMetadataItem('STATISTICS_MEAN') == '2.5' and ds.flushed == 1
    # an existing PixelCount column is reused (writeHistogramToFile looks it up by usage)
    tiling._finishGdalOutput((ds, band), hist * 2, True, {}, [])
    assert rat.GetColumnCount() == 1 and np.array_equal(rat.ReadAsArray(0), (hist * 2).astype(np.float64))
    with pytest.raises(tiling.PyShepSegTilingError, match="does not support driver"):     # tiling.py:497-499
        tiling._createGdalOutput('o2.kea', 3, 3, 'in.kea', 'NOPE', [])


def test_gdal_rat_and_alignment_checks(gdal):
    """createStatColumns' column types (tilingstats.py:682-720) and doImageAlignmentChecks' errors (:409-461)"""
    from pyshepseg_amd import tilingstats as ts
    seg = np.array([[1, 1, 2], [3, 3, 0]], dtype=np.uint32)
    img = np.array([[[5, 7, 9], [1, 3, 0]]], dtype=np.uint16)
    make_image(gdal, 'img.kea', img, 0)
    make_image(gdal, 'seg.kea', seg[None], 0)
    with pytest.raises(ts.PyShepSegStatsError, match='Histogram column must exist'):
        ts._readGdal('img.kea', 1, 'seg.kea', None)
    rat = gdal.REGISTRY['seg.kea'].GetRasterBand(1).GetDefaultRAT()
    rat.SetRowCount(4)
    rat.CreateColumn('Histogram', gdal.GFT_Real, gdal.GFU_PixelCount)
    rat.WriteArray(np.array([1, 2, 1, 2], dtype=np.float64), 0)
    (s, b, nullv, segds, segSize) = ts._readGdal('img.kea', 1, 'seg.kea', None)
    assert np.array_equal(s, seg) and np.array_equal(b, img[0]) and nullv == 0 and segds.path == 'seg.kea'
    assert segSize.dtype == np.uint32 and segSize.tolist() == [1, 2, 1, 2]
    sel = [('mn', 'min'), ('avg', 'mean'), ('sd', 'stddev'), ('p', 'percentile', 50)]
    (fast, nInt, nFloat) = ts.makeFastStatsSelection(list(range(len(sel))), sel)
    tbl = ts._GdalRat(segds, sel, fast)
    types = {rat.GetNameOfCol(i): rat.GetTypeOfCol(i) for i in range(rat.GetColumnCount())}
    assert types == {'Histogram': gdal.GFT_Real, 'mn': gdal.GFT_Integer, 'avg': gdal.GFT_Real, 'sd': gdal.GFT_Real,
                     'p': gdal.GFT_Integer}
    tbl.WriteArray(np.array([4.5, 6.5], dtype=np.float32), 1, start=2)          # column 1 of the selection = 'avg'
    assert np.array_equal(rat.ReadAsArray(2), [0, 0, 4.5, 6.5])
    ts._GdalRat(segds, sel[:1], fast[:1])                                         # 'Column mn already exists'
    assert rat.GetColumnCount() == 5
    make_image(gdal, 'small.kea', img[:, :1], 0)
    with pytest.raises(ts.PyShepSegStatsError, match='different sizes'):
        ts._readGdal('small.kea', 1, 'seg.kea', None)
    make_image(gdal, 'shifted.kea', img, 0, gt=(1.0, 30.0, 0.0, 2.0, 0.0, -30.0))
    with pytest.raises(ts.PyShepSegStatsError, match='different spatial extents'):
        ts._readGdal('shifted.kea', 1, 'seg.kea', None)
