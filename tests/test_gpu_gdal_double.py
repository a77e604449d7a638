"""End to end through the GDAL branches with the in-memory ``osgeo`` stand-in (tests/fake_osgeo): the tiled
segmentation reads a "KEA file", streams its rows to the device, writes the stitched labels, overviews,
histogram and statistics into another, and the per-segment statistics land in its attribute table page by
page -- every result compared with the in-memory (ndarray) run of the same job."""
import numpy as np
import pytest

from test_gdal_double import gdal, make_image  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_tiled_segmentation_and_stats_through_gdal(gdal, oracle):  # noqa: F811
    from pyshepseg_amd import tiling, tilingstats as ts
    img = oracle.synthimg(31, 3, 700, 900).copy()
    img[:, :9, :] = 65535
    img[:, 300:330, 500:640] = 65535
    make_image(gdal, 'in.kea', img, 65535)
    kw = dict(tileSize=256, overlapSize=64, minSegmentSize=30, numClusters=20, fixedKMeansInit=True)
    want = tiling.doTiledShepherdSegmentation(img, None, imgNullVal=65535, **kw)
    r = tiling.doTiledShepherdSegmentation('in.kea', 'out.kea', **kw)              # null value from the file
    out = gdal.REGISTRY['out.kea']
    band = out.GetRasterBand(1)
    assert r.maxSegId == want.maxSegId and np.array_equal(band.ReadAsArray(), want.segimg)
    assert np.array_equal(r.kmeans.cluster_centers_, want.kmeans.cluster_centers_)
    assert out.GetProjection() == 'PROJCS["fake"]' and out.GetGeoTransform() == gdal.REGISTRY['in.kea'].GetGeoTransform()
    assert band.GetMetadataItem('LAYER_TYPE') == 'thematic' and band.GetNoDataValue() == 0
    # one dataset handle is not safe for concurrent calls: the row blocks arrived one at a time
    assert gdal.CONCURRENT_BAND_CALLS[0] == 0
    writes = [c for c in gdal.CALLS if c[0] == 'Band.WriteArray' and c[1] == 'out.kea' and c[2][1] == 900]
    assert sum(c[2][0] for c in writes) == 700                                      # every row exactly once
    rat = band.GetDefaultRAT()
    assert rat.GetNameOfCol(0) == 'Histogram' and rat.GetUsageOfCol(0) == gdal.GFU_PixelCount
    assert np.array_equal(rat.ReadAsArray(0), np.asarray(want.hist).astype(np.float64))
    assert band.GetMetadataItem('STATISTICS_MEAN') is not None
    for (j, lvl) in enumerate(sorted(want.overviews)):
        assert np.array_equal(band.GetOverview(j).ReadAsArray(), want.overviews[lvl])
    # per-segment statistics into the attribute table (calcPerSegmentStatsTiled on file names)
    sel = [('b2_min', 'min'), ('b2_mean', 'mean'), ('b2_sd', 'stddev'), ('b2_med', 'median'), ('b2_n', 'pixcount')]
    ref = ts.calcPerSegmentStatsTiled(img, 2, want.segimg, sel, imgNullVal=65535)
    del gdal.CALLS[:]
    res = ts.calcPerSegmentStatsTiled('in.kea', 2, 'out.kea', sel)
    assert res.columns is None                                                      # they went to the file
    names = [rat.GetNameOfCol(i) for i in range(rat.GetColumnCount())]
    for (name, stat) in [s[:2] for s in sel]:
        i = names.index(name)
        assert rat.GetTypeOfCol(i) == (gdal.GFT_Real if stat in ('mean', 'stddev') else gdal.GFT_Integer)
        assert np.array_equal(rat.ReadAsArray(i)[1:], ref.columns[name][1:].astype(np.float64 if stat in ('mean', 'stddev') else np.int64)), name
    assert any(c[0] == 'RAT.WriteArray' for c in gdal.CALLS) and gdal.REGISTRY['out.kea'].flushed >= 2
