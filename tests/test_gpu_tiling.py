"""GPU parity of the tiled driver + stitch against the reference's golden stitch vectors and
against the oracle on a larger seeded image."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
STITCH = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'stitch_*.npz')))


@pytest.mark.parametrize('name', STITCH)
@pytest.mark.parametrize('workers', [1, 3])
def test_golden_tiled(name, workers, golden):
    from pyshepseg_amd import tiling, shepseg
    g = golden(name)
    null = int(g['null_val']) if int(g['has_null']) else None
    cfg = tiling.SegmentationConcurrencyConfig(
        concurrencyType=tiling.CONC_THREADS if workers > 1 else tiling.CONC_NONE, numWorkers=workers)
    r = tiling.doTiledShepherdSegmentation(
        g['img'], None, tileSize=int(g['tile_size']), overlapSize=int(g['overlap']),
        minSegmentSize=int(g['min_seg']), imgNullVal=null, fourConnected=bool(g['four']),
        kmeansObj=shepseg.KMeansModel(g['centres']), concurrencyCfg=cfg)
    assert (r.numTileCols, r.numTileRows) == (int(g['ntcols']), int(g['ntrows']))
    assert r.maxSpectralDiff == g['msd']
    assert r.maxSegId == int(g['max_seg_id'])
    assert np.array_equal(r.segimg, g['mosaic'])
    assert np.array_equal(r.hist, g['hist'])


def _oracle_tiled(oracle, img, centres, tile, ov, minseg, msd, null, four, simple=False):
    nr, nc = img.shape[1:]
    tiles, ntc, ntr = oracle.get_tiles(nr, nc, tile, ov)
    local = {}
    for (c, r), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        local[(c, r)] = oracle.segment_tile(sub, centres, minseg, msd, null, four)['segimg']
    return oracle.stitch_tiles(local, tiles, ntc, ntr, nr, nc, ov, simple=simple)


@pytest.mark.parametrize('simple', [False, True])
def test_tiled_device_raster_vs_oracle(simple, oracle):
    """synthetic raster generated in HBM, whole-image k-means (subsample + device Lloyd),
    4 worker streams; compared with the oracle tile by tile + oracle stitch."""
    from pyshepseg_amd import tiling
    ras = tiling.DeviceRaster.synth(11, 6, 1500, 1300)
    try:
        img = ras.toArray()
        assert np.array_equal(img, oracle.synthimg(11, 6, 1500, 1300))
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=4)
        r = tiling.doTiledShepherdSegmentation(ras, None, tileSize=512, overlapSize=128,
                                               minSegmentSize=50, numClusters=30,
                                               fixedKMeansInit=True, simpleTileRecode=simple,
                                               concurrencyCfg=cfg)
    finally:
        ras.free()
    want, mx, hist = _oracle_tiled(oracle, img, r.kmeans.cluster_centers_, 512, 128, 50,
                                   float(r.maxSpectralDiff), None, True, simple=simple)
    assert r.maxSegId == mx
    assert np.array_equal(r.segimg, want)
    assert np.array_equal(r.hist, hist)
    # size-independent properties of a stitched result
    assert r.hist[1:].sum() == (r.segimg != 0).sum()
    assert r.subsamplePcnt == pytest.approx(100 * min(1, np.sqrt(1e6 / (1500 * 1300))) ** 2)


def test_tiled_npy_roundtrip(tmp_path, oracle):
    from pyshepseg_amd import tiling, shepseg
    img = oracle.synthimg(5, 3, 400, 420)
    np.save(tmp_path / 'in.npy', img)
    km = shepseg.fitSpectralClusters(img, 8, 5, None, True)
    r = tiling.doTiledShepherdSegmentation(str(tmp_path / 'in.npy'), str(tmp_path / 'out.npy'),
                                           tileSize=128, overlapSize=32, minSegmentSize=20,
                                           kmeansObj=km)
    out = np.load(tmp_path / 'out.npy')
    want, mx, hist = _oracle_tiled(oracle, img, km.cluster_centers_, 128, 32, 20,
                                   float(r.maxSpectralDiff), None, True)
    assert np.array_equal(out, want) and r.maxSegId == mx
    assert np.array_equal(np.load(tmp_path / 'out_hist.npy'), hist)
