"""Test data of the reference's own CI scenario (pyshepseg/cmdline/runtests.py:27-265), restated: a
Voronoi partition of an n x n raster around 100 fixed centres, a 10-pixel null border, and a 3-band
uint16 image that paints every cell with its own colour (null = 65535).  Shared by
tests/test_gpu_ci_scenario.py and oracle/refgen/gen_golden_ci_scenario.py; plain numpy, no GDAL.
The centre table is the reference's data (runtests.py:27-37); `scale` shrinks the scenario (the
reference runs it at 8000 x 8000 = scale 1)."""
import numpy as np

NBANDS = 3
NULLVAL = 65535
MARGIN = 10
CENTRES = np.array([
    (116, 3495), (142, 3100), (236, 6033), (290, 796), (297, 6152), (310, 5318), (409, 5867), (410, 2125),
    (442, 2913), (472, 1135), (486, 5296), (628, 667), (655, 2677), (672, 4001), (677, 5513), (736, 3720),
    (913, 3552), (1056, 347), (1085, 3391), (1121, 6623), (1150, 1906), (1196, 5663), (1694, 3244),
    (1761, 2172), (1761, 7460), (1882, 6151), (1893, 626), (2014, 433), (2065, 3157), (2132, 378),
    (2161, 2352), (2200, 7485), (2393, 5191), (2489, 2519), (2508, 1575), (2509, 7089), (2599, 3151),
    (2645, 2672), (2782, 3380), (2906, 3676), (3072, 2934), (3133, 3418), (3188, 1653), (3624, 7812),
    (3661, 3603), (3694, 2929), (3759, 3418), (4155, 630), (4233, 4753), (4423, 1377), (4427, 6635),
    (4462, 7392), (4715, 6908), (4856, 2559), (4898, 3371), (5051, 2268), (5064, 5969), (5071, 2019),
    (5107, 3533), (5172, 5478), (5294, 4210), (5305, 1512), (5310, 2846), (5365, 3715), (5447, 6215),
    (5513, 5017), (5549, 297), (5579, 4076), (5623, 5044), (5688, 3614), (5728, 1802), (5747, 7801),
    (5758, 4377), (5779, 4148), (5784, 3239), (5812, 5091), (5862, 4664), (5897, 4963), (6299, 4702),
    (6320, 6936), (6462, 2844), (6615, 4979), (6726, 5970), (6754, 7652), (6765, 714), (6826, 3162),
    (6827, 3770), (6844, 1170), (6884, 226), (7023, 213), (7094, 6472), (7157, 647), (7196, 7710),
    (7293, 7588), (7495, 5912), (7693, 3966), (7718, 7759), (7737, 6002), (7745, 1347), (7889, 2850)],
    dtype=np.int64)


def true_segments(n=8000, scale=1, block=250, workers=8):
    """generateTrueSegments (runtests.py:145-195): every pixel takes the id of its closest centre, the
    centres visited in table order, the running minimum kept in float32 as the reference keeps it (a
    later centre wins only when its float64 distance is below the float32-rounded minimum so far); the
    first coordinate of a centre pairs with the ROW index, as numpy.mgrid hands it out there.
    Row blocks are independent (the arithmetic is per pixel), a few threads share them."""
    from concurrent.futures import ThreadPoolExecutor
    cen = CENTRES // scale
    seg = np.zeros((n, n), dtype=np.uint32)
    cols = np.arange(n, dtype=np.int64)[None, :]

    def one(r0):
        rows = np.arange(r0, min(n, r0 + block), dtype=np.int64)[:, None]
        mind = np.full((rows.shape[0], n), 10.0 * n, dtype=np.float32)
        sblk = np.zeros((rows.shape[0], n), dtype=np.uint32)
        for i in range(len(cen)):
            dist = np.sqrt(((rows - cen[i, 0]) ** 2 + (cols - cen[i, 1]) ** 2).astype(np.float64))
            closer = dist < mind
            sblk[closer] = i + 1
            mind[closer] = dist[closer]
        seg[r0:r0 + rows.shape[0]] = sblk

    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(one, range(0, n, block)))
    m = MARGIN
    seg[:m, :] = 0
    seg[-m:, :] = 0
    seg[:, :m] = 0
    seg[:, -m:] = 0
    return seg


def palette(numSeg):
    """createPallete (runtests.py:198-227): distinct made-up colours in [0, 10000], uint16 (numSeg, 3)"""
    (lo, hi) = (0, 10000)
    step = (hi - lo) / (numSeg - 1)
    mid = numSeg / 2
    c = np.zeros((numSeg, NBANDS), dtype=np.uint16)
    for i in range(numSeg):
        c[i, 0] = round(lo + i * step)
        c[i, 1] = round(hi - i * step)
        c[i, 2] = round(lo + i * 2 * step) if i < mid else round(hi - (i - mid) * 2 * step)
    return c


def multispectral(trueseg):
    """createMultispectral (runtests.py:230-265): band b of a pixel = its cell's colour, nulls 65535"""
    pal = palette(int(trueseg.max()))
    lut = np.full((int(trueseg.max()) + 1, NBANDS), NULLVAL, dtype=np.uint16)
    lut[1:] = pal
    return np.ascontiguousarray(np.transpose(lut[trueseg], (2, 0, 1)))
