"""Extreme-value check of findNearestNeighbourPixel (shepseg.py:677-736) for 32-bit imagery: is the
reference's dSqr the wrapping int64 sum of squared exact differences, compared as the reference
writes it (`minDsqr < 0 or dSqr < minDsqr`, so a sum that wrapped negative counts as "unset")?"""
import numpy as np
import refenv
from refenv import shepseg

M = (1 << 64)


def model(img, seg, i, j, segSize, four):
    (nb, nr, nc) = img.shape
    mind = -1
    (ii, jj) = (-1, -1)
    for a in range(max(i - 1, 0), min(i + 1, nr - 1) + 1):
        for b in range(max(j - 1, 0), min(j + 1, nc - 1) + 1):
            if (not four) or a == i or b == j:
                if segSize[seg[a, b]] > 1:
                    s = 0
                    for k in range(nb):
                        d = int(img[k, i, j]) - int(img[k, a, b])
                        s = (s + d * d) % M
                    if s >= (1 << 63):
                        s -= M                       # two's complement int64
                    if mind < 0 or s < mind:
                        (mind, ii, jj) = (s, a, b)
    return (ii, jj)


rng = np.random.default_rng(7)
bad = 0
n = 0
for dt in (np.uint32, np.int32):
    info = np.iinfo(dt)
    for case in range(4000):
        nb = int(rng.integers(1, 9))
        img = rng.integers(info.min, int(info.max) + 1, size=(nb, 3, 3), dtype=np.int64).astype(dt)
        if case % 3 == 0:                            # values at the limits
            img = np.where(rng.random(img.shape) < 0.5, info.max, info.min).astype(dt)
        seg = np.arange(1, 10, dtype=np.uint32).reshape(3, 3)
        segSize = rng.integers(1, 3, size=10).astype(np.uint32)
        segSize[5] = 1
        four = bool(case & 1)
        got = shepseg.findNearestNeighbourPixel(img, seg, 1, 1, segSize, four)
        want = model(img, seg, 1, 1, segSize, four)
        n += 1
        if tuple(int(x) for x in got) != want:
            bad += 1
            if bad < 5:
                print('differs', np.dtype(dt).name, nb, got, want)
print(refenv.STACK)
print('cases %d, reference != wrapping-int64 model: %d' % (n, bad))
