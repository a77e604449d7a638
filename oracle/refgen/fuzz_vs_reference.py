"""Pin the C oracle against the unmodified reference on many random inputs.

    /opt/conda/bin/python3.9 oracle/refgen/fuzz_vs_reference.py [ncases] [wide]

`wide`: 32-bit imagery (uint32 and int32) over the types' whole range and at their limits, where
the reference's integer arithmetic wraps (findNearestNeighbourPixel) and float32 sums are inexact.

Stage by stage (assign, clump, single-pixel elimination, small-segment elimination, and the
k-means fit: n_iter_, labels_, cluster_centers_) the oracle must equal the reference bit for
bit.  Run with OMP_NUM_THREADS=1 (the fit's sums).  Build container only; see refenv.py.
"""
import sys
import time

import numpy as np

import refenv
from refenv import shepseg
from oracle import oracle


class FakeKM(object):
    def __init__(self, centres):
        self.cluster_centers_ = centres


def make_img_wide(rng, case):
    dt = np.uint32 if case % 2 == 0 else np.int32
    info = np.iinfo(dt)
    nb = int(rng.choice([1, 2, 3, 6, 8]))
    nr = int(rng.randint(5, 110))
    nc = int(rng.randint(5, 110))
    kind = (case // 2) % 4
    levels = np.array([info.min, info.min + 1, int(info.min) // 2 + int(info.max) // 2, info.max - 1, info.max],
                      dtype=np.int64)
    if kind == 0:      # blocks of limit values with single-pixel noise of limit values
        base = levels[rng.randint(0, 5, size=(nb, nr // 5 + 1, nc // 5 + 1))]
        img = np.kron(base, np.ones((1, 5, 5), dtype=np.int64))[:, :nr, :nc]
        m = rng.rand(nr, nc) < 0.15
        img = np.where(m[None], levels[rng.randint(0, 5, size=(nb, nr, nc))], img)
    elif kind == 1:    # the whole range, uniformly
        img = rng.randint(info.min, int(info.max) + 1, size=(nb, nr, nc), dtype=np.int64)
    elif kind == 2:    # smooth, scaled to the range
        img = oracle.synthimg(int(rng.randint(1, 1000)), nb, nr, nc).astype(np.int64)
        img = img * ((int(info.max) - int(info.min)) // 65536) + int(info.min)
    else:              # few levels far apart + small noise
        img = levels[rng.randint(0, 5, size=(nb, nr, nc))] // 3 * 2 + rng.randint(0, 4, size=(nb, nr, nc))
    img = np.clip(img, info.min, info.max).astype(dt)
    null_val = None
    if rng.rand() < 0.5:
        null_val = int(info.max) if rng.rand() < 0.7 else int(info.min)
        m = rng.rand(nr, nc) < rng.choice([0.01, 0.1])
        img[rng.randint(0, nb)][m] = null_val
    return np.ascontiguousarray(img), null_val


def make_img(rng, case):
    kind = case % 8
    nb = int(rng.choice([1, 3, 4, 6, 7]))
    nr = int(rng.randint(5, 140))
    nc = int(rng.randint(5, 140))
    if kind == 0:      # smooth synthetic
        img = oracle.synthimg(int(rng.randint(1, 1000)), nb, nr, nc, int(rng.randint(0, 5000)),
                              int(rng.randint(0, 5000)))
    elif kind == 1:    # few grey levels -> big clumps, many ties
        img = (rng.randint(0, 3, size=(nb, nr, nc)) * 1000 + 500).astype(np.uint16)
    elif kind == 2:    # high values: float32 sums inexact, squares large
        img = (60000 + rng.randint(0, 5000, size=(nb, nr, nc))).astype(np.uint16)
        img[:, : nr // 2, :] = img[:, :1, :1]      # one big flat segment
    elif kind == 3:    # uint8 noise
        img = rng.randint(0, 256, size=(nb, nr, nc)).astype(np.uint8)
    elif kind == 4:    # int16 with negatives
        img = rng.randint(-3000, 3000, size=(nb, nr, nc)).astype(np.int16)
    elif kind == 5:    # blocky
        base = rng.randint(0, 4000, size=(nb, nr // 6 + 1, nc // 6 + 1))
        img = np.kron(base, np.ones((1, 6, 6), dtype=np.int64))[:, :nr, :nc]
        img = (img + rng.randint(0, 30, size=img.shape)).astype(np.uint16)
    elif kind == 6:    # int32 large
        img = rng.randint(0, 1 << 20, size=(nb, nr, nc)).astype(np.int32)
    else:              # thin shapes
        if rng.rand() < 0.5:
            nr = 1
        else:
            nc = 1
        img = rng.randint(0, 2000, size=(nb, nr, nc)).astype(np.uint16)
    null_val = None
    if rng.rand() < 0.5:
        null_val = int(np.iinfo(img.dtype).max)
        m = rng.rand(nr, nc) < rng.choice([0.01, 0.1, 0.4])
        if rng.rand() < 0.3:
            m[:] = False
            m[rng.randint(0, nr), rng.randint(0, nc)] = True     # segSize[0]==1
        b = rng.randint(0, nb)
        img[b][m] = null_val
    return np.ascontiguousarray(img), null_val


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    wide = len(sys.argv) > 2 and sys.argv[2] == 'wide'
    rng = np.random.RandomState(54321 if wide else 12345)
    nfail = 0
    t0 = time.time()
    for case in range(ncases):
        img, null_val = make_img_wide(rng, case) if wide else make_img(rng, case)
        nb, nr, nc = img.shape
        k = int(rng.choice([2, 5, 10, 60]))
        min_seg = int(rng.choice([2, 5, 20, 50]))
        four = bool(rng.rand() < 0.6)
        pcnt = int(rng.choice([1, 10, 50, 100]))
        try:
            ref, km = refenv.ref_stages(img, k, min_seg, null_val, four, pcnt=pcnt)
        except Exception as e:       # e.g. too few non-null samples for k clusters
            print('case %d skipped: %s' % (case, str(e)[:80]))
            continue
        c = ref['centres']
        tag = 'case %d %s %s k=%d minSeg=%d four=%s null=%s' % (case, img.shape, img.dtype, k,
                                                                 min_seg, four, null_val)
        ok = True
        cl = oracle.kmeans_assign(img, c, null_val)
        if not np.array_equal(cl, ref['clusters']):
            print('ASSIGN MISMATCH', tag, (cl != ref['clusters']).sum()); ok = False
        seg, nxt = oracle.clump(ref['clusters'], 0, four, 1)
        if not np.array_equal(seg, ref['clump']) or nxt - 1 != ref['num_clumps']:
            print('CLUMP MISMATCH', tag); ok = False
        seg1 = ref['clump'].copy()
        ss = oracle.make_seg_size(seg1)
        if not np.array_equal(ss, shepseg.makeSegSize(ref['clump'])):
            print('SEGSIZE MISMATCH', tag); ok = False
        oracle.eliminate_single_pixels(img, seg1, ss, 1, int(ref['num_clumps']), four)
        if not np.array_equal(seg1, ref['seg_single']):
            print('SINGLE MISMATCH', tag, (seg1 != ref['seg_single']).sum()); ok = False
        seg2 = ref['seg_single'].copy()
        ne = oracle.eliminate_small_segments(seg2, img, int(seg2.max()), min_seg,
                                             float(ref['msd']), four)
        if not np.array_equal(seg2, ref['seg_final']) or ne != ref['num_small']:
            print('SMALL MISMATCH', tag, (seg2 != ref['seg_final']).sum(), ne, ref['num_small'])
            ok = False
        full = oracle.segment_tile(img, c, min_seg, float(ref['msd']), null_val, four)
        if not np.array_equal(full['segimg'], ref['seg_final']):
            print('FULL MISMATCH', tag); ok = False
        # k-means fit, the reference's algorithm (Elkan's, sklearn 0.24.2 algorithm="auto"): n_iter_, labels_ and
        # cluster_centers_ bit for bit.  Needs OMP_NUM_THREADS=1 (with more threads sklearn's M-step sums are
        # added in the order its threads finish).
        x = np.transpose(img, (1, 2, 0)).reshape(nr * nc, nb)
        if null_val is not None:
            x = x[(x != null_val).all(axis=1)]
        xs = x[::int(round(100. / pcnt))]
        init = shepseg.diagonalClusterCentres(xs, k)
        cfit, lab, nit = oracle.kmeans_fit(xs.astype(np.float64), init.astype(np.float64), algorithm='elkan')
        if nit != km.n_iter_ or not np.array_equal(lab, km.labels_) or \
                not np.array_equal(cfit.view(np.uint64), np.asarray(km.cluster_centers_, dtype=np.float64).view(np.uint64)):
            print('KMFIT MISMATCH', tag, nit, km.n_iter_, int((lab != km.labels_).sum()),
                  float(np.abs(cfit - km.cluster_centers_).max()))
            ok = False
        nfail += (not ok)
        if case % 20 == 0:
            print('case', case, 'ok' if ok else 'FAIL', tag, 'clumps', int(ref['num_clumps']),
                  'final', int(ref['seg_final'].max()), '%.0fs' % (time.time() - t0))
            sys.stdout.flush()
    print('stack:', refenv.STACK)
    print('DONE: %d cases, %d failures' % (ncases, nfail))
    return 1 if nfail else 0


if __name__ == '__main__':
    sys.exit(main())
