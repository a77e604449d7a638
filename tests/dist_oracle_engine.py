"""Test-only engine for pyshepseg_amd.distributed.runDistributed built on the CPU oracle, so
that the sharding / boundary-exchange logic of the multi-GPU driver can be exercised with
world_size 2 over gloo on a machine without GPUs."""
import numpy as np


class _KM(object):
    def __init__(self, c):
        self.cluster_centers_ = c


class OracleEngine(object):
    def __init__(self, img, oracle):
        self.img = img              # the whole raster; only this rank's rows are touched
        self.orc = oracle

    def setup(self, tileInfo, jobs, total, yLo, yHi, outLo, outHi, nCols, overlapSize):
        self.tileInfo, self.jobs = tileInfo, jobs
        (self.yLo, self.yHi, self.outLo, self.outHi) = (yLo, yHi, outLo, outHi)
        self.nCols, self.ov = nCols, overlapSize
        self.slice = self.img[:, yLo:yHi]
        self.out = np.zeros((outHi - outLo, nCols), dtype=np.uint32)
        self.local, self.recoded = {}, {}
        self.maxSeg = 0
        self.touched = set()

    def subsample(self, rowsGlobal, cols):
        assert len(rowsGlobal) == 0 or (rowsGlobal.min() >= self.yLo and rowsGlobal.max() < self.yHi)
        return np.ascontiguousarray(self.slice[:, rowsGlobal - self.yLo][:, :, cols])

    def fit(self, img, numClusters, imgNullVal, fixedKMeansInit):
        from pyshepseg_amd import shepseg
        xs = shepseg._sample_rows(img, 100, imgNullVal)
        init = shepseg.diagonalClusterCentres(xs, numClusters).astype(np.float64)
        c, _l, _n = self.orc.kmeans_fit(xs.astype(np.float64), init, algorithm='elkan')     # the reference's algorithm
        return _KM(c)

    def startSegmentation(self, centres, msd, imgNullVal, fourConnected, minSegmentSize):
        for j in self.jobs:
            sub = np.ascontiguousarray(self.slice[:, j.ypos - self.yLo:j.ypos - self.yLo + j.ysize,
                                                  j.xpos:j.xpos + j.xsize])
            self.local[(j.col, j.row)] = self.orc.segment_tile(
                sub, centres, minSegmentSize, float(msd), imgNullVal, fourConnected)['segimg']

    def waitTile(self, j):
        pass

    def setMaxSegId(self, v):
        self.maxSeg = int(v)

    def getMaxSegId(self):
        return self.maxSeg

    def bottomStripOf(self, a):
        return self.recoded[(a.col, a.row)][-self.ov:, :]

    def rightStripOf(self, a):
        return self.recoded[(a.col, a.row)][:, -self.ov:]

    def stitchTile(self, j, top, left, win, simple):
        (t, b, l, r, xout, yout) = win
        tile = self.local[(j.col, j.row)]
        if simple:
            rec = np.where(tile == 0, 0, tile + np.uint32(self.maxSeg)).astype(np.uint32)
        else:
            rec = self.orc.recode_tile(tile, self.ov, top, left, self.maxSeg, t, b, l, r)
        self.recoded[(j.col, j.row)] = rec
        trimmed = rec[t:b, l:r]
        self.out[yout - self.outLo:yout - self.outLo + trimmed.shape[0],
                 xout:xout + trimmed.shape[1]] = trimmed
        self.maxSeg = max(self.maxSeg, int(trimmed.max()))

    # ---- parallel stitch ----
    def beginProvisional(self, stride, ntAll):
        self.counts = {}

    def stitchTileAt(self, j, top, left, win, t, stride, slot):
        base = t * stride
        self.maxSeg = base
        self.stitchTile(j, top, left, win, False)
        rec = self.recoded[(j.col, j.row)]
        (tt, b, l, r, _x, _y) = win
        k = max(0, int(rec.max()) - base)
        trimmed = rec[tt:b, l:r]
        rr = max(0, int(trimmed.max()) - base) if trimmed.size else 0
        self.counts[slot] = (k, rr)

    def tileCounts(self, n):
        return [self.counts[i] for i in range(n)]

    def renumber(self, stride, base):
        v = self.out
        t = (v // np.uint32(stride)).astype(np.int64)
        self.out = np.where(v == 0, 0, base[np.minimum(t, len(base) - 1)] + (v % np.uint32(stride))).astype(np.uint32)

    def renumberKept(self, stride, base, keptJobs, recvStrips):
        def fix(v):
            t = (v // np.uint32(stride)).astype(np.int64)
            return np.where(v == 0, 0, base[np.minimum(t, len(base) - 1)] + (v % np.uint32(stride))).astype(np.uint32)
        self.renumber(stride, base)
        for j in keptJobs:
            self.recoded[(j.col, j.row)] = fix(self.recoded[(j.col, j.row)])
        for a in recvStrips:
            a[...] = fix(a)
        self.chainRedone = len(self.jobs) - len(keptJobs)

    def sendStrip(self, comm, dst, item, a):
        (kind, _c, _r, h, w) = item
        s = self.bottomStripOf(a) if kind == 'b' else self.rightStripOf(a)
        assert s.shape == (h, w)
        comm.send_bytes(np.ascontiguousarray(s, dtype=np.uint32), dst)

    def recvStrip(self, comm, src, item):
        (_kind, _c, _r, h, w) = item
        return np.frombuffer(comm.recv_bytes(src), dtype=np.uint32).reshape(h, w).copy()

    def sendBoundary(self, comm, dst, maxSegId, items):
        comm.send_obj(int(maxSegId), dst)
        for (kind, a, h, w) in items:
            s = self.bottomStripOf(a) if kind == 'b' else self.rightStripOf(a)
            assert s.shape == (h, w)
            comm.send_bytes(np.ascontiguousarray(s, dtype=np.uint32), dst)

    def recvBoundary(self, comm, src, plan):
        maxSegId = int(comm.recv_obj(src))
        strips = {}
        for (kind, col, row, h, w) in plan:
            strips[(kind, col, row)] = np.frombuffer(comm.recv_bytes(src), dtype=np.uint32).reshape(h, w).copy()
        return maxSegId, strips

    def histogram(self, maxSegId):
        return np.bincount(self.out.ravel(), minlength=maxSegId + 1)[:maxSegId + 1]

    def finish(self):
        pass

    # ---- per-segment statistics (calcPerSegmentStatsDistributed) ----
    _NAMES = {0: 'min', 1: 'max', 2: 'mean', 3: 'stddev', 4: 'median', 5: 'mode', 6: 'percentile',
              7: 'pixcount'}

    def _sel(self, fast):
        return [('c%d' % i, self._NAMES[int(r[1])], int(r[4])) if int(r[1]) == 6
                else ('c%d' % i, self._NAMES[int(r[1])]) for i, r in enumerate(fast)]

    def _band(self, imgbandnum):
        return np.ascontiguousarray(self.img[imgbandnum - 1, self.outLo:self.outHi])

    def localStats(self, imgbandnum, S, fast, nInt, nFloat, missing, imgNullVal):
        return self.orc.segstats(self.out, self._band(imgbandnum), self._sel(fast), imgNullVal,
                                 missing, max_seg_id=S)

    def gatherFlagged(self, imgbandnum, S, flags, count):
        m = flags[self.out] != 0
        m &= self.out != 0
        assert int(m.sum()) == count
        return self.out[m].astype(np.uint32), self._band(imgbandnum)[m].astype(np.int64)

    def statsOfPairs(self, segs, vals, K, fast, nInt, nFloat, missing, imgNullVal):
        return self.orc.segstats(segs, vals.astype(self.img.dtype), self._sel(fast), imgNullVal,
                                 missing, max_seg_id=K)
