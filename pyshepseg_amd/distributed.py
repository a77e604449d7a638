"""Multi-GPU tiled Shepherd segmentation: one process per GPU, RCCL over xGMI bound directly
behind the C-ABI (pyshepseg_amd/comm.py: RcclComm; SocketComm in the CPU tests).

What shards and what does not
-----------------------------
* Tiles are independent once the global k-means model is known (reference tiling.py:1430-1453),
  so the tiles, in row-major order, are dealt to ranks in contiguous ranges balanced by area
  (whole tile rows when there are too few tiles per rank); every rank holds only the rows of the
  raster its tiles need and the rows of the stitched output they write.  No collective on that
  path.
* The cross-tile stitch is specified sequentially (reference stitchTiles, tiling.py:979-1043:
  every tile's new ids start after the largest id of all earlier tiles, and shared segments
  take the id the tile above / to the left already gave them).  Two forms, same result:
  - sequential: rank r stitches its tiles once it has received, from the rank holding the tiles
    before, the running maxSegId, the recoded bottom overlap strips of that rank's last ncols
    tiles (the top neighbours of this rank's first tiles) and, when this rank starts in the middle
    of a tile row, the right strip of the tile before -- point-to-point send/recv, <= 25 MB per
    strip -- and passes its own on at the end;
  - parallel (default with more than one rank): tile t numbers its new segments from a
    PROVISIONAL base t * stride (stride = 2^32 / number of tiles), so its chain step needs the
    strips of the tiles above and to the left only -- a rank starts as soon as the first strip of
    the previous rank's last row arrives, strips travel tile by tile, and a rank takes its tiles
    along anti-diagonals so that the tiles of its last row are ready two steps apart instead of a
    row apart (SHEPSEG_CHAIN_ORDER=rowmajor: the old order).  Every tile reports K_t =
    ids handed out and R_t = the largest of them present in its trimmed window; an all-gather
    later, if K_t == R_t everywhere, the sequential run would have found maxSegId = sum of the
    earlier K (the reference advances it to trimmed.max()), the provisional ids are in the same
    order as the final ones (ties in the mode are broken by id) and no two of them collide, so
    id -> base[id / stride] + id % stride over the output gives the identical raster.  A tile
    with K_t != R_t (the reference then reuses ids: tests/golden/stitch_quirk_empties) or with
    more local segments than the stride sends every rank back to the sequential form.
  A final all-reduce sums the per-rank histograms.
* The k-means fit: every rank gets the whole sub-sample (all-gathered from the ranks' slices).  On device
  transports (RCCL) the fit's E-step is sharded by sample rows over the ranks, its labels all-gathered on the
  device every iteration, and every rank runs the same M-step: the same model everywhere, no broadcast
  (shp_kmeans_fit_planar_dist).  Other transports fit on rank 0 and broadcast the centres.

The driver below is engine-agnostic: ``HipEngine`` drives libshepseg_hip.so on this rank's GPU;
the CPU tests plug in an engine built on the oracle to exercise the sharding / exchange logic
with world_size 2 over gloo.
"""
import ctypes
import sys
import json
import os
import time

import numpy

from . import _lib
from . import shepseg
from . import tiling


# ------------------------------------------------------------------------------------------
# sharding
# ------------------------------------------------------------------------------------------
def shardTileRows(tileInfo, world):
    """Contiguous blocks of tile rows per rank, balanced by pixel area.  Returns a list of
    (row0, row1) half-open ranges, one per rank (empty when there are more ranks than rows)."""
    nrows = tileInfo.nrows
    weights = []
    for r in range(nrows):
        weights.append(sum(tileInfo.getTile(c, r)[2] * tileInfo.getTile(c, r)[3]
                           for c in range(tileInfo.ncols)))
    total = float(sum(weights))
    out = []
    r = 0
    acc = 0.0
    for k in range(world):
        r0 = r
        remainingRanks = world - k
        if nrows - r <= 0:
            out.append((r, r))
            continue
        target = total * (k + 1) / world
        # take rows while it brings the running sum closer to this rank's share, keeping at
        # least one row for every later rank that can still get one
        while r < nrows and (nrows - r) > (remainingRanks - 1):
            if r > r0 and abs(acc + weights[r] - target) > abs(acc - target):
                break
            acc += weights[r]
            r += 1
        if r == r0 and r < nrows:
            acc += weights[r]
            r += 1
        out.append((r0, r))
    if r < nrows:                      # leftovers go to the last rank that has rows
        last = max(i for i, (a, b) in enumerate(out) if b > a)
        out[last] = (out[last][0], nrows)
    return out


def shardTiles(tileInfo, world, wholeRows=False):
    """Contiguous ranges [t0, t1) of tiles (row-major index row * ncols + col) per rank, balanced
    by pixel area.  Every rank that has a successor holds at least ncols tiles, so a tile's top
    neighbour is either local or in the previous rank; with fewer than ncols tiles per rank, or
    with wholeRows, the ranges are whole tile rows (shardTileRows), which has the same property.
    Whole rows are what the parallel stitch wants: a rank whose range starts in the middle of a
    tile row needs the right strip of the previous rank's LAST tile before its first chain step,
    whereas a row's first tile only needs the tile above it, which the previous rank stitches at
    the start of its last row -- the ranks then work one tile behind each other."""
    (ncols, nrows) = (tileInfo.ncols, tileInfo.nrows)
    nt = ncols * nrows
    if world <= 1:
        return [(0, nt)]
    if nt // world < ncols or wholeRows:
        return [(a * ncols, b * ncols) for (a, b) in shardTileRows(tileInfo, world)]
    weights = []
    for r in range(nrows):
        for c in range(ncols):
            (_x, _y, xs, ys) = tileInfo.getTile(c, r)
            weights.append(xs * ys)
    total = float(sum(weights))
    out = []
    i = 0
    acc = 0.0
    for k in range(world):
        i0 = i
        later = world - k - 1
        if k == world - 1:
            i = nt
        else:
            target = total * (k + 1) / world
            while i < nt - later * ncols:
                if i - i0 >= ncols and abs(acc + weights[i] - target) > abs(acc - target):
                    break
                acc += weights[i]
                i += 1
        out.append((i0, i))
    return out


def boundaryPlan(tileInfo, shards, p, overlapSize):
    """What rank p hands to the next rank that has tiles: a list of (kind, col, row, h, w) --
    'b' the recoded bottom strip (h x w) of a tile that is the top neighbour of one of the next
    rank's tiles, 'r' the right strip of rank p's last tile when the next rank starts in the
    middle of that tile row.  Both sides derive it from the shard table alone."""
    ncols = tileInfo.ncols
    nonEmpty = [i for i, (a, b) in enumerate(shards) if b > a]
    pos = nonEmpty.index(p)
    if pos + 1 >= len(nonEmpty):
        return []
    (p0, p1) = shards[p]
    (q0, q1) = shards[nonEmpty[pos + 1]]
    plan = []
    for t in range(max(p0, q0 - ncols), p1):
        if t + ncols < q1:                     # its bottom neighbour belongs to the next rank
            (col, row) = (t % ncols, t // ncols)
            (_x, _y, xs, ys) = tileInfo.getTile(col, row)
            plan.append(('b', col, row, min(overlapSize, ys), xs))
    if q0 % ncols != 0:
        (col, row) = ((p1 - 1) % ncols, (p1 - 1) // ncols)
        (_x, _y, xs, ys) = tileInfo.getTile(col, row)
        plan.append(('r', col, row, ys, min(overlapSize, xs)))
    return plan


def Comm(_unused=None, device=None):
    """World-size-1 communicator (kept for callers that ran the sharded driver in one process);
    multi-rank communicators come from pyshepseg_amd.comm."""
    from . import comm as _comm
    return _comm.LocalComm()


# ------------------------------------------------------------------------------------------
# the engine-agnostic driver
# ------------------------------------------------------------------------------------------
class DistResult(object):
    pass


def runDistributed(engine, comm, nRows, nCols, tileSize, overlapSize, minSegmentSize=50,
                   numClusters=60, subsamplePcnt=None, maxSpectralDiff='auto', imgNullVal=None,
                   fixedKMeansInit=True, fourConnected=True, simpleTileRecode=False,
                   spectDistPcntile=50, kmeansObj=None, stitchMode=None):
    """Tiled segmentation of an (nRows x nCols) raster, its tiles sharded over comm.world ranks.
    ``engine`` owns this rank's slice of the raster and of the output (see HipEngine).  Returns a
    DistResult with maxSegId, hist (global), kmeans, maxSpectralDiff, tileRange (row-major tile
    indices of this rank), rowRange (the tile rows they touch) and outRows (image rows of the
    output buffer this rank holds: its tiles' trimmed windows are written, the rest is 0) and
    stitchMode ('sequential', 'parallel', or 'parallel->sequential' when part of the parallel form
    had to be redone -- chainStepsRedone says how many tiles, from the first one whose ids the
    provisional numbering cannot express; argument / SHEPSEG_STITCH: None = parallel when
    comm.world > 1)."""
    import time as _time
    _t = [_time.time()]
    _marks = []

    def _mark(what):                       # SHEPSEG_IO_TIMING: this rank's milestones of the step, to stderr at its end
        now = _time.time()
        _marks.append('%s %.3f' % (what, now - _t[0]))
        _t[0] = now
    if stitchMode is None:
        stitchMode = os.environ.get('SHEPSEG_STITCH') or ('parallel' if comm.world > 1 else 'sequential')
    if stitchMode not in ('sequential', 'parallel'):
        raise ValueError("stitchMode must be 'sequential' or 'parallel'")
    if simpleTileRecode:
        stitchMode = 'sequential'          # (no shared segments: nothing to gain)
    if (overlapSize % 2) != 0:
        raise tiling.PyShepSegTilingError("Overlap size must be an even number")

    class _Ds(object):
        RasterXSize, RasterYSize = nCols, nRows
    tileInfo = tiling.getTilesForFile(_Ds(), tileSize, overlapSize)
    ncolsT = tileInfo.ncols
    shardBy = os.environ.get('SHEPSEG_SHARD') or ('rows' if stitchMode == 'parallel' else 'tiles')
    shards = shardTiles(tileInfo, comm.world, wholeRows=(shardBy == 'rows'))
    (t0, t1) = shards[comm.rank]
    haveTiles = t1 > t0
    myTiles = [(t % ncolsT, t // ncolsT) for t in range(t0, t1)]
    (r0, r1) = (t0 // ncolsT, (t1 - 1) // ncolsT + 1) if haveTiles else (0, 0)
    jobs, total = tiling.makeTileJobs(tileInfo, tiles=set(myTiles))

    def _winOf(col, row):
        return tiling.trimmedWindow(tileInfo, col, row, *tileInfo.getTile(col, row), overlapSize)
    # image rows this rank needs (its tiles) and writes in the output (their trimmed windows)
    if haveTiles:
        yLo = min(tileInfo.getTile(c, r)[1] for (c, r) in myTiles)
        yHi = max(tileInfo.getTile(c, r)[1] + tileInfo.getTile(c, r)[3] for (c, r) in myTiles)
        wins = [_winOf(c, r) for (c, r) in myTiles]
        outLo = min(w[5] for w in wins)
        outHi = max(w[5] + (w[1] - w[0]) for w in wins)
    else:
        yLo = yHi = outLo = outHi = 0
    # a disjoint split of the image rows for the k-means sample: from the first output row of this
    # rank's first tile to that of the next rank's first tile (inside both ranks' slices)
    nonEmpty = [i for i, (a, b) in enumerate(shards) if b > a]
    firstRow = {i: _winOf(shards[i][0] % ncolsT, shards[i][0] // ncolsT)[5] for i in nonEmpty}
    if haveTiles:
        pos = nonEmpty.index(comm.rank)
        sLo = 0 if pos == 0 else firstRow[comm.rank]
        sHi = nRows if pos + 1 == len(nonEmpty) else max(sLo, firstRow[nonEmpty[pos + 1]])
    else:
        sLo = sHi = 0
    engine.setup(tileInfo, jobs, total, yLo, yHi, outLo, outHi, nCols, overlapSize)
    _mark('setup')

    # ---- one global k-means model (reference tiling.py:154-226) ----
    if kmeansObj is None:
        if subsamplePcnt is None:
            subsampleProp = min(1, numpy.sqrt(1000000 / (nRows * nCols)))
            subsamplePcnt = 100 * subsampleProp**2
        else:
            subsampleProp = numpy.sqrt(subsamplePcnt / 100.0)
        skip = int(round(1. / subsampleProp))
        ry = tiling._subsample_indices(nRows, skip)
        rx = tiling._subsample_indices(nCols, skip)
        mine = ry[(ry >= sLo) & (ry < sHi)]
        part = engine.subsample(mine, rx)                      # (nBands, len(mine), len(rx))
        parts = [p[0].reshape(p[1]) for p in comm.allgather_arrays(
            [numpy.ascontiguousarray(part), numpy.array(part.shape, dtype=numpy.int64)])]
        img = numpy.concatenate([p for p in parts if p.shape[1] > 0], axis=1)
        if (comm.world > 1 and getattr(comm, 'onDevice', False) and hasattr(comm, 'h') and fixedKMeansInit and
                hasattr(engine, 'fitSharded') and os.environ.get('SHEPSEG_FIT_SHARDED', '1') != '0'):
            # every rank holds the whole sample (12 MB for a 40000^2 raster); the E-step of the fit is sharded by
            # sample rows over the ranks, its labels all-gathered on the device every iteration, the M-step run by
            # all of them alike: the same model on every rank, no broadcast (shp_kmeans_fit_planar_dist)
            km = engine.fitSharded(img, numClusters, imgNullVal, comm)
            centres = numpy.ascontiguousarray(km.cluster_centers_, dtype=numpy.float64)
        else:
            centres = None
            if comm.rank == 0:
                km = engine.fit(img, numClusters, imgNullVal, fixedKMeansInit)
                centres = numpy.ascontiguousarray(km.cluster_centers_, dtype=numpy.float64)
            centres = comm.bcast_obj(centres, src=0)
        kmeansObj = shepseg.KMeansModel(centres)
    centres = numpy.ascontiguousarray(kmeansObj.cluster_centers_, dtype=numpy.float64)
    msd = shepseg.autoMaxSpectralDiff(kmeansObj, maxSpectralDiff, spectDistPcntile)

    # ---- segment this rank's tiles (asynchronously) ----
    _mark('model')
    engine.startSegmentation(centres, msd, imgNullVal, fourConnected, minSegmentSize)
    _mark('workers started')

    # ---- the stitch ----
    jobmap = {(j.col, j.row): j for j in jobs}
    pos = nonEmpty.index(comm.rank) if haveTiles else -1
    prevRank = nonEmpty[pos - 1] if pos > 0 else None
    nextRank = nonEmpty[pos + 1] if haveTiles and pos + 1 < len(nonEmpty) else None

    def _neighbours(j, fromPrev):
        top = left = None
        if not simpleTileRecode:
            if j.row > 0:
                a = jobmap.get((j.col, j.row - 1))
                top = engine.bottomStripOf(a) if a is not None else fromPrev[('b', j.col, j.row - 1)]
            if j.col > 0:
                a = jobmap.get((j.col - 1, j.row))
                left = engine.rightStripOf(a) if a is not None else fromPrev[('r', j.col - 1, j.row)]
        return top, left

    def _sequential():
        maxSegId = 0
        if haveTiles:
            fromPrev = {}
            if prevRank is not None:
                maxSegId, fromPrev = engine.recvBoundary(
                    comm, prevRank, boundaryPlan(tileInfo, shards, prevRank, overlapSize))
            engine.setMaxSegId(maxSegId)
            for j in jobs:
                engine.waitTile(j)
                (top, left) = _neighbours(j, fromPrev)
                engine.stitchTile(j, top, left, _winOf(j.col, j.row), simpleTileRecode)
            _mark('chain issued')
            maxSegId = engine.getMaxSegId()
            _mark('chain done')
            if nextRank is not None:
                plan = boundaryPlan(tileInfo, shards, comm.rank, overlapSize)
                engine.sendBoundary(comm, nextRank, maxSegId,
                                    [(kind, jobmap[(c, r)], h, w) for (kind, c, r, h, w) in plan])
        # final maxSegId lives on the last rank that has tiles
        vals = comm.allgather_obj(int(maxSegId))
        return vals[nonEmpty[-1]] if nonEmpty else 0

    def _parallel():
        """Returns the final maxSegId, or None when some tile makes the provisional form unsafe."""
        ntAll = ncolsT * tileInfo.nrows
        stride = 0xFFFFFFFF // max(ntAll, 1)
        mine = []
        if haveTiles:
            # With provisional ids a chain step needs its two neighbours only, so a rank takes its
            # tiles along anti-diagonals (row + col ascending): the tiles of its LAST row are then
            # done two steps apart instead of a row apart, and the next rank, which waits for them
            # one by one, follows two steps behind instead of a row behind.  Strips cross the rank
            # boundary tile by tile in that order ('b' before 'r'); both sides derive it.
            wave = os.environ.get('SHEPSEG_CHAIN_ORDER', 'diagonal') != 'rowmajor'
            tkey = (lambda c, r: (r + c, r)) if wave else (lambda c, r: (r * ncolsT + c, 0))
            order = lambda plan: sorted(plan, key=lambda it: tkey(it[1], it[2]) + (it[0] != 'b',))
            planPrev = order(boundaryPlan(tileInfo, shards, prevRank, overlapSize)) if prevRank is not None else []
            sendOf = {}
            if nextRank is not None:
                for it in order(boundaryPlan(tileInfo, shards, comm.rank, overlapSize)):
                    sendOf.setdefault((it[1], it[2]), []).append(it)
            fromPrev = {}
            got = [0]

            def need(key):
                while key not in fromPrev:
                    it = planPrev[got[0]]
                    got[0] += 1
                    fromPrev[(it[0], it[1], it[2])] = engine.recvStrip(comm, prevRank, it)
            engine.beginProvisional(stride, ntAll)
            for (slot, j) in sorted(enumerate(jobs), key=lambda sj: tkey(sj[1].col, sj[1].row)):
                if j.row > 0 and (j.col, j.row - 1) not in jobmap:
                    need(('b', j.col, j.row - 1))
                if j.col > 0 and (j.col - 1, j.row) not in jobmap:
                    need(('r', j.col - 1, j.row))
                engine.waitTile(j)
                (top, left) = _neighbours(j, fromPrev)
                t = j.row * ncolsT + j.col
                engine.stitchTileAt(j, top, left, _winOf(j.col, j.row), t, stride, slot)
                for it in sendOf.get((j.col, j.row), ()):
                    engine.sendStrip(comm, nextRank, it, j)
            while got[0] < len(planPrev):          # (every planned strip has a reader; be safe)
                need((planPrev[got[0]][0], planPrev[got[0]][1], planPrev[got[0]][2]))
            if hasattr(engine, 'drainStrips'):
                engine.drainStrips()               # strips in flight must land before anything is renumbered
            counts = engine.tileCounts(len(jobs))
            mine = [(j.row * ncolsT + j.col, int(k), int(r), int(j.maxLocal))
                    for (j, (k, r)) in zip(jobs, counts)]
        everyone = [x for part in comm.allgather_obj(mine) for x in part]
        K = numpy.zeros(ntAll, dtype=numpy.int64)
        R = numpy.zeros(ntAll, dtype=numpy.int64)
        hard = len(everyone) != ntAll
        for (t, k, r, mloc) in everyone:
            K[t] = k
            R[t] = r
            if mloc >= stride or k >= stride:
                hard = True
        if hard or int(K.sum()) > 0xFFFFFFFF:
            return None
        base = numpy.concatenate(([0], numpy.cumsum(K)[:-1])).astype(numpy.uint32)
        off = numpy.nonzero(K != R)[0]
        if len(off) == 0:
            if haveTiles:
                engine.renumber(stride, base)
            return int(K.sum())
        # Tile `bad` is the first (row-major) to hide ids it handed out from its trimmed window.  Up to
        # and including it the sequential run has maxSegId = sum of the earlier tiles' K at every step
        # (the induction of the safety test), so every decision taken so far -- bad's own recode
        # included -- stands and the provisional ids of tiles <= bad renumber to the final ones.  What
        # changes is where the NEXT tile starts: at base[bad] + R[bad], not + K[bad] (tiling.py:1029-1043:
        # maxSegId follows trimmed.max()).  The chain is redone from there only.
        bad = int(off[0])
        return ('partial', bad, base, int(base[bad]) + int(R[bad]), stride, fromPrev if haveTiles else {})

    def _resume(bad, base, mAfter, stride, fromPrevProv):
        """The sequential chain from tile bad + 1 on, after the tiles up to `bad` were kept."""
        maxSegId = 0
        if haveTiles:
            kept = [j for j in jobs if j.row * ncolsT + j.col <= bad]
            redo = [j for j in jobs if j.row * ncolsT + j.col > bad]
            ownsBad = t0 <= bad < t1
            # final ids for what is kept: output rows, the kept tiles' strips, and (on the rank that owns
            # `bad`) the previous rank's strips, which arrived with provisional ids
            # (a rank all of whose tiles are redone has nothing to renumber; one that redoes none never reads
            #  its strips again: only its output rows get their final ids)
            if kept:
                engine.renumberKept(stride, base, kept if redo else [],
                                    list(fromPrevProv.values()) if (ownsBad and redo) else [])
            if redo:
                if ownsBad:
                    (maxSegId, fromPrev) = (mAfter, fromPrevProv)
                else:
                    (maxSegId, fromPrev) = (0, {})
                    if prevRank is not None:
                        maxSegId, fromPrev = engine.recvBoundary(
                            comm, prevRank, boundaryPlan(tileInfo, shards, prevRank, overlapSize))
                engine.setMaxSegId(maxSegId)
                for j in redo:
                    engine.waitTile(j)
                    (top, left) = _neighbours(j, fromPrev)
                    engine.stitchTile(j, top, left, _winOf(j.col, j.row), simpleTileRecode)
                maxSegId = engine.getMaxSegId()
            elif ownsBad:
                maxSegId = mAfter
            if nextRank is not None and t1 - 1 >= bad:          # the next rank redoes all its tiles
                plan = boundaryPlan(tileInfo, shards, comm.rank, overlapSize)
                engine.sendBoundary(comm, nextRank, maxSegId,
                                    [(kind, jobmap[(c, r)], h, w) for (kind, c, r, h, w) in plan])
        vals = comm.allgather_obj(int(maxSegId))
        return vals[nonEmpty[-1]] if nonEmpty else 0

    chainRedone = 0
    if stitchMode == 'parallel':
        maxSegId = _parallel()
        if maxSegId is None:               # (a tile outgrew the provisional id range: everything again)
            stitchMode = 'parallel->sequential'
            chainRedone = ncolsT * tileInfo.nrows
            maxSegId = _sequential()
        elif isinstance(maxSegId, tuple):
            (_tag, bad, base, mAfter, stride, fromPrevProv) = maxSegId
            stitchMode = 'parallel->sequential'
            chainRedone = ncolsT * tileInfo.nrows - (bad + 1)
            maxSegId = _resume(bad, base, mAfter, stride, fromPrevProv)
    else:
        maxSegId = _sequential()
    _mark('tiles + stitch')
    hist = engine.histogram(maxSegId) if haveTiles else numpy.zeros(maxSegId + 1, numpy.int64)
    hist = comm.allreduce_sum_i64(numpy.asarray(hist, dtype=numpy.int64)).astype(numpy.uint32)
    hist[0] = 0
    _mark('histogram')
    engine.finish()
    _mark('finish')
    if os.environ.get('SHEPSEG_IO_TIMING'):
        tm = getattr(engine, 'timings', None)
        sys.stderr.write('  [dist rank %d] %s%s\n' % (comm.rank, ', '.join(_marks),
                                                      ' | worker timers %s' % tm.makeSummaryDict() if tm is not None else ''))

    res = DistResult()
    res.maxSegId = int(maxSegId)
    res.hist = hist
    res.kmeans = kmeansObj
    res.maxSpectralDiff = msd
    res.stitchMode = stitchMode
    res.chainStepsRedone = chainRedone      # parallel form: tiles whose chain step ran a second time
    res.subsamplePcnt = subsamplePcnt
    res.rowRange = (r0, r1)
    res.tileRange = (t0, t1)
    res.outRows = (outLo, outHi)
    res.numTileRows, res.numTileCols = tileInfo.nrows, tileInfo.ncols
    res.hasEmptySegments = bool((hist[1:] == 0).any())
    return res


def idRange(rank, world, maxSegId):
    """This rank's share [lo, hi) of the id space 0..maxSegId for the reduction of straddling segments
    ('GPU g owns ids in [g S / N, (g + 1) S / N)')."""
    ns = int(maxSegId) + 1
    return (rank * ns) // world, ((rank + 1) * ns) // world


def calcPerSegmentStatsDistributed(engine, comm, hist, imgbandnum, statsSelection,
                                   missingStatsValue=-9999, imgNullVal=None, info=None):
    """Per-segment statistics of one image band against the stitched label raster that
    runDistributed left sharded by rows over the ranks (reference tilingstats.py:85-216; SURVEY
    8e).  ``hist`` is the global histogram of the labels (DistResult.hist), which plays the part
    of the reference's segSize: a segment whose local pixel count equals hist[id] is complete on
    this rank and its statistics are final (checkSegComplete, tilingstats.py:518-553).  The
    pixels of the segments that straddle a rank boundary are packed as (id, value) pairs and
    all-gathered; every rank then reduces the straddlers whose id lies in ITS share of the id space
    (idRange) with the same kernel.  Finished rows are disjoint between ranks, so one integer
    all-reduce assembles the columns.
      With a device engine under RCCL nothing of this crosses the host (shp_dstats_local_dev ->
    ncclAllGather of the pairs -> shp_dstats_merge_dev -> ncclAllReduce of the column block); other
    transports carry the same arrays as raw bytes.  ``info`` (a dict, optional) receives 'straddlers'
    (segments) and 'straddler_pixels' of the whole job and 'path'.
    Returns (intcols int64 (nInt, maxSegId+1), floatcols float32 (nFloat, maxSegId+1),
    statsSelection_fast) on every rank -- bit-identical to the single-GPU result."""
    from . import tilingstats
    (fast, nInt, nFloat) = tilingstats.makeFastStatsSelection(
        list(range(len(statsSelection))), statsSelection)
    if getattr(comm, 'onDevice', False) and hasattr(engine, 'statsOnDevice'):
        (ic, fc, nStrad, nPix) = engine.statsOnDevice(comm, hist, imgbandnum, fast, nInt, nFloat,
                                                      missingStatsValue, imgNullVal)
        if info is not None:
            info.update(straddlers=nStrad, straddler_pixels=nPix, path='device')
        return ic, fc, fast
    hist = numpy.asarray(hist).astype(numpy.int64)
    S = len(hist) - 1
    lh = numpy.asarray(engine.histogram(S)).astype(numpy.int64)
    lh[0] = 0
    complete = (lh == hist) & (lh > 0)
    strad = (lh > 0) & (lh < hist)
    (ic, fc) = engine.localStats(imgbandnum, S, fast, nInt, nFloat, missingStatsValue, imgNullVal)
    keep = complete.copy()
    if comm.rank == 0:
        keep |= (hist == 0)                 # ids nobody holds (and row 0): "missing" rows, once
    ic[:, ~keep] = 0
    fc[:, ~keep] = 0
    pairs = engine.gatherFlagged(imgbandnum, S, strad.astype(numpy.uint8), int(lh[strad].sum()))
    allPairs = comm.allgather_arrays([pairs[0], pairs[1]])
    (lo, hi) = idRange(comm.rank, comm.world, S)
    segs = numpy.concatenate([p[0] for p in allPairs])
    vals = numpy.concatenate([p[1] for p in allPairs])
    mine = (segs >= lo) & (segs < hi)
    if mine.any():
        (ids, compact) = numpy.unique(segs[mine], return_inverse=True)
        (ic2, fc2) = engine.statsOfPairs((compact + 1).astype(numpy.uint32), vals[mine], len(ids), fast,
                                         nInt, nFloat, missingStatsValue, imgNullVal)
        ic[:, ids] = ic2[:, 1:]
        fc[:, ids] = fc2[:, 1:]
    if info is not None:
        # (a straddler is counted by every rank that holds a part of it: count the ids, once each)
        info.update(straddlers=int(len(numpy.unique(segs))), straddler_pixels=int(len(segs)), path='host')
    if comm.world > 1:
        ic = comm.allreduce_sum_i64(ic.reshape(-1)).reshape(nInt, S + 1)
        fbits = comm.allreduce_sum_i64(fc.view(numpy.int32).reshape(-1).astype(numpy.int64))
        fc = fbits.astype(numpy.int32).view(numpy.float32).reshape(nFloat, S + 1)
    return ic, fc, fast


def deviceStats(c, comm, d_seg, d_band, dtypeCode, nRows, nCols, hist, fast, nInt, nFloat, missing, imgNullVal,
                fetch=True):
    """The device-resident data path of calcPerSegmentStatsDistributed for ONE rank: label rows d_seg
    (nRows x nCols uint32) and band rows d_band in the HBM of context ``c``; comm: allgather_obj (control
    data only), allgather_dev, allreduce_dev_i64.  ``hist``: the global histogram, a numpy array or
    ('dev', address, length) when it is in device memory already.  fetch=False: the assembled columns are not
    copied to the host (ic = fc = None).  Returns (ic, fc, straddling segments, their pixels)."""
    L = c._L
    if isinstance(hist, tuple):
        (d_hist, ns, ownHist) = (ctypes.c_void_p(hist[1]), int(hist[2]), False)
    else:
        h32 = numpy.ascontiguousarray(hist, dtype=numpy.uint32)
        ns = len(h32)
        d_hist = tiling._devAlloc(c, ns * 4)
        c.check(L.shp_dev_upload(c.handle, d_hist, _lib.ptr(h32), ns * 4))
        ownHist = True
    S = ns - 1
    colWords = ((nInt * 8 + nFloat * 4) * ns + 7) // 8
    # (blocks from / back to the driver's cache of device scratch blocks: a 1.2-GB hipMalloc per call otherwise)
    d_cols = tiling._devAlloc(c, colWords * 8)
    toFree = [(d_cols, colWords * 8)] + ([(d_hist, ns * 4)] if ownHist else [])
    try:
        c.check(L.shp_dev_memset(c.handle, ctypes.c_void_p(d_cols.value + (colWords - 1) * 8), 0, 8))
        (pSeg, pVal) = (ctypes.c_void_p(), ctypes.c_void_p())
        (nPairs, nStrad) = (ctypes.c_int64(0), ctypes.c_int64(0))
        hasNull = int(imgNullVal is not None)
        nullVal = 0 if imgNullVal is None else int(imgNullVal)
        c.check(L.shp_dstats_local_dev(c.handle, ctypes.c_void_p(d_seg), ctypes.c_void_p(d_band), dtypeCode, nRows, nCols,
                                       S, hasNull, nullVal, _lib.ptr(fast), fast.shape[0], int(missing), d_hist,
                                       int(comm.rank == 0), d_cols, ctypes.byref(pSeg), ctypes.byref(pVal),
                                       ctypes.byref(nPairs), ctypes.byref(nStrad)))
        counts = [int(x) for x in comm.allgather_obj(int(nPairs.value))]          # control data
        slot = max(counts)
        (merged, nIds) = (ctypes.c_int64(0), ctypes.c_int64(0))
        if slot > 0:
            bufs = []
            for sz in (slot * 4, slot * 8, comm.world * slot * 4, comm.world * slot * 8):
                p = tiling._devAlloc(c, sz)
                bufs.append(p)
                toFree.append((p, sz))
            (d_sendS, d_sendV, d_allS, d_allV) = bufs
            if nPairs.value:
                c.check(L.shp_dev_copy(c.handle, d_sendS, pSeg, nPairs.value * 4))
                c.check(L.shp_dev_copy(c.handle, d_sendV, pVal, nPairs.value * 8))
            comm.allgather_dev(d_sendS.value, d_allS.value, slot * 4)
            comm.allgather_dev(d_sendV.value, d_allV.value, slot * 8)
            (lo, hi) = idRange(comm.rank, comm.world, S)
            cnts = numpy.array(counts, dtype=numpy.uint32)
            c.check(L.shp_dstats_merge_dev(c.handle, d_allS, d_allV, slot, comm.world, _lib.ptr(cnts), dtypeCode, S,
                                           hasNull, nullVal, _lib.ptr(fast), fast.shape[0], int(missing), lo, hi, d_cols,
                                           ctypes.byref(merged), ctypes.byref(nIds)))
        if comm.world > 1:
            comm.allreduce_dev_i64(d_cols.value, colWords)
        (ic, fc) = (None, None)
        if fetch:
            ic = numpy.empty((nInt, ns), dtype=numpy.int64)
            fc = numpy.empty((nFloat, ns), dtype=numpy.float32)
        if fetch and nInt:
            c.check(L.shp_dev_download(c.handle, _lib.ptr(ic), d_cols, ic.nbytes))
        if fetch and nFloat:
            c.check(L.shp_dev_download(c.handle, _lib.ptr(fc), ctypes.c_void_p(d_cols.value + nInt * 8 * ns), fc.nbytes))
    finally:
        for (p, sz) in toFree:
            tiling._devRelease(c, p, sz)
    # the job's figures: the ranks' id shares partition the straddlers, the ranks' rows their pixels
    tot = comm.allgather_obj((int(nIds.value), int(nPairs.value)))
    return ic, fc, int(sum(t[0] for t in tot)), int(sum(t[1] for t in tot))


# ------------------------------------------------------------------------------------------
# HIP engine: this rank's GPU
# ------------------------------------------------------------------------------------------
class HipEngine(object):
    """Holds rows [yLo, yHi) of the raster in HBM (a DeviceRaster created by `makeSlice`),
    segments this rank's tiles with pooled worker contexts and stitches them on the device.
    Boundary strips go from this rank's strip block straight into ncclSend (RcclComm), or through
    host memory when the communicator is not on the device (SocketComm: ranks sharing a GPU)."""

    def __init__(self, makeSlice, numWorkers=16, keepOutput=False):
        self.makeSlice = makeSlice          # f(yLo, yHi) -> DeviceRaster of those rows
        self.numWorkers = numWorkers
        self.keepOutput = keepOutput
        self.ras = None
        self.timings = tiling.Timers()
        self._sliceKey = None

    def setup(self, tileInfo, jobs, total, yLo, yHi, outLo, outHi, nCols, overlapSize):
        self.c = _lib.chain_ctx()
        self.L = self.c._L
        self.tileInfo, self.jobs = tileInfo, jobs
        (self.yLo, self.yHi, self.outLo, self.outHi) = (yLo, yHi, outLo, outHi)
        self.nCols, self.overlap = nCols, overlapSize
        if self._sliceKey != (yLo, yHi):
            if self.ras is not None:
                self.ras.free()
            self.ras = self.makeSlice(yLo, yHi) if yHi > yLo else None
            self._sliceKey = (yLo, yHi)
        self.nbTiles = max(total, 1) * 4
        self.nbOut = max((outHi - outLo) * nCols, 1) * 4
        self.d_tiles = tiling._devAlloc(self.c, self.nbTiles)
        self.d_out = tiling._devAlloc(self.c, self.nbOut)
        # other ranks' tiles share these rows: what this rank does not write must read as null
        self.c.check(self.L.shp_dev_memset(self.c.handle, self.d_out, 0, self.nbOut))
        self.d_scal = tiling._devAlloc(self.c, 256)
        self.c.check(self.L.shp_dev_memset(self.c.handle, self.d_scal, 0, 256))
        self.nbStrips = max(tiling.layoutStrips(jobs, overlapSize), 1) * 4
        self.d_strips = tiling._devAlloc(self.c, self.nbStrips)
        self.arena = tiling._MetaArena(self.c, 16 * (total // 8 + 1024))
        self.simple = False
        self.threads, self.forceExit = [], None
        self.recvBufs = []
        self.recvDev = []

    def subsample(self, rowsGlobal, cols):
        nb = self.ras.shape[0] if self.ras is not None else 0
        if self.ras is None or len(rowsGlobal) == 0:
            return numpy.zeros((nb, 0, len(cols)), dtype=numpy.uint16)
        ry = (rowsGlobal - self.yLo).astype(numpy.uint32)
        out = numpy.empty((nb, len(ry), len(cols)), dtype=self.ras.dtype)
        self.c.check(self.L.shp_dev_subsample(
            self.c.handle, ctypes.c_void_p(self.ras.ptr), _lib.SHP_DTYPES[self.ras.dtype], nb,
            self.ras.shape[1], self.ras.shape[2], _lib.ptr(ry), len(ry),
            _lib.ptr(numpy.ascontiguousarray(cols, dtype=numpy.uint32)), len(cols), _lib.ptr(out)))
        return out

    def fit(self, img, numClusters, imgNullVal, fixedKMeansInit):
        with self.timings.interval('spectralclusters'):
            return shepseg.fitSpectralClusters(img, numClusters, 100, imgNullVal, fixedKMeansInit)

    def fitSharded(self, img, numClusters, imgNullVal, comm):
        with self.timings.interval('spectralclusters'):
            return shepseg.fitSpectralClusters(img, numClusters, 100, imgNullVal, True, _commHandle=comm.h)

    def startSegmentation(self, centres, msd, imgNullVal, fourConnected, minSegmentSize):
        if not self.jobs:
            return
        self.threads, self.forceExit = tiling.startSegmentationWorkers(
            self.ras, self.jobs, self.d_tiles, centres, msd, imgNullVal, fourConnected,
            minSegmentSize, self.numWorkers, self.timings, yOrigin=self.yLo,
            stitchPrep=(self.tileInfo, self.overlap, self.arena, self.simple))

    def waitTile(self, j):
        tiling.waitForTile(j, self.jobs, self.threads, self.forceExit, 600)

    def setMaxSegId(self, v):
        a = numpy.array([v], dtype=numpy.uint32)
        self.c.check(self.L.shp_dev_upload(self.c.handle, self.d_scal, _lib.ptr(a), 4))

    def getMaxSegId(self):
        a = numpy.zeros(1, dtype=numpy.uint32)
        self.c.check(self.L.shp_sync(self.c.handle))
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(a), self.d_scal, 4))
        return int(a[0])

    # strips are (device pointer, row pitch in elements)
    def bottomStripOf(self, a):      # dense recoded strip written by the chain step of tile a
        return (self.d_strips.value + 4 * a.bottomOff, a.xsize)

    def rightStripOf(self, a):
        return (self.d_strips.value + 4 * a.rightOff, min(self.overlap, a.xsize))

    def stitchTile(self, j, top, left, win, simple):
        (t, b, l, r, xout, yout) = win
        with self.timings.interval('stitchtiles'):
            # every tile's strips are written: a later rank may need the last row's bottom strips
            self.c.check(self.L.shp_stitch_chain_dev(
                self.c.handle, ctypes.c_void_p(self.d_tiles.value + 4 * j.offset), j.ysize, j.xsize,
                self.overlap, ctypes.c_void_p(top[0]) if top else None, top[1] if top else 0,
                ctypes.c_void_p(left[0]) if left else None, left[1] if left else 0, j.maxLocal,
                int(bool(simple)), self.d_scal, t, b, l, r, ctypes.c_void_p(j.meta),
                ctypes.c_void_p(self.d_strips.value + 4 * j.rightOff),
                ctypes.c_void_p(self.d_strips.value + 4 * j.bottomOff), self.d_out, self.nCols, xout,
                yout - self.outLo, j.crossPx[0], j.crossPx[1]))

    # ---- parallel stitch: provisional bases, per-tile counts, eager strips ----
    def beginProvisional(self, stride, ntAll):
        bases = (numpy.arange(ntAll, dtype=numpy.uint64) * stride).astype(numpy.uint32)
        self.nbBases = (ntAll + 2 * len(self.jobs) + 16) * 4
        self.d_bases = tiling._devAlloc(self.c, self.nbBases)
        self.c.check(self.L.shp_dev_memset(self.c.handle, self.d_bases, 0, self.nbBases))
        self.c.check(self.L.shp_dev_upload(self.c.handle, self.d_bases, _lib.ptr(bases), ntAll * 4))
        self.ntAll = ntAll

    def stitchTileAt(self, j, top, left, win, t, stride, slot):
        """The chain step of tile t with its provisional base (a device word of its own, so nothing
        is uploaded or read back per tile), then its two counts into slot `slot`."""
        (tt, b, l, r, xout, yout) = win
        with self.timings.interval('stitchtiles'):
            self.c.check(self.L.shp_stitch_chain_dev(
                self.c.handle, ctypes.c_void_p(self.d_tiles.value + 4 * j.offset), j.ysize, j.xsize,
                self.overlap, ctypes.c_void_p(top[0]) if top else None, top[1] if top else 0,
                ctypes.c_void_p(left[0]) if left else None, left[1] if left else 0, j.maxLocal,
                0, ctypes.c_void_p(self.d_bases.value + 4 * t), tt, b, l, r, ctypes.c_void_p(j.meta),
                ctypes.c_void_p(self.d_strips.value + 4 * j.rightOff),
                ctypes.c_void_p(self.d_strips.value + 4 * j.bottomOff), self.d_out, self.nCols, xout,
                yout - self.outLo, j.crossPx[0], j.crossPx[1]))
            self.c.check(self.L.shp_stitch_counts_dev(
                self.c.handle, ctypes.c_void_p(j.meta), j.maxLocal, (t * stride) & 0xFFFFFFFF,
                ctypes.c_void_p(self.d_bases.value + 4 * (self.ntAll + 2 * slot))))

    def tileCounts(self, n):
        a = numpy.zeros(2 * max(n, 1), dtype=numpy.uint32)
        self.c.check(self.L.shp_sync(self.c.handle))
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(a),
                                             ctypes.c_void_p(self.d_bases.value + 4 * self.ntAll), 8 * max(n, 1)))
        tiling._devRelease(self.c, self.d_bases, self.nbBases)
        self.d_bases = None
        return a[:2 * n].reshape(n, 2)

    def renumber(self, stride, base):
        self.c.check(self.L.shp_sync(self.c.handle))
        base = numpy.ascontiguousarray(base, dtype=numpy.uint32)
        self.c.check(self.L.shp_renumber_dev(self.c.handle, self.d_out, (self.outHi - self.outLo) * self.nCols,
                                             int(stride), _lib.ptr(base), len(base)))

    def renumberKept(self, stride, base, keptJobs, recvStrips):
        """provisional -> final ids in the output rows, in the recoded strips of the tiles that are kept
        and in EXACTLY the strips handed in (received from the previous rank with provisional ids: the
        partial redo of the parallel stitch) -- a buffer received with final ids must not be touched again."""
        self.renumber(stride, base)
        base = numpy.ascontiguousarray(base, dtype=numpy.uint32)

        def stripWords(j):
            return j.ysize * min(self.overlap, j.xsize) + min(self.overlap, j.ysize) * j.xsize      # right | bottom
        # the kept tiles' strips: runs of jobs whose strips are adjacent in the strip block go in one call
        runs = []
        for j in sorted(keptJobs, key=lambda q: q.rightOff):
            n = stripWords(j)
            if runs and runs[-1][0] + runs[-1][1] == j.rightOff:
                runs[-1][1] += n
            else:
                runs.append([j.rightOff, n])
        for (o, n) in runs:
            self.c.check(self.L.shp_renumber_dev(self.c.handle, ctypes.c_void_p(self.d_strips.value + 4 * o),
                                                 n, int(stride), _lib.ptr(base), len(base)))
        sizes = {d.value: nbytes for (d, nbytes) in self.recvDev}
        for (ptr, _w) in recvStrips:
            self.c.check(self.L.shp_renumber_dev(self.c.handle, ctypes.c_void_p(ptr), sizes[ptr] // 4, int(stride),
                                                 _lib.ptr(base), len(base)))

    def sendStrip(self, comm, dst, item, a):
        (kind, _c, _r, h, w) = item
        if hasattr(comm, 'isend_dev'):
            # asynchronous: the send waits ON THE DEVICE for the chain step that writes the strip, the chain
            # (this thread and its stream) goes on with the next tile
            (ptr, _pitch) = self.bottomStripOf(a) if kind == 'b' else self.rightStripOf(a)
            comm.isend_dev(ptr, h * w * 4, dst, self.c)
            self.asyncComm = comm
            return
        self.c.check(self.L.shp_sync(self.c.handle))          # the chain step that wrote it is done
        self._sendOne(comm, dst, kind, a, h * w)

    def recvStrip(self, comm, src, item):
        (kind, _c, _r, h, w) = item
        if hasattr(comm, 'irecv_dev'):
            d = tiling._devAlloc(self.c, h * w * 4)
            self.recvDev.append((d, h * w * 4))
            comm.irecv_dev(d.value, h * w * 4, src, self.c)     # the chain's stream waits for the data, not the host
            self.asyncComm = comm
            return (d.value, w)
        return (self._recvOne(comm, src, h * w), w)

    def drainStrips(self):
        """every strip sent or received asynchronously so far has arrived (host wait)"""
        c = getattr(self, 'asyncComm', None)
        if c is not None:
            c.drain()
            self.asyncComm = None

    def _sendOne(self, comm, dst, kind, a, n):
        (ptr, _pitch) = self.bottomStripOf(a) if kind == 'b' else self.rightStripOf(a)
        if comm.onDevice:        # device memory straight into RCCL
            comm.send_dev(ptr, n * 4, dst)
        else:                    # ranks without a device transport: stage through host memory
            buf = numpy.empty(n, dtype=numpy.uint32)
            self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(buf), ctypes.c_void_p(ptr), n * 4))
            comm.send_bytes(buf, dst)

    def _recvOne(self, comm, src, n):
        d = tiling._devAlloc(self.c, n * 4)
        self.recvDev.append((d, n * 4))
        if comm.onDevice:
            comm.recv_dev(d.value, n * 4, src)
        else:
            buf = numpy.frombuffer(comm.recv_bytes(src), dtype=numpy.uint32)
            self.c.check(self.L.shp_dev_upload(self.c.handle, d, _lib.ptr(numpy.ascontiguousarray(buf)), n * 4))
        return d.value

    def sendBoundary(self, comm, dst, maxSegId, items):
        """items: (kind, job, h, w) from boundaryPlan; strips are dense h x w blocks."""
        self.c.check(self.L.shp_sync(self.c.handle))
        comm.send_obj(int(maxSegId), dst)
        for (kind, a, h, w) in items:
            self._sendOne(comm, dst, kind, a, h * w)

    def recvBoundary(self, comm, src, plan):
        maxSegId = int(comm.recv_obj(src))
        strips = {}
        for (kind, col, row, h, w) in plan:
            strips[(kind, col, row)] = (self._recvOne(comm, src, h * w), w)
        return maxSegId, strips

    def histogram(self, maxSegId):
        hist = numpy.zeros(maxSegId + 1, dtype=numpy.uint32)
        self.c.check(self.L.shp_histogram_dev(self.c.handle, self.d_out,
                                              (self.outHi - self.outLo) * self.nCols, self.nCols, maxSegId,
                                              _lib.ptr(hist)))
        return hist

    def _bandPtr(self, imgbandnum):
        """Device address of this rank's OUTPUT rows of one image band (1-based band number)."""
        (nb, rows, cols) = self.ras.shape
        isz = numpy.dtype(self.ras.dtype).itemsize
        return self.ras.ptr + (((imgbandnum - 1) * rows + (self.outLo - self.yLo)) * cols) * isz

    def localStats(self, imgbandnum, S, fast, nInt, nFloat, missing, imgNullVal):
        n = (self.outHi - self.outLo) * self.nCols
        ic = numpy.zeros((nInt, S + 1), dtype=numpy.int64)
        fc = numpy.zeros((nFloat, S + 1), dtype=numpy.float32)
        if n > 0:
            self.c.check(self.L.shp_segstats_dev(
                self.c.handle, self._lastOut, ctypes.c_void_p(self._bandPtr(imgbandnum)),
                _lib.SHP_DTYPES[self.ras.dtype], n, S, int(imgNullVal is not None),
                0 if imgNullVal is None else int(imgNullVal), _lib.ptr(fast), fast.shape[0],
                int(missing), _lib.ptr(ic), _lib.ptr(fc)))
        return ic, fc

    def gatherFlagged(self, imgbandnum, S, flags, count):
        n = (self.outHi - self.outLo) * self.nCols
        segs = numpy.empty(max(count, 1), dtype=numpy.uint32)
        vals = numpy.empty(max(count, 1), dtype=numpy.int64)
        got = ctypes.c_int64(0)
        if n > 0 and count > 0:
            self.c.check(self.L.shp_gather_flagged_dev(
                self.c.handle, self._lastOut, ctypes.c_void_p(self._bandPtr(imgbandnum)),
                _lib.SHP_DTYPES[self.ras.dtype], n, S, _lib.ptr(flags), count, _lib.ptr(segs),
                _lib.ptr(vals), ctypes.byref(got)))
            if got.value != count:
                raise tiling.PyShepSegTilingError(
                    "straddling-segment gather found %d pixels, histogram says %d" % (got.value, count))
        return segs[:count], vals[:count]

    def statsOfPairs(self, segs, vals, K, fast, nInt, nFloat, missing, imgNullVal):
        """Statistics of a list of (compact id 1..K, value) pairs: the same kernel on a 1 x M raster."""
        band = vals.astype(self.ras.dtype if self.ras is not None else numpy.uint16)
        ic = numpy.zeros((nInt, K + 1), dtype=numpy.int64)
        fc = numpy.zeros((nFloat, K + 1), dtype=numpy.float32)
        self.c.check(self.L.shp_segstats(
            self.c.handle, _lib.ptr(segs), _lib.ptr(band), _lib.SHP_DTYPES[band.dtype], len(segs), K,
            int(imgNullVal is not None), 0 if imgNullVal is None else int(imgNullVal),
            _lib.ptr(fast), fast.shape[0], int(missing), _lib.ptr(ic), _lib.ptr(fc)))
        return ic, fc

    def statsOnDevice(self, comm, hist, imgbandnum, fast, nInt, nFloat, missing, imgNullVal):
        """calcPerSegmentStatsDistributed's device path for this rank's output rows (deviceStats)."""
        return deviceStats(self.c, comm, self._lastOut.value if hasattr(self._lastOut, 'value') else int(self._lastOut),
                           self._bandPtr(imgbandnum), _lib.SHP_DTYPES[self.ras.dtype], self.outHi - self.outLo,
                           self.nCols, hist, fast, nInt, nFloat, missing, imgNullVal)

    def localOutput(self):
        out = numpy.empty((self.outHi - self.outLo, self.nCols), dtype=numpy.uint32)
        self.c.check(self.L.shp_dev_download(self.c.handle, _lib.ptr(out), self._lastOut, out.nbytes))
        return out

    def finish(self):
        for t in self.threads:
            t.join()
        self.drainStrips()
        self.c.check(self.L.shp_sync(self.c.handle))
        self.recvBufs = []
        for (d, nbytes) in self.recvDev:
            tiling._devRelease(self.c, d, nbytes)
        self.recvDev = []
        tiling._devRelease(self.c, self.d_tiles, self.nbTiles)
        tiling._devRelease(self.c, self.d_scal, 256)
        tiling._devRelease(self.c, self.d_strips, self.nbStrips)
        self.arena.release()
        if self.keepOutput:
            self._lastOut = self.d_out               # caller must releaseOutput()
        else:
            tiling._devRelease(self.c, self.d_out, self.nbOut)

    def releaseOutput(self):
        tiling._devRelease(self.c, self.d_out, self.nbOut)


# ------------------------------------------------------------------------------------------
# bench.py entry for --gpus N > 1
# ------------------------------------------------------------------------------------------
def bench_stats_main(args, rank, world, local_rank):
    """One rank of `bench.py --workload c5 --gpus N`: the label raster of 4 x 8-pixel blocks (50 M segments at
    40000^2) and one uint16 band, sharded by rows at boundaries that CUT blocks (every shard boundary makes
    nCols / 8 straddling segments); a step = calcPerSegmentStatsDistributed's device path (deviceStats), the
    assembled columns copied to the host on rank 0."""
    from . import comm as _comm
    from . import tilingstats
    comm = _comm.fromEnvironment()
    c = _lib.ctx()
    L = c._L
    dcomm = comm if getattr(comm, 'onDevice', False) else _comm.HostStagedDev(comm, c)
    (N, BH, BW) = (args.size, 4, 8)
    # shard boundaries two rows into a block row
    cuts = [0] + [min(N, ((N * (r + 1)) // world) // BH * BH + 2) for r in range(world - 1)] + [N]
    (y0, y1) = (cuts[rank], cuts[rank + 1])
    ras = tiling.DeviceRaster.synth(getattr(args, 'seed', 11), 1, y1 - y0, N, y0=y0, x0=0)
    d_full = ctypes.c_void_p()
    c.check(L.shp_dev_alloc(c.handle, N * N * 4, ctypes.byref(d_full)))
    S = ctypes.c_uint32(0)
    c.check(L.shp_dev_block_labels(c.handle, N, N, BH, BW, d_full, ctypes.byref(S)))
    S = S.value
    d_seg = d_full.value + y0 * N * 4                       # this rank's rows of the label raster
    h = numpy.full(S + 1, BH * BW, dtype=numpy.uint32)      # every block whole: the RAT's Histogram column
    h[0] = 0
    d_hist = ctypes.c_void_p()
    c.check(L.shp_dev_alloc(c.handle, h.nbytes, ctypes.byref(d_hist)))
    c.check(L.shp_dev_upload(c.handle, d_hist, _lib.ptr(h), h.nbytes))
    sel = [('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'), ('n', 'pixcount')]
    (fast, nInt, nFloat) = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)

    def step():
        return deviceStats(c, dcomm, d_seg, ras.ptr, 2, y1 - y0, N, ('dev', d_hist.value, S + 1), fast, nInt, nFloat,
                           -9999, None, fetch=(rank == 0))

    for _ in range(args.warmup):
        res = step()
    c.check(L.shp_sync(c.handle))
    comm.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        res = step()
    c.check(L.shp_sync(c.handle))
    comm.barrier()
    dt = comm.max_f64((time.time() - t0) / max(args.steps, 1))
    rcclRanks = comm.count() if hasattr(comm, 'count') else None
    if rank == 0:
        (ic, fc, nStrad, nPix) = res
        assert int(ic[fast[3, 3]].sum()) == N * N and (ic[fast[3, 3]][1:] == BH * BW).all()
        npix = N * N
        alg = 6 * npix + 4 * len(sel) * (S + 1)            # 6 B/px + 4 * nCols B/segment
        out = {
            "metric": "Mpixels/sec, tilingstats per-segment mean/stddev/median/pixcount",
            "value": round(npix / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "C5: %dx%d label raster of %d x %d-pixel blocks (%d segments) + one uint16 synthimg v1 "
                                   "band, rows sharded over %d GPUs at boundaries that cut blocks; 4 result columns "
                                   "assembled by one all-reduce and copied to the host on rank 0" % (N, N, BH, BW, S, world),
                       "segments": S, "straddlers": int(nStrad), "straddler_pixels": int(nPix),
                       "rows_per_rank": [cuts[r + 1] - cuts[r] for r in range(world)],
                       "transport": getattr(dcomm, 'transport', type(dcomm).__name__), "rccl_nranks": rcclRanks,
                       "parallelism": "label rows sharded over %d ranks, one process per GPU; straddling segments' pixels "
                                      "all-gathered as packed device buffers and reduced by id share; columns all-reduced" % world},
            "roofline": {"bound": "hbm", "kernel": "whole call (local statistics + exchange + all-reduce + download)",
                         "achieved": round(alg / dt / 1e9, 3), "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": round(alg / dt / 1e9 / (8000.0 * world), 6), "traffic": None},
        }
        print(json.dumps(out), flush=True)
    comm.barrier()
    c.check(L.shp_dev_free(c.handle, d_hist))
    c.check(L.shp_dev_free(c.handle, d_full))
    ras.free()
    comm.close()


def bench_main(args, rank, world, local_rank):
    """One rank of the multi-GPU benchmark: this rank's rows of the synthetic image are generated
    in its own HBM (synthimg is position-deterministic); a step = runDistributed."""
    from . import comm as _comm
    comm = _comm.fromEnvironment()
    nb = args.bands

    def makeSlice(yLo, yHi):
        return tiling.DeviceRaster.synth(getattr(args, 'seed', 11), nb, yHi - yLo, args.size, y0=yLo, x0=0)
    engine = HipEngine(makeSlice, numWorkers=args.workers)
    sync = _lib.ctx()

    def step():
        return runDistributed(engine, comm, args.size, args.size, args.tile, args.overlap,
                              minSegmentSize=50, numClusters=60, fixedKMeansInit=True)

    for _ in range(args.warmup):
        r = step()
    sync.check(sync._L.shp_sync(sync.handle))
    comm.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        r = step()
    sync.check(sync._L.shp_sync(sync.handle))
    comm.barrier()
    dt = comm.max_f64((time.time() - t0) / max(args.steps, 1))
    tilesPerRank = [int(x) for x in comm.allgather_obj(int(r.tileRange[1] - r.tileRange[0]))]
    rcclRanks = comm.count() if hasattr(comm, 'count') else None      # what RCCL itself says (ncclCommCount)
    fitShardedHere = (world > 1 and getattr(comm, 'onDevice', False) and hasattr(comm, 'h') and
                      hasattr(engine, 'fitSharded') and os.environ.get('SHEPSEG_FIT_SHARDED', '1') != '0')
    if rank == 0:
        npix = args.size * args.size
        value = npix / dt / 1e6
        out = {
            "metric": "Mpixels/sec segmented, %d-band 40k x 40k tiled" % nb,
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "%s: tiled %dx%d, %d-band uint16 synthimg v1, tileSize=%d, "
                                   "overlap=%d, k=60, minSegmentSize=50, fixedKMeansInit, tile rows "
                                   "sharded over %d GPUs, image + labels resident in HBM"
                                   % (getattr(args, 'workload', 'c3').upper(), args.size, args.size, nb,
                                      args.tile, args.overlap, world),
                       "tiles": r.numTileRows * r.numTileCols, "workers": args.workers,
                       "max_seg_id": int(r.maxSegId),
                       "stitch": r.stitchMode, "chain_steps_redone": int(r.chainStepsRedone),
                       "tiles_per_rank": tilesPerRank,
                       "transport": getattr(comm, 'transport', type(comm).__name__),
                       "rccl_nranks": rcclRanks,
                       "parallelism": "tiles sharded by area over %d ranks, one process per GPU; the k-means fit %s; "
                                      "overlap strips point to point, histogram all-reduced" % (
                                          world, "with its E-step sharded by sample rows (labels all-gathered per iteration)"
                                          if fitShardedHere else "on rank 0")},
            # (cpu_baseline is a one-GPU figure: `python bench.py` prints it; BASELINE.md: the reference's numba
            #  path does ~1.3 Mpixels/s per core)
            "reference_numba_mpx_per_core": 1.3,
            "roofline": {"bound": "hbm", "kernel": "whole path", "achieved": round(
                value * 1e6 * (2 * nb + 4) / 1e9, 3), "peak": 8000.0 * world, "unit": "GB/s",
                "frac": round(value * 1e6 * (2 * nb + 4) / 1e9 / (8000.0 * world), 6),
                "traffic": None},
        }
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()
