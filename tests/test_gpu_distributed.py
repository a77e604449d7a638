"""GPU: the multi-GPU driver with the HIP engine.  The test box has one GPU, so two ranks share it
and exchange the boundary strips over the socket transport (host-staged); at world size 1 the
RCCL communicator itself runs (ncclCommInitRank, its collectives and a self send/recv are
exercised separately).  Result must equal the single-process tiled run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_ranks(world, argv, tmp_path, timeout=900, extra=None):
    procs = []
    import secrets
    nonce = secrets.token_hex(8)
    for r in range(world):
        env = dict(os.environ, SHEPSEG_LAUNCH_NONCE=nonce, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT='0', SHEPSEG_COMM_DIR=str(tmp_path / 'comm'))
        env.update(extra or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for (p, (_o, e)) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]


def test_rccl_comm_world_one(tmp_path):
    """the RCCL binding itself: communicator creation from a unique id, every collective the
    driver uses, in a fresh process (one rank: more ranks need more GPUs)"""
    code = (
        "import sys, ctypes, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from pyshepseg_amd import comm as C, _lib\n"
        "c = C.RcclComm()\n"
        "assert (c.rank, c.world) == (0, 1)\n"
        "st = c._staging(4096)\n"
        "a = np.arange(16, dtype=np.int64)\n"
        "c.c.check(c.L.shp_dev_upload(c.c.handle, st, _lib.ptr(a), a.nbytes))\n"
        "c.c.check(c.L.shp_comm_allreduce(c.h, st, 16, 0))\n"
        "c.c.check(c.L.shp_comm_bcast(c.h, st, 128, 0))\n"
        "out = ctypes.c_void_p(st.value + 2048)\n"
        "c.c.check(c.L.shp_comm_allgather(c.h, st, out, 128))\n"
        "b = np.zeros(16, dtype=np.int64)\n"
        "c.c.check(c.L.shp_dev_download(c.c.handle, _lib.ptr(b), out, 128))\n"
        "assert np.array_equal(a, b)\n"
        "assert c.count() == 1\n"                                  # ncclCommCount
        # the asynchronous strip exchange on the communicator's own stream, ordered against a context's
        # stream on the device: this rank sends to itself (a send and its receive of ONE rank are grouped)
        "src = ctypes.c_void_p(st.value); dst = ctypes.c_void_p(st.value + 1024)\n"
        "w = _lib.Context()\n"
        "x = (np.arange(32, dtype=np.int64) * 7 + 3)\n"
        "w.check(w._L.shp_dev_upload(w.handle, src, _lib.ptr(x), x.nbytes))\n"      # producer: w's stream
        "c.group(True); c.isend_dev(src.value, 256, 0, w); c.irecv_dev(dst.value, 256, 0, None); c.group(False)\n"
        "c.wait_dev(w)\n"
        "y = np.zeros(32, dtype=np.int64)\n"
        "w.check(w._L.shp_dev_download(w.handle, _lib.ptr(y), dst, 256))\n"        # consumer: waits on the device
        "assert np.array_equal(x, y)\n"
        "c.drain(); w.close()\n"
        "c.close()\n" % ROOT)
    _run_ranks(1, ['-c', code], tmp_path, timeout=300)


@pytest.mark.parametrize('world,mode', [(1, 'sequential'), (1, 'parallel'), (2, 'parallel'), (2, 'sequential')])
def test_hip_engine_chain_matches_single_process(world, mode, tmp_path):
    """both forms of the sharded stitch (sequential chain / provisional ids + renumber) against the
    in-process tiled run"""
    from pyshepseg_amd import tiling
    _run_ranks(world, [os.path.join(ROOT, 'tests', 'dist_worker_gpu.py'), str(tmp_path),
                       'socket' if world > 1 else 'rccl'], tmp_path, extra={'SHEPSEG_STITCH': mode})
    parts = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    assert {str(q['mode']).split('->')[0] for q in parts} == {mode}
    print('stitch mode:', {str(q['mode']) for q in parts})
    from pyshepseg_amd import tilingstats
    from oracle import oracle
    ras = tiling.DeviceRaster.synth(11, 6, 1500, 1300)
    try:
        cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS, numWorkers=3)
        ref = tiling.doTiledShepherdSegmentation(ras, None, tileSize=512, overlapSize=128,
                                                 minSegmentSize=50, numClusters=30,
                                                 fixedKMeansInit=True, concurrencyCfg=cfg)
    finally:
        ras.free()
    got = np.zeros_like(ref.segimg)
    for q in parts:
        assert np.array_equal(q['centres'], ref.kmeans.cluster_centers_)
        lo, hi = int(q['outLo']), int(q['outHi'])
        got[lo:hi] = np.maximum(got[lo:hi], q['out'])
        assert int(q['maxSegId']) == ref.maxSegId
        assert np.array_equal(q['hist'], ref.hist)
    assert np.array_equal(got, ref.segimg)
    # statistics sharded over the ranks == the single-GPU statistics of the whole raster
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'),
           ('f', 'mode'), ('g', 'percentile', 90), ('h', 'pixcount')]
    band = oracle.synthimg(11, 6, 1500, 1300)[2]
    wic, wfc, _f = tilingstats.calcPerSegmentStats(ref.segimg, band, sel, maxSegId=ref.maxSegId)
    for r in range(world):
        st = np.load(tmp_path / ('stats%d.npz' % r))
        assert np.array_equal(st['ic'], wic)
        assert np.array_equal(st['fc'].view(np.uint32), wfc.view(np.uint32))


class _ThreadDevComm(object):
    """An in-process stand-in for RcclComm's device collectives: `world` threads of one process, one GPU.
    allgather_dev copies device to device, allreduce_dev_i64 adds through the host (test data is small)."""
    onDevice = True

    def __init__(self, rank, world, shared):
        import threading
        (self.rank, self.world, self.sh) = (rank, world, shared)
        if 'bar' not in shared:                # (made once, before the rank threads start: a second barrier
            shared['bar'] = threading.Barrier(world)        # would strand whoever waits at the first)
            shared['slots'] = [None] * world
        self.c = None

    def _xchg(self, v):
        self.sh['bar'].wait()
        self.sh['slots'][self.rank] = v
        self.sh['bar'].wait()
        out = list(self.sh['slots'])
        self.sh['bar'].wait()
        return out

    def allgather_obj(self, obj):
        return self._xchg(obj)

    def allgather_dev(self, d_send, d_recv, nbytes):
        import ctypes
        ptrs = self._xchg(d_send)
        for (r, p) in enumerate(ptrs):
            self.c.check(self.c._L.shp_dev_copy(self.c.handle, ctypes.c_void_p(d_recv + r * nbytes), ctypes.c_void_p(p), nbytes))
        self.sh['bar'].wait()                  # nobody frees a send buffer another rank still reads

    def allreduce_dev_i64(self, d_buf, count):
        import ctypes
        from pyshepseg_amd import _lib
        mine = np.empty(count, dtype=np.int64)
        self.c.check(self.c._L.shp_dev_download(self.c.handle, _lib.ptr(mine), ctypes.c_void_p(d_buf), mine.nbytes))
        tot = np.sum(self._xchg(mine), axis=0, dtype=np.int64)
        self.c.check(self.c._L.shp_dev_upload(self.c.handle, ctypes.c_void_p(d_buf), _lib.ptr(tot), tot.nbytes))


@pytest.mark.parametrize('world,dtype,nullv', [(2, np.uint16, None), (3, np.uint16, 7), (3, np.uint8, None), (2, np.int16, -5)])
def test_device_stats_split_with_straddlers(world, dtype, nullv, oracle):
    """The device-resident data path of calcPerSegmentStatsDistributed (shp_dstats_local_dev -> all-gather of
    the packed pairs -> shp_dstats_merge_dev by id share -> all-reduce of the column block) with `world`
    row shards of one raster on this one GPU, segments crossing every shard boundary, against the oracle on
    the whole raster: every column bit for bit on every rank."""
    import ctypes
    import threading
    from pyshepseg_amd import distributed, tilingstats, _lib
    rng = np.random.default_rng(world * 10 + np.dtype(dtype).itemsize)
    (nr, nc) = (203, 190)
    # segments: blobs of a coarse random field -> long vertical streaks cross the shard boundaries
    base = rng.integers(1, 40, size=(nr // 7 + 2, nc // 5 + 1))
    seg = np.kron(base, np.ones((7, 5), dtype=np.int64))[:nr, :nc]
    seg = (seg + (np.arange(nc)[None, :] // 37) * 40).astype(np.uint32)
    seg[rng.random((nr, nc)) < 0.03] = 0
    S = int(seg.max()) + 3                                             # ids nobody holds at the top
    info = np.iinfo(dtype)
    band = rng.integers(max(info.min, -300), min(info.max, 300) + 1, size=(nr, nc)).astype(dtype)
    if nullv is not None:
        band[rng.random((nr, nc)) < 0.1] = nullv
        band[seg == 5] = nullv                                         # an all-nodata segment
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'), ('f', 'mode'),
           ('g', 'percentile', 30), ('h', 'pixcount')]
    (fast, nInt, nFloat) = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)
    (wic, wfc) = oracle.segstats(seg, band, sel, nullv, -9999, max_seg_id=S)
    hist = np.bincount(seg.ravel(), minlength=S + 1).astype(np.uint32)
    hist[0] = 0
    cuts = [0] + [int(round(nr * (r + 1) / world)) for r in range(world)]
    shared, results, errors = {}, [None] * world, []

    def rank(r):
        try:
            c = _lib.Context()
            comm = _ThreadDevComm(r, world, shared)
            comm.c = c
            (lo, hi) = (cuts[r], cuts[r + 1])
            (ds, db) = (ctypes.c_void_p(), ctypes.c_void_p())
            ss, bb = np.ascontiguousarray(seg[lo:hi]), np.ascontiguousarray(band[lo:hi])
            c.check(c._L.shp_dev_alloc(c.handle, ss.nbytes, ctypes.byref(ds)))
            c.check(c._L.shp_dev_alloc(c.handle, bb.nbytes, ctypes.byref(db)))
            c.check(c._L.shp_dev_upload(c.handle, ds, _lib.ptr(ss), ss.nbytes))
            c.check(c._L.shp_dev_upload(c.handle, db, _lib.ptr(bb), bb.nbytes))
            results[r] = distributed.deviceStats(c, comm, ds.value, db.value, _lib.SHP_DTYPES[np.dtype(dtype)], hi - lo, nc,
                                                 hist, fast, nInt, nFloat, -9999, nullv)
            c.check(c._L.shp_dev_free(c.handle, ds))
            c.check(c._L.shp_dev_free(c.handle, db))
            c.close()
        except BaseException as e:      # noqa: B902  (a dead rank must not leave the others at a barrier)
            errors.append(e)
            try:
                shared['bar'].abort()
            except Exception:
                pass
    _ThreadDevComm(0, world, shared)          # the barrier exists before any thread runs
    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    held = [set(np.unique(seg[cuts[r]:cuts[r + 1]])) - {0} for r in range(world)]
    strad = set()
    for a in range(world):
        for b in range(a + 1, world):
            strad |= held[a] & held[b]
    assert len(strad) > 10
    for (ic, fc, nStrad, nPix) in results:
        assert np.array_equal(ic, wic)
        assert np.array_equal(fc.view(np.uint32), wfc.view(np.uint32))
        assert nStrad == len(strad) and nPix == int(np.isin(seg, list(strad)).sum())
