"""CPU (socket transport, world_size 2 and 3): the multi-GPU driver's sharding, k-means gather/broadcast,
stitch chain with boundary exchange and histogram all-reduce, run with the oracle engine, must
reproduce the single-process tiled result exactly."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


class _Ds(object):
    def __init__(self, ys, xs):
        self.RasterYSize, self.RasterXSize = ys, xs


def test_socket_comm_collectives(tmp_path):
    """the socket transport's point-to-point and collectives at world size 3"""
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from pyshepseg_amd import comm as C\n"
        "c = C.SocketComm()\n"
        "r, w = c.rank, c.world\n"
        "assert c.allgather_obj({'r': r}) == [{'r': i} for i in range(w)]\n"
        "assert c.bcast_obj('x' * 100000 if r == 1 else None, src=1) == 'x' * 100000\n"
        "assert c.allreduce_sum_i64(np.arange(5) * (r + 1)).tolist() == (np.arange(5) * sum(range(1, w + 1))).tolist()\n"
        "assert c.max_f64(1.5 * r) == 1.5 * (w - 1)\n"
        "a = np.arange(300000, dtype=np.uint32) + r\n"
        "c.send_bytes(a, (r + 1) %% w)\n"
        "b = np.frombuffer(c.recv_bytes((r - 1) %% w), dtype=np.uint32)\n"
        "assert np.array_equal(b, np.arange(300000, dtype=np.uint32) + (r - 1) %% w)\n"
        "c.barrier(); c.close()\n" % ROOT)
    _run_ranks(3, ['-c', code], tmp_path, timeout=120)


def test_shard_tile_rows():
    from pyshepseg_amd import tiling, distributed
    ti = tiling.getTilesForFile(_Ds(40000, 40000), 4096, 1024)
    for world in (1, 2, 3, 4, 8, 12, 16):
        sh = distributed.shardTileRows(ti, world)
        assert len(sh) == world
        rows = [r for (a, b) in sh for r in range(a, b)]
        assert rows == list(range(ti.nrows))                       # contiguous, complete, ordered
        assert sum(1 for (a, b) in sh if b > a) == min(world, ti.nrows)
    sh = distributed.shardTileRows(ti, 8)
    assert max(b - a for (a, b) in sh) <= 2


def test_shard_tiles():
    from pyshepseg_amd import tiling, distributed
    ti = tiling.getTilesForFile(_Ds(40000, 40000), 4096, 1024)
    nt = ti.ncols * ti.nrows
    for world in (1, 2, 3, 4, 5, 8, 12, 16, 200):
        sh = distributed.shardTiles(ti, world)
        assert len(sh) == world
        assert [t for (a, b) in sh for t in range(a, b)] == list(range(nt))       # contiguous, complete
        ne = [(a, b) for (a, b) in sh if b > a]
        assert all(b - a >= ti.ncols for (a, b) in ne[:-1])         # top neighbours: local or previous rank
        for p, (a, b) in enumerate(sh):
            if b > a:
                for (kind, col, row, h, w) in distributed.boundaryPlan(ti, sh, p, 1024):
                    assert a <= row * ti.ncols + col < b
    sh = distributed.shardTiles(ti, 8)                               # 144 tiles, balanced by pixel area
    area = [sum(ti.getTile(t % 12, t // 12)[2] * ti.getTile(t % 12, t // 12)[3] for t in range(a, b))
            for (a, b) in sh]
    assert max(area) < 1.06 * min(area)
    assert [b - a for (a, b) in distributed.shardTiles(ti, 16)].count(12) == 12     # whole rows


def test_shard_plan_random_grids():
    """Random rasters / tile sizes / world sizes: every tile's top and left neighbours are either
    in the same shard or delivered by the previous shard's boundary plan."""
    from pyshepseg_amd import tiling, distributed
    rng = np.random.RandomState(7)
    for _case in range(200):
        (nr, nc) = (int(rng.randint(50, 3000)), int(rng.randint(50, 3000)))
        tile = int(rng.randint(32, 700))
        ov = 2 * int(rng.randint(1, max(2, tile // 4)))
        ti = tiling.getTilesForFile(_Ds(nr, nc), tile, ov)
        world = int(rng.randint(1, 12))
        sh = distributed.shardTiles(ti, world)
        nt = ti.ncols * ti.nrows
        assert [t for (a, b) in sh for t in range(a, b)] == list(range(nt))
        ne = [i for i, (a, b) in enumerate(sh) if b > a]
        for pos, r in enumerate(ne):
            (a, b) = sh[r]
            got = set()
            if pos > 0:
                got = {(k, col, row) for (k, col, row, _h, _w) in
                       distributed.boundaryPlan(ti, sh, ne[pos - 1], ov)}
            for t in range(a, b):
                (col, row) = (t % ti.ncols, t // ti.ncols)
                if row > 0 and not (a <= t - ti.ncols < b):
                    assert ('b', col, row - 1) in got
                if col > 0 and not (a <= t - 1 < b):
                    assert ('r', col - 1, row) in got


def _run_ranks(world, argv, tmp_path, timeout=600, extra_env=None):
    """start `world` rank processes with the environment a launcher sets; all must exit 0"""
    procs = []
    import secrets
    nonce = secrets.token_hex(8)
    for r in range(world):
        env = dict(os.environ, SHEPSEG_LAUNCH_NONCE=nonce, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT='0', OMP_NUM_THREADS='1',
                   SHEPSEG_COMM_DIR=str(tmp_path / 'comm'))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for (p, (_o, e)) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]


STITCH_GOLDEN = ['stitch_2x2', 'stitch_3x3_null', 'stitch_3x4_8conn', 'stitch_quirk_empties',
                 'stitch_quirk_zeros']


@pytest.mark.parametrize('name', STITCH_GOLDEN)
@pytest.mark.parametrize('world', [2, 3])
def test_parallel_stitch_equals_reference_mosaic(name, world, tmp_path):
    """The parallel (provisional-id) stitch of the sharded driver against the REFERENCE's own stitched
    mosaics, both quirk fixtures included: where a tile hands out an id that its trimmed window
    does not show (stitch_quirk_empties) the driver must notice and redo the stitch sequentially."""
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    g = np.load(path, allow_pickle=True)
    _run_ranks(world, [os.path.join(ROOT, 'tests', 'dist_worker.py'), str(tmp_path), '0', '0', '0', path],
               tmp_path)
    parts = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    want = g['mosaic']
    got = np.zeros_like(want)
    modes = set()
    for q in parts:
        lo, hi = int(q['outLo']), int(q['outHi'])
        got[lo:hi] = np.maximum(got[lo:hi], q['out'])
        assert int(q['maxSegId']) == int(g['max_seg_id'])
        assert np.array_equal(q['hist'], g['hist'])
        modes.add(str(q['mode']))
    assert np.array_equal(got, want)
    assert len(modes) == 1                       # every rank took the same decision
    redone = {int(q['redone']) for q in parts}
    ntiles = int(parts[0]['ntiles'])
    assert len(redone) == 1
    if name == 'stitch_quirk_empties':
        # the chain is redone only from the tile after the first one that hides an id it handed out: the
        # tiles up to it keep their (renumbered) result
        assert modes == {'parallel->sequential'}
        assert 0 < redone.pop() < ntiles
    elif name in ('stitch_2x2', 'stitch_3x3_null', 'stitch_3x4_8conn'):
        assert modes == {'parallel'} and redone == {0}


@pytest.mark.parametrize('world,simple,NR,mode,shard,order', [
    (2, 0, 330, 'parallel', 'rows', 'diagonal'), (3, 0, 330, 'parallel', 'tiles', 'diagonal'),
    (2, 1, 330, 'sequential', 'tiles', 'diagonal'), (3, 0, 150, 'parallel', 'rows', 'diagonal'),
    (3, 0, 330, 'sequential', 'tiles', 'diagonal'), (2, 0, 330, 'parallel', 'tiles', 'rowmajor'),
    (3, 0, 330, 'sequential', 'rows', 'diagonal'), (2, 0, 330, 'parallel', 'rows', 'rowmajor'),
    (2, 0, 330, 'parallel', 'tiles', 'diagonal')])
def test_two_rank_chain_matches_single_process(world, simple, NR, mode, shard, order, tmp_path, oracle):
    # order: a rank's chain steps of the parallel stitch along anti-diagonals or in row-major order
    # NR = 150: two tile rows for three ranks -> whole-row shards and a rank without tiles
    img = oracle.synthimg(31, 3, NR, 260)
    img[:, :4, :] = 65535                      # a null border row band (nulls are not given here)
    np.save(tmp_path / 'img.npy', img)
    tile, ov = 96, 32
    _run_ranks(world, [os.path.join(ROOT, 'tests', 'dist_worker.py'), str(tmp_path), str(tile), str(ov),
                       str(simple)], tmp_path,
               extra_env={'SHEPSEG_STITCH': mode, 'SHEPSEG_SHARD': shard, 'SHEPSEG_CHAIN_ORDER': order})
    parts = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
    assert {str(q['mode']).split('->')[0] for q in parts} == {mode}
    # single-process reference: oracle tiles + oracle stitch with the same centres
    centres, msd = parts[0]['centres'], float(parts[0]['msd'])
    for q in parts[1:]:
        assert np.array_equal(q['centres'], centres)
    tiles, ntc, ntr = oracle.get_tiles(NR, 260, tile, ov)
    local = {}
    for (c, r), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        local[(c, r)] = oracle.segment_tile(sub, centres, 12, msd, None, True)['segimg']
    want, mx, hist = oracle.stitch_tiles(local, tiles, ntc, ntr, NR, 260, ov, simple=bool(simple))
    got = np.zeros_like(want)
    cover = np.zeros(NR, dtype=bool)
    for q in parts:
        lo, hi = int(q['outLo']), int(q['outHi'])
        got[lo:hi] = np.maximum(got[lo:hi], q['out'])          # a rank writes only its tiles' windows
        cover[lo:hi] = True
        assert int(q['maxSegId']) == mx
        assert np.array_equal(q['hist'], hist)
    assert cover.all()
    assert np.array_equal(got, want)
    # per-segment statistics sharded the same way == the oracle on the whole raster, on every rank
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'),
           ('f', 'mode'), ('g', 'percentile', 25), ('h', 'pixcount')]
    wic, wfc = oracle.segstats(want, np.ascontiguousarray(img[1]), sel, 65535, -9999, max_seg_id=mx)
    for r in range(world):
        st = np.load(tmp_path / ('stats%d.npz' % r))
        assert np.array_equal(st['ic'], wic)
        assert np.array_equal(st['fc'].view(np.uint32), wfc.view(np.uint32))
        # segments DO straddle the rank boundaries (their pixels travelled as raw arrays and were reduced by the
        # rank whose share of the id space holds them): the ids present in more than one rank's rows
        if world > 1:
            held = [set(np.unique(q['out'])) - {0} for q in parts]
            strad = set()
            for a in range(world):
                for b in range(a + 1, world):
                    strad |= held[a] & held[b]
            assert int(st['straddlers']) == len(strad) and (simple or NR < 330 or len(strad) > 0)


@pytest.mark.parametrize('seed', range(8))
def test_parallel_stitch_fuzz_in_process(seed, oracle):
    """One rank, both forms of the stitch on random small rasters with many tiles: identical mosaics,
    maxSegId and histograms whichever way the parallel form ends (kept, or redone sequentially)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dist_oracle_engine import OracleEngine
    from pyshepseg_amd import distributed, shepseg
    from pyshepseg_amd import comm as shpcomm
    rng = np.random.default_rng(seed)
    (nr, nc) = (int(rng.integers(120, 260)), int(rng.integers(120, 260)))
    img = oracle.synthimg(100 + seed, 3, nr, nc)
    if seed % 2:
        img[:, : int(rng.integers(1, 9)), :] = 65535
    (tile, ov) = [(48, 32), (64, 24), (80, 40), (56, 16)][seed % 4]
    xs = shepseg._sample_rows(img, 100, 65535 if seed % 2 else None)
    init = shepseg.diagonalClusterCentres(xs, 6).astype(np.float64)
    centres, _l, _n = oracle.kmeans_fit(xs.astype(np.float64), init)
    res = {}
    for mode in ('sequential', 'parallel'):
        eng = OracleEngine(img, oracle)
        r = distributed.runDistributed(
            eng, shpcomm.LocalComm(), nr, nc, tile, ov, minSegmentSize=int(rng.integers(8, 30)) if mode == 'x' else 14,
            maxSpectralDiff='auto', imgNullVal=(65535 if seed % 2 else None), fourConnected=bool(seed % 3),
            kmeansObj=shepseg.KMeansModel(centres), stitchMode=mode)
        res[mode] = (eng.out.copy(), r.maxSegId, r.hist.copy(), r.stitchMode)
    assert res['sequential'][3] == 'sequential'
    assert res['parallel'][3] in ('parallel', 'parallel->sequential')
    assert np.array_equal(res['sequential'][0], res['parallel'][0])
    assert res['sequential'][1] == res['parallel'][1]
    assert np.array_equal(res['sequential'][2], res['parallel'][2])


def test_socket_comm_handshake_and_private_rendezvous(tmp_path, monkeypatch):
    """a connection that does not hold the launch key is dropped and the acceptor keeps accepting; the
    rendezvous directory must be private to this user"""
    import struct
    from pyshepseg_amd import comm as C
    d = tmp_path / 'rv'
    monkeypatch.setenv('SHEPSEG_COMM_DIR', str(d))
    c = C.SocketComm(rank=0, world=1)
    assert (os.stat(d).st_mode & 0o777) == 0o700
    port = c.srv.getsockname()[1]
    for junk in (b'', b'\x00' * 3, struct.pack('<i', 0) + b'x' * 32):
        s = socket.create_connection(('127.0.0.1', port), timeout=10)
        s.recv(16)
        s.sendall(junk)
        s.close()
    assert 0 not in c.inc                                      # nobody was registered as rank 0
    c.send_bytes(np.arange(1000, dtype=np.uint32), 0)          # the real thing still gets through
    assert np.array_equal(np.frombuffer(c.recv_bytes(0, timeout=20), dtype=np.uint32), np.arange(1000))
    c.closing = True
    c.srv.close()
    os.chmod(d, 0o777)
    with pytest.raises(C.CommError, match='not a private directory'):
        C.rendezvousDir()
