"""GPU: the multi-GPU driver with the HIP engine.  The test box has one GPU, so two ranks
share it and exchange the boundary strips over gloo (host-staged); the strips, the maxSegId
chain and the histogram all-reduce are the same code as with nccl.  Result must equal the
single-process tiled run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world', [1, 2])
def test_hip_engine_chain_matches_single_process(world, tmp_path):
    from pyshepseg_amd import tiling
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker_gpu.py'), str(tmp_path), 'gloo']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    parts = [np.load(tmp_path / ('rank%d.npz' % r)) for r in range(world)]
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
