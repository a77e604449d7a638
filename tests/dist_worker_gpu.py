"""Rank program of tests/test_gpu_distributed.py: the multi-GPU driver with the HIP engine.
Transport 'socket': every rank uses GPU 0 (one-GPU test box), strips staged through host memory;
'rccl': one GPU per rank, strips device to device."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    outdir, transport = sys.argv[1], sys.argv[2]
    if transport == 'socket':
        os.environ['SHEPSEG_DEVICE'] = '0'
    else:
        os.environ['SHEPSEG_DEVICE'] = os.environ.get('LOCAL_RANK', '0')
    from pyshepseg_amd import distributed, tiling
    from pyshepseg_amd import comm as shpcomm
    comm = shpcomm.SocketComm() if transport == 'socket' else shpcomm.RcclComm()
    nb, nr, nc = 6, 1500, 1300

    def makeSlice(yLo, yHi):
        return tiling.DeviceRaster.synth(11, nb, yHi - yLo, nc, y0=yLo, x0=0)
    eng = distributed.HipEngine(makeSlice, numWorkers=3, keepOutput=True)
    r = distributed.runDistributed(eng, comm, nr, nc, 512, 128, minSegmentSize=50, numClusters=30,
                                   fixedKMeansInit=True)
    out = eng.localOutput()
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'),
           ('f', 'mode'), ('g', 'percentile', 90), ('h', 'pixcount')]
    ic, fc, _fast = distributed.calcPerSegmentStatsDistributed(eng, comm, r.hist, 3, sel)
    np.savez(os.path.join(outdir, 'stats%d.npz' % comm.rank), ic=ic, fc=fc)
    eng.releaseOutput()
    np.savez(os.path.join(outdir, 'rank%d.npz' % comm.rank), out=out, outLo=r.outRows[0],
             outHi=r.outRows[1], maxSegId=r.maxSegId, hist=r.hist,
             centres=r.kmeans.cluster_centers_, msd=r.maxSpectralDiff, mode=r.stitchMode)
    comm.close()


if __name__ == '__main__':
    main()
