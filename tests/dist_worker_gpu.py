"""Rank program of tests/test_gpu_distributed.py: the multi-GPU driver with the HIP engine.
With backend gloo every rank uses GPU 0 (one-GPU test box); with nccl one GPU per rank."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    outdir, backend = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    dist.init_process_group(backend=backend)
    if backend == 'gloo':
        os.environ['SHEPSEG_DEVICE'] = '0'
        device = None
    else:
        import torch
        lr = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(lr)
        device = torch.device('cuda', lr)
    from pyshepseg_amd import distributed, tiling
    comm = distributed.Comm(dist, device=device)
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
             centres=r.kmeans.cluster_centers_, msd=r.maxSpectralDiff)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
