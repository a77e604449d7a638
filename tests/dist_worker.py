"""Rank program of tests/test_distributed_cpu.py (socket transport, no GPU): runs the multi-GPU driver with
the oracle engine and writes this rank's output rows to disk."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    outdir = sys.argv[1]
    tile, ov, simple = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from oracle import oracle
    from pyshepseg_amd import distributed
    from pyshepseg_amd import comm as shpcomm
    from dist_oracle_engine import OracleEngine
    comm = shpcomm.SocketComm()
    golden = sys.argv[5] if len(sys.argv) > 5 else None
    if golden:
        # a stitch fixture: the reference's own image, model and parameters; only the label mosaic is
        # compared (by the test), so nothing else is written
        from pyshepseg_amd import shepseg
        g = np.load(golden, allow_pickle=True)
        img = g['img']
        eng = OracleEngine(img, oracle)
        r = distributed.runDistributed(
            eng, comm, img.shape[1], img.shape[2], int(g['tile_size']), int(g['overlap']),
            minSegmentSize=int(g['min_seg']), maxSpectralDiff=float(g['msd']),
            imgNullVal=(int(g['null_val']) if int(g['has_null']) else None),
            fourConnected=bool(int(g['four'])), kmeansObj=shepseg.KMeansModel(g['centres']))
        np.savez(os.path.join(outdir, 'rank%d.npz' % comm.rank), out=eng.out, outLo=r.outRows[0],
                 outHi=r.outRows[1], maxSegId=r.maxSegId, hist=r.hist, mode=r.stitchMode,
                 redone=r.chainStepsRedone, ntiles=r.numTileRows * r.numTileCols)
        comm.close()
        return
    img = np.load(os.path.join(outdir, 'img.npy'))
    eng = OracleEngine(img, oracle)
    r = distributed.runDistributed(eng, comm, img.shape[1], img.shape[2], tile, ov,
                                   minSegmentSize=12, numClusters=8, fixedKMeansInit=True,
                                   simpleTileRecode=bool(simple))
    sel = [('a', 'min'), ('b', 'max'), ('c', 'mean'), ('d', 'stddev'), ('e', 'median'),
           ('f', 'mode'), ('g', 'percentile', 25), ('h', 'pixcount')]
    info = {}
    ic, fc, _fast = distributed.calcPerSegmentStatsDistributed(eng, comm, r.hist, 2, sel,
                                                               imgNullVal=65535, info=info)
    np.savez(os.path.join(outdir, 'stats%d.npz' % comm.rank), ic=ic, fc=fc, straddlers=info['straddlers'],
             straddler_pixels=info['straddler_pixels'])
    np.savez(os.path.join(outdir, 'rank%d.npz' % comm.rank), out=eng.out, outLo=r.outRows[0],
             outHi=r.outRows[1], maxSegId=r.maxSegId, hist=r.hist,
             centres=r.kmeans.cluster_centers_, msd=r.maxSpectralDiff, rows=np.array(r.rowRange),
             mode=r.stitchMode)
    comm.close()


if __name__ == '__main__':
    main()
