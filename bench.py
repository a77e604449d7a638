#!/usr/bin/env python
"""Headline benchmark: Mpixels/s segmented on synthetic 6-band 40000x40000 tiled imagery.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one complete tiled Shepherd segmentation of the image (global k-means subsample +
Lloyd fit, every tile through assign -> clump -> elimination, cross-tile stitch, histogram)
with the image already resident in HBM (synthimg v1 generated on the device) and the
stitched labels left in HBM.  Workload = BASELINE.json configs[2] (C3): tile 4096 / overlap
1024, k = 60, minSegmentSize = 50, fixed k-means init.  value = image pixels / step time.

torch is used only for process-group plumbing (barrier, max over ranks); the product path is
pyshepseg_amd -> ctypes -> libshepseg_hip.so.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# must be in the environment before the HIP runtime starts (torch initialises it first when N > 1)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=40000, help='image rows = cols')
    ap.add_argument('--bands', type=int, default=6)
    ap.add_argument('--tile', type=int, default=4096)
    ap.add_argument('--overlap', type=int, default=1024)
    ap.add_argument('--workers', type=int, default=int(os.environ.get('SHEPSEG_WORKERS', '20')))
    ap.add_argument('--simple-recode', type=int, default=0, help='diagnostic: simpleTileRecode')
    ap.add_argument('--cpu-sample', type=int, default=9216,
                    help='window edge of the cpu_baseline sample (0 = skip)')
    return ap.parse_args()


def prof_totals(contexts):
    """sum shp_prof_get over worker contexts -> {id: (ms, count)}"""
    tot = {}
    for c in contexts:
        ms = (ctypes.c_double * 16)()
        cnt = (ctypes.c_uint64 * 16)()
        c._L.shp_prof_get(c.handle, ms, cnt, 16, 1)
        for i in range(16):
            a, b = tot.get(i, (0.0, 0))
            tot[i] = (a + ms[i], b + cnt[i])
    return tot


def cpu_baseline(ras, args, centres, msd):
    """The C oracle (a port of the reference, oracle/shepseg_oracle.c) timed on this box's host
    cores, single thread, on a bounded window of the same image with the same tiling."""
    from oracle import oracle
    from pyshepseg_amd import _lib
    oracle.build()
    w = min(args.cpu_sample, args.size)
    idx = np.arange(w, dtype=np.uint32)
    img = np.empty((args.bands, w, w), dtype=np.uint16)
    c = _lib.ctx()
    c.check(c._L.shp_dev_subsample(c.handle, ctypes.c_void_p(ras.ptr), 2, args.bands, ras.shape[1],
                                   ras.shape[2], _lib.ptr(idx), w, _lib.ptr(idx), w, _lib.ptr(img)))
    t0 = time.time()
    tiles, ntc, ntr = oracle.get_tiles(w, w, args.tile, args.overlap)
    local = {}
    for (tc, tr), (x, y, xs, ys) in tiles.items():
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        local[(tc, tr)] = oracle.segment_tile(sub, centres, 50, msd, None, True)['segimg']
    oracle.stitch_tiles(local, tiles, ntc, ntr, w, w, args.overlap)
    dt = time.time() - t0
    return {"value": round(w * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "top-left %dx%d window of the same synthetic image, %d tiles (tile %d / "
                      "overlap %d) + stitch, %.1f s of single-thread C oracle"
                      % (w, w, len(tiles), args.tile, args.overlap, dt)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (this process has not touched the GPU and never does), relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if p.returncode != 0 or line is None:
        sys.stderr.write('bench.py: the %d-rank run failed (exit code %d)\n' % (args.gpus, p.returncode))
        sys.exit(p.returncode or 1)
    if json.loads(line).get('n_gpus') != args.gpus:
        sys.stderr.write('bench.py: the ranks report n_gpus != %d\n' % args.gpus)
        sys.exit(1)
    print(line)


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)
    os.environ.setdefault('SHEPSEG_DEVICE', str(local_rank))
    dist = None
    force_dist = os.environ.get('SHEPSEG_FORCE_DIST', '0') == '1' and 'RANK' in os.environ
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl')

    from pyshepseg_amd import tiling, _lib
    if world > 1 or force_dist:
        from pyshepseg_amd import distributed
        return distributed.bench_main(args, rank, world, local_rank, dist)

    ras = tiling.DeviceRaster.synth(11, args.bands, args.size, args.size)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS,
                                               numWorkers=args.workers)

    def step():
        r = tiling.doTiledShepherdSegmentation(
            ras, tiling._KEEP_ON_DEVICE, tileSize=args.tile, overlapSize=args.overlap,
            minSegmentSize=50, numClusters=60, fixedKMeansInit=True, concurrencyCfg=cfg,
            simpleTileRecode=bool(args.simple_recode))
        tiling.freeDeviceOutput(r)
        return r

    for _ in range(args.warmup):
        r = step()
    prof_totals(_lib.pool_contexts())           # reset the per-kernel timers
    c = _lib.ctx()
    c.check(c._L.shp_sync(c.handle))
    t0 = time.time()
    step_s = []
    for _ in range(args.steps):
        ts = time.time()
        r = step()
        step_s.append(round(time.time() - ts, 4))
    c.check(c._L.shp_sync(c.handle))
    dt = (time.time() - t0) / max(args.steps, 1)
    prof = prof_totals(_lib.pool_contexts())

    npix = args.size * args.size
    value = npix / dt / 1e6
    # dominant kernel by accumulated device time
    names = {0: 'k_assign (cluster-map blocks)', 1: 'ccl (k_ccl_local+k_ccl_border+k_ccl_flatten)', 2: 'k_dfs_split',
             3: 'CSR build (k_run_tile_*+k_sort_hist+k_sort_scatter+k_run_expand)', 4: 'k_spectra_small+k_spectra_big',
             5: 'k_small_loop', 7: 'seed scan + k_clump_final'}
    dom = max((i for i in names), key=lambda i: prof.get(i, (0, 0))[0])
    ms, cnt = prof[dom]
    ti = tiling.getTilesForFile(ras, args.tile, args.overlap)
    tile_px = sum(t[2] * t[3] for t in ti.tiles.values()) / max(len(ti.tiles), 1)
    # algorithmic bytes per launch: clump kernels read a 2-byte cluster id and write a 4-byte label
    # per tile pixel; assign reads nB*2 B and writes 2 B; the rest move (2*nB + 4) B per pixel.
    bpp = {0: 2 * args.bands + 2, 1: 6, 2: 6, 7: 6}.get(dom, 2 * args.bands + 4)
    avg_s = (ms / max(cnt, 1)) / 1e3
    achieved = (bpp * tile_px / avg_s / 1e9) if avg_s > 0 else 0.0
    # HBM traffic per launch of that kernel from the committed PMC passes (rocprofv3 cannot run
    # inside this process): FETCH_SIZE + WRITE_SIZE in KB, see profiles/r01_n_pmc_summary.json
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_n_pmc_summary.json')))['kernels']
        k = pmc.get(names[dom].split(' ')[0])
        if k and args.size == 40000:
            traffic = int((k['FETCH_SIZE_KB_per_launch'] + k['WRITE_SIZE_KB_per_launch']) * 1024)
            if dom == 2:          # the timed region holds a tile's k_dfs_split launches (1 or 2)
                traffic = int(traffic * k['launches'] / 144.0)
    except Exception:
        traffic = None
    out = {
        "metric": "Mpixels/sec segmented, 6-band 40k x 40k tiled",
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "C3: tiled %dx%d, %d-band uint16 synthimg v1, tileSize=%d, "
                               "overlap=%d, k=60, minSegmentSize=50, fixedKMeansInit, image + "
                               "labels resident in HBM" % (args.size, args.size, args.bands,
                                                           args.tile, args.overlap),
                   "tiles": len(ti.tiles), "worker_streams": args.workers,
                   "max_seg_id": int(r.maxSegId)},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "avg_launch_ms": round(ms / max(cnt, 1), 3), "launches": int(cnt),
                     "bytes_per_launch": int(bpp * tile_px),
                     "whole_path_frac_of_hbm_roofline":
                         round(value * 1e6 * (2 * args.bands + 4) / 1e9 / HBM_PEAK_GBS, 6),
                     "device_ms_by_kernel": {names[i]: round(prof.get(i, (0, 0))[0] / max(args.steps, 1), 1)
                                             for i in names}},
    }
    out["config"]["step_s"] = step_s
    out["config"]["host_timers_s"] = {k: round(v['total'], 3)
                                       for k, v in r.timings.makeSummaryDict().items()}
    if args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(ras, args, r.kmeans.cluster_centers_,
                                           float(r.maxSpectralDiff))
    print(json.dumps(out))
    ras.free()


if __name__ == '__main__':
    main()
