#!/usr/bin/env python
"""Headline benchmark: Mpixels/s segmented on synthetic 6-band 40000x40000 tiled imagery.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c4|c5]
    (N > 1 without a launcher: bench.py starts the N ranks itself; under
     `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one rank)

Workloads (BASELINE.json configs):
  c3 (default, the config the metric is quoted on): a step = one complete tiled Shepherd
      segmentation of the 40000^2 x 6 image (global k-means subsample + the reference's k-means fit, every tile
      through assign -> clump -> elimination, cross-tile stitch, histogram) with the image already
      resident in HBM (synthimg v1 generated on the device) and the stitched labels left in HBM;
      tile 4096 / overlap 1024, k = 60, minSegmentSize = 50, fixed k-means init.
  c4: the same with the 10-band image synthimg(13, 10, ...).
  c5: a step = per-segment statistics (mean, stddev, median, pixcount) of one uint16 band over a
      1.6 Gpx label raster of 50 M segments (4 x 8-pixel blocks), rasters resident in HBM, the
      result columns copied to the host.
value = image pixels / step time.

No torch anywhere: with N > 1 the ranks talk over RCCL bound directly behind the C-ABI
(pyshepseg_amd/comm.py); the product path is pyshepseg_amd -> ctypes -> libshepseg_hip.so.
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
# must be in the environment before the HIP runtime starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=('c3', 'c4', 'c5'), default='c3')
    ap.add_argument('--source', choices=('hbm', 'npy'), default='hbm',
                    help="c3/c4: 'npy' = the image is a .npy file read through the I/O pipeline and the "
                         "labels are written to a .npy file (PCIe-inclusive; never the headline value)")
    ap.add_argument('--scratch', default=os.environ.get('SHEPSEG_SCRATCH', '/tmp'),
                    help='directory of the --source npy files')
    ap.add_argument('--size', type=int, default=40000, help='image rows = cols')
    ap.add_argument('--bands', type=int, default=None, help='default: 6 (c3), 10 (c4), 1 (c5)')
    ap.add_argument('--tile', type=int, default=4096)
    ap.add_argument('--overlap', type=int, default=1024)
    ap.add_argument('--workers', type=int, default=int(os.environ.get('SHEPSEG_WORKERS', '36')))
    ap.add_argument('--simple-recode', type=int, default=0, help='diagnostic: simpleTileRecode')
    ap.add_argument('--also', type=int, default=1,
                    help='default c3 run on one GPU: append "also": {"c5": ..., "c3_npy": ...} from two short extra '
                         'runs after the headline timing (0 = skip)')
    ap.add_argument('--cpu-sample', type=int, default=14336,
                    help='window edge of the cpu_baseline sample (0 = skip)')
    args = ap.parse_args()
    if args.bands is None:
        args.bands = {'c3': 6, 'c4': 10, 'c5': 1}[args.workload]
    args.seed = 13 if args.workload == 'c4' else 11
    return args


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


def host_window(ras, w):
    """Top-left w x w window of every band of a DeviceRaster as a host array."""
    from pyshepseg_amd import _lib
    idx = np.arange(w, dtype=np.uint32)
    img = np.empty((ras.shape[0], w, w), dtype=ras.dtype)
    c = _lib.ctx()
    c.check(c._L.shp_dev_subsample(c.handle, ctypes.c_void_p(ras.ptr), _lib.SHP_DTYPES[ras.dtype],
                                   ras.shape[0], ras.shape[1], ras.shape[2], _lib.ptr(idx), w,
                                   _lib.ptr(idx), w, _lib.ptr(img)))
    return img


def cpu_baseline(ras, args, centres, msd):
    """The C oracle (a port of the reference, oracle/shepseg_oracle.c) timed on this box's host
    cores on a bounded window of the same image with the same tiling: the tiles of the window one
    per thread over all the cores this process may use (ctypes releases the GIL; the reference
    itself farms tiles to workers the same way, tiling.py:1560-1600), then the stitch; one tile
    alone gives the single-thread rate."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    oracle.build()
    w = min(args.cpu_sample, args.size)
    img = host_window(ras, w)
    tiles, ntc, ntr = oracle.get_tiles(w, w, args.tile, args.overlap)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, len(tiles)))

    def one(key):
        (x, y, xs, ys) = tiles[key]
        sub = np.ascontiguousarray(img[:, y:y + ys, x:x + xs])
        return key, oracle.segment_tile(sub, centres, 50, msd, None, True)['segimg']

    t0 = time.time()
    one((0, 0))
    t_one = time.time() - t0
    (_x, _y, xs0, ys0) = tiles[(0, 0)]
    t0 = time.time()
    order = sorted(tiles, key=lambda k: -(tiles[k][2] * tiles[k][3]))
    with ThreadPoolExecutor(nthreads) as ex:
        local = dict(ex.map(one, order))
    (mosaic, mosaicMax, mosaicHist) = oracle.stitch_tiles(local, tiles, ntc, ntr, w, w, args.overlap)
    dt = time.time() - t0
    check = gpu_window_check(args, w, centres, mosaic, mosaicMax, mosaicHist)
    return {"gpu_equals_oracle_on_sample": check, "value": round(w * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": nthreads, "kind": "port",
            "single_thread_value": round(xs0 * ys0 / t_one / 1e6 / 1.64, 3),
            "sample": "top-left %dx%d window of the same synthetic image, %d tiles (tile %d / overlap %d) "
                      "one per thread on %d threads + stitch: %.1f s of C oracle; single_thread_value = one "
                      "%dx%d tile alone (%.1f s) per output pixel (tiled runs process 1.64 x the image)"
                      % (w, w, len(tiles), args.tile, args.overlap, nthreads, dt, xs0, ys0, t_one)}


def gpu_window_check(args, w, centres, mosaic, mosaicMax, mosaicHist):
    """The checker's other use: the SAME window through the device path (untimed, the model handed in),
    compared bit for bit with the oracle's stitched mosaic of the cpu_baseline leg."""
    from pyshepseg_amd import tiling, shepseg, _lib
    ras = tiling.DeviceRaster.synth(args.seed, args.bands, w, w)
    try:
        r = tiling.doTiledShepherdSegmentation(
            ras, tiling._KEEP_ON_DEVICE, tileSize=args.tile, overlapSize=args.overlap, minSegmentSize=50,
            numClusters=60, kmeansObj=shepseg.KMeansModel(np.ascontiguousarray(centres, dtype=np.float64)),
            concurrencyCfg=tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS,
                                                                numWorkers=min(args.workers, 16)))
        got = np.empty((w, w), dtype=np.uint32)
        c = _lib.ctx()
        c.check(c._L.shp_dev_download(c.handle, _lib.ptr(got), ctypes.c_void_p(r.outDev[0]), got.nbytes))
        tiling.freeDeviceOutput(r)
    finally:
        ras.free()
    ok = (int(r.maxSegId) == int(mosaicMax) and np.array_equal(got, mosaic) and
          np.array_equal(np.asarray(r.hist).astype(np.int64), mosaicHist.astype(np.int64)))
    if not ok:
        sys.stderr.write('bench.py: the device mosaic of the %d x %d cpu_baseline window DIFFERS from the oracle\n' % (w, w))
    return bool(ok)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (this process has not touched the GPU and never does) with the environment a launcher would
    set, relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    import secrets
    nonce = secrets.token_hex(8)            # names this launch's rendezvous files (pyshepseg_amd/comm.py launchTag)
    for r in range(args.gpus):
        env = dict(os.environ, SHEPSEG_LAUNCH_NONCE=nonce, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None, text=True))
    # rank 0's stdout is drained by a thread; all ranks are polled: when one dies the others would sit in a
    # receive for ever, so they are stopped (fresh children of this process), and there is a deadline
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    deadline = time.time() + float(os.environ.get('SHEPSEG_BENCH_DEADLINE', '3000'))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for (r, p) in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else ('deadline', None)
            sys.stderr.write('bench.py: rank %s ended with %s: stopping the other ranks\n' % failed)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(5)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10)
    out0 = out0[0] if out0 else ''
    line = None
    for ln in (out0 or '').splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if any(rcs) or line is None:
        sys.stderr.write('bench.py: the %d-rank run failed (exit codes %s)\n' % (args.gpus, rcs))
        sys.exit(next((rc for rc in rcs if rc), 1))
    if json.loads(line).get('n_gpus') != args.gpus:
        sys.stderr.write('bench.py: the ranks report n_gpus != %d\n' % args.gpus)
        sys.exit(1)
    print(line)


def pmc_traffic(kernel_name, scale=1.0):
    """HBM bytes per launch of a kernel from the newest committed PMC passes (rocprofv3 cannot run inside this
    process): raw FETCH_SIZE + WRITE_SIZE, with the file and the commit the passes were taken at -- see the
    file's own note for the gfx950 corrections (profiles/README.md)."""
    for fn in ('r04_pmc_summary.json', 'r03_pmc_summary.json', 'r02_pmc_summary.json', 'r01_n_pmc_summary.json'):
        try:
            doc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
            k = doc['kernels'].get(kernel_name)
            if k:
                return (int((k['FETCH_SIZE_KB_per_launch'] + k['WRITE_SIZE_KB_per_launch']) * 1024 * scale),
                        '%s @ %s' % (fn, doc.get('taken_at_commit', 'commit not recorded')))
        except Exception:
            pass
    return None, None


def write_npy_image(ras, path):
    """The synthetic image as a band-planar .npy file (row blocks downloaded from the device)."""
    from pyshepseg_amd import _lib
    (nb, nr, nc) = ras.shape
    mm = np.lib.format.open_memmap(path, mode='w+', dtype=ras.dtype, shape=ras.shape)
    c = _lib.ctx()
    rows = max(1, (256 << 20) // (nc * ras.dtype.itemsize))
    buf = np.empty((rows, nc), dtype=ras.dtype)
    for b in range(nb):
        for y0 in range(0, nr, rows):
            y1 = min(nr, y0 + rows)
            v = buf[:y1 - y0]
            c.check(c._L.shp_dev_download(c.handle, _lib.ptr(v), ctypes.c_void_p(
                ras.ptr + ((b * nr + y0) * nc) * ras.dtype.itemsize), v.nbytes))
            mm[b, y0:y1] = v
    mm.flush()
    del mm


def bench_segmentation(args):
    from pyshepseg_amd import tiling, _lib
    ras = tiling.DeviceRaster.synth(args.seed, args.bands, args.size, args.size)
    cfg = tiling.SegmentationConcurrencyConfig(concurrencyType=tiling.CONC_THREADS,
                                               numWorkers=args.workers)
    (src, dst) = (ras, tiling._KEEP_ON_DEVICE)
    if args.source == 'npy':
        src = os.path.join(args.scratch, 'shepseg_bench_%d_%db_seed%d.npy' % (args.size, args.bands, args.seed))
        dst = os.path.join(args.scratch, 'shepseg_bench_%d_out.npy' % args.size)
        if not os.path.exists(src):
            t0 = time.time()
            write_npy_image(ras, src)
            sys.stderr.write('bench.py: wrote %s (%.1f GB) in %.1f s\n' % (src, ras.nbytes / 1e9, time.time() - t0))
        ras.free()                        # the pipeline brings the file into HBM itself

    def step():
        r = tiling.doTiledShepherdSegmentation(
            src, dst, tileSize=args.tile, overlapSize=args.overlap,
            minSegmentSize=50, numClusters=60, fixedKMeansInit=True, concurrencyCfg=cfg,
            simpleTileRecode=bool(args.simple_recode))
        if args.source == 'hbm':
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
    names = {0: 'k_assign (cluster-map blocks)', 1: 'ccl (k_ccl_local+k_ccl_border+k_ccl_flatten)', 2: 'k_dfs_pool',
             3: 'CSR build (k_run_tile_*+k_sort_hist+k_sort_scatter+k_run_expand)', 4: 'k_spectra_small+k_spectra_big',
             5: 'k_small_loop', 7: 'seed scan + k_clump_final'}
    dom = max((i for i in names), key=lambda i: prof.get(i, (0, 0))[0])
    ms, cnt = prof[dom]
    class _Geom(object):
        RasterXSize = RasterYSize = args.size
    ti = tiling.getTilesForFile(_Geom, args.tile, args.overlap)
    tile_px = sum(t[2] * t[3] for t in ti.tiles.values()) / max(len(ti.tiles), 1)
    # algorithmic bytes per launch: clump kernels read a 2-byte cluster id and write a 4-byte label
    # per tile pixel; assign reads nB*2 B and writes 2 B; the rest move (2*nB + 4) B per pixel.
    bpp = {0: 2 * args.bands + 2, 1: 6, 2: 6, 7: 6}.get(dom, 2 * args.bands + 4)
    avg_s = (ms / max(cnt, 1)) / 1e3
    achieved = (bpp * tile_px / avg_s / 1e9) if avg_s > 0 else 0.0
    traffic, traffic_src = (pmc_traffic(names[dom].split(' ')[0]) if args.size == 40000 and args.bands == 6
                            else (None, None))
    wl = args.workload.upper()
    out = {
        "metric": "Mpixels/sec segmented, %d-band 40k x 40k tiled" % args.bands,
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "%s: tiled %dx%d, %d-band uint16 synthimg v1 (seed %d), tileSize=%d, "
                               "overlap=%d, k=60, minSegmentSize=50, fixedKMeansInit, image + "
                               "%s" % (wl, args.size, args.size, args.bands, args.seed,
                                       args.tile, args.overlap,
                                       "labels resident in HBM" if args.source == 'hbm' else
                                       "read from a .npy file through the I/O pipeline (page-locked "
                                       "staging, one upload per pixel overlapped with compute), labels "
                                       "streamed to a .npy file: %.1f GB in + %.1f GB out over PCIe per step"
                                       % (args.bands * npix * 2 / 1e9, npix * 4 / 1e9)),
                   "source": args.source,
                   "tiles": len(ti.tiles), "workers": cfg.numWorkers, "fill_streams": int(os.environ.get('SHEPSEG_FILL_MAX', '6')), "walker_streams": int(os.environ.get('SHEPSEG_WALK_STREAMS', '10')),
                   "max_seg_id": int(r.maxSegId)},
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "traffic_source": traffic_src,
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
    if args.cpu_sample > 0 and args.source == 'hbm':
        out["cpu_baseline"] = cpu_baseline(ras, args, r.kmeans.cluster_centers_,
                                           float(r.maxSpectralDiff))
        out["cpu_baseline"]["reference_numba_mpx_per_core"] = 1.3       # the reference's own path
    if args.source == 'hbm':
        ras.free()
    return out


def also_runs(args):
    """Two short extra runs behind the headline (never part of `value`): BASELINE config 5 (tilingstats, C5) and
    config 3's I/O variant (the image read from / the labels written to .npy files through the pipeline: PCIe and
    the file system inclusive).  Each is its own line of this benchmark in miniature."""
    import copy
    from pyshepseg_amd import tiling
    res = {}
    for (name, over) in (('c5', dict(workload='c5', bands=1, seed=11, steps=2, warmup=1, cpu_sample=0, source='hbm')),
                         ('c3_npy', dict(workload='c3', bands=6, seed=11, steps=2, warmup=1, cpu_sample=0, source='npy'))):
        a = copy.copy(args)
        for (k, v) in over.items():
            setattr(a, k, v)
        try:
            o = bench_stats(a) if name == 'c5' else bench_segmentation(a)
            keep = {k: o[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'roofline') if k in o}
            keep['workload'] = o['config']['workload']
            if name == 'c3_npy':
                keep['pcie_gb_per_s_both_directions'] = round(
                    (a.bands * 2 + 4) * a.size * a.size / 1e9 / (o['ms_per_step'] / 1e3), 2)
                keep['roofline'] = {k: o['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac')}
            res[name] = keep
        except Exception as e:           # an extra must never cost the headline line
            res[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            tiling.clearDeviceCache()
        except Exception:
            pass
    return res


def bench_stats(args):
    """C5: tilingstats on a resident label raster + band."""
    from pyshepseg_amd import tiling, tilingstats, _lib
    N, BH, BW = args.size, 4, 8
    c = _lib.ctx()
    ras = tiling.DeviceRaster.synth(args.seed, 1, N, N)
    d_seg = ctypes.c_void_p()
    c.check(c._L.shp_dev_alloc(c.handle, N * N * 4, ctypes.byref(d_seg)))
    S = ctypes.c_uint32(0)
    c.check(c._L.shp_dev_block_labels(c.handle, N, N, BH, BW, d_seg, ctypes.byref(S)))
    S = S.value
    sel = [('mean', 'mean'), ('sd', 'stddev'), ('med', 'median'), ('n', 'pixcount')]
    fast, ni, nf = tilingstats.makeFastStatsSelection(list(range(len(sel))), sel)
    ic = np.zeros((ni, S + 1), dtype=np.int64)
    fc = np.zeros((nf, S + 1), dtype=np.float32)

    def step():
        c.check(c._L.shp_segstats2d_dev(c.handle, d_seg, ctypes.c_void_p(ras.ptr), 2, N, N, S, 0, 0,
                                        _lib.ptr(fast), len(sel), -9999, _lib.ptr(ic), _lib.ptr(fc)))

    for _ in range(args.warmup):
        step()
    prof_totals([c])
    c.check(c._L.shp_sync(c.handle))
    t0 = time.time()
    for _ in range(args.steps):
        step()
    c.check(c._L.shp_sync(c.handle))
    dt = (time.time() - t0) / max(args.steps, 1)
    ms, cnt = prof_totals([c]).get(8, (0.0, 0))
    assert int(ic[fast[3, 3]].sum()) == N * N
    npix = N * N
    alg = 6 * npix + 4 * len(sel) * (S + 1)            # 6 B/px + 4 * nCols B/segment
    avg_s = (ms / max(cnt, 1)) / 1e3
    achieved = alg / avg_s / 1e9 if avg_s > 0 else 0.0
    out = {
        "metric": "Mpixels/sec, tilingstats per-segment mean/stddev/median/pixcount",
        "value": round(npix / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "C5: %dx%d label raster of %d x %d-pixel blocks (%d segments) + one uint16 "
                               "synthimg v1 band resident in HBM; 4 result columns (%.2f GB) copied to "
                               "pageable host arrays inside the step" % (
                                   N, N, BH, BW, S, (ni * 8 + nf * 4) * (S + 1) / 1e9),
                   "segments": S, "segments_per_s": round(S / dt, 0)},
        "roofline": {"bound": "hbm", "kernel": "segstats device pipeline (k_label_hist + k_stats_prefill + k_stats_patch; "
                                               "the sorts only for segments that straddle 32 x 64-pixel patches)",
                     "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                     "avg_launch_ms": round(ms / max(cnt, 1), 3), "launches": int(cnt),
                     "bytes_per_launch": int(alg)},
    }
    if args.cpu_sample > 0:
        from oracle import oracle
        oracle.build()
        w = min(4096, N)
        band = host_window(ras, w)[0]
        lab = (((np.arange(w, dtype=np.uint32) // BH)[:, None] * np.uint32((w + BW - 1) // BW)) +
               (np.arange(w, dtype=np.uint32) // BW)[None, :] + np.uint32(1))
        t0 = time.time()
        oracle.segstats(np.ascontiguousarray(lab), band, sel)
        t = time.time() - t0
        out["cpu_baseline"] = {"value": round(w * w / t / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                               "kind": "port",
                               "sample": "top-left %dx%d window (%d segments), %.1f s of single-thread C "
                                         "oracle (orc_segstats)" % (w, w, int(lab.max()), t)}
    c.check(c._L.shp_dev_free(c.handle, d_seg))
    ras.free()
    return out


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
    force_dist = os.environ.get('SHEPSEG_FORCE_DIST', '0') == '1' and 'RANK' in os.environ
    if world > 1 or force_dist:
        # RCCL brings streams of its own: two walker streams fewer keep the process inside its hardware queues
        # (LABNOTES, round 4: 522 -> 477 ms per step at world 1; must be set before the library loads)
        if os.environ.get('SHEPSEG_COMM', 'rccl') == 'rccl':
            os.environ.setdefault('SHEPSEG_WALK_STREAMS', '8')
        from pyshepseg_amd import distributed
        if args.workload == 'c5':
            return distributed.bench_stats_main(args, rank, world, local_rank)
        return distributed.bench_main(args, rank, world, local_rank)
    if args.workload == 'c5':
        out = bench_stats(args)
    else:
        out = bench_segmentation(args)
        if args.also and args.workload == 'c3' and args.source == 'hbm' and args.size == 40000:
            out["also"] = also_runs(args)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
