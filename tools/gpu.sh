#!/bin/bash
# The measurement runs of this repository, one task per invocation on the GPU box:
#     gpurun -- 'bash tools/gpu.sh TASK [ARGS...]'          (results under gpurun_out/)
# Each task is what one of the round-1..3 one-off scripts did; see tools/README.md.
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
task=$1; shift
case "$task" in
tests)      # every GPU test, then smoke()
    timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -3 gpurun_out/tg.log
    timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -2 ;;
bench)      # the default bench (C3), then C4, C5 and the file-source variant with "$@" passed on
    for wl in "" "--workload c4" "--workload c5" "--source npy"; do
      timeout -k 10 600 python bench.py $wl "$@" > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-400
    done ;;
sweep)      # the default bench under settings "ENV=..,ENV=..:bench args" ...   (knobs: README.md)
    i=0
    for spec in "$@"; do
      i=$((i+1)); envs=$(echo "${spec%%:*}" | tr ',' ' '); args="${spec#*:}"
      echo "== $spec"
      env $envs timeout -k 10 300 python bench.py --cpu-sample 0 --steps 3 $args > gpurun_out/sweep_$i.log 2>&1
      tail -1 gpurun_out/sweep_$i.log | cut -c60-160
    done ;;
walk)       # the replay walker: clump parity tests, then per-component statistics of one 4096^2 tile with the
            # old walk, the register-window walk and its phase profile (make PROF=1 OUT=../libshepseg_hip_prof.so)
    timeout -k 10 300 python -m pytest tests/test_gpu_tile.py -x -q -m gpu > gpurun_out/walk_tests.log 2>&1 || { tail -30 gpurun_out/walk_tests.log; exit 1; }
    tail -2 gpurun_out/walk_tests.log
    for spec in SHEPSEG_DFS_OLDWALK=1 SHEPSEG_DFS_OLDWALK=0 SHEPSEG_LIBPATH=$R/pyshepseg_amd/libshepseg_hip_prof.so; do
      [ "${spec#SHEPSEG_LIBPATH}" != "$spec" ] && [ ! -f "${spec#SHEPSEG_LIBPATH=}" ] && continue
      echo "== $spec"
      env $spec SHEPSEG_DFS_STATS=1 timeout -k 10 120 python tools/perf_tile.py ${1:-4096} > gpurun_out/walk.log 2>&1 || { tail -5 gpurun_out/walk.log; exit 1; }
      grep -A16 "^dfs:" gpurun_out/walk.log | tail -17 | grep -v "^  rank [2-9]\|^  rank 1[0-9]"; grep "^rep 2" gpurun_out/walk.log
    done ;;
small)      # the pass loop: elimination parity tests, then its per-pass timing on one tile
    timeout -k 10 400 python -m pytest tests/test_gpu_tile.py tests/test_gpu_tiling.py -x -q -m gpu > gpurun_out/small_tests.log 2>&1 || { tail -30 gpurun_out/small_tests.log; exit 1; }
    tail -2 gpurun_out/small_tests.log
    SHEPSEG_SMALL_TIMING=1 timeout -k 10 120 python tools/perf_tile.py ${1:-4096} > gpurun_out/small.log 2>&1 || { tail -5 gpurun_out/small.log; exit 1; }
    awk '/^small loop/{buf=""} {buf=buf $0 "\n"} END{printf "%s", buf}' gpurun_out/small.log | head -56 ;;
fit)        # the whole-image k-means fit on the benchmark sample: Elkan trace, timings, kernel statistics
    SHEPSEG_FIT_TRACE=1 timeout -k 10 200 python tools/perf_fit.py > gpurun_out/fit_trace.log 2>&1; grep "elkan batch" gpurun_out/fit_trace.log | awk 'NR%5==1' | head -12
    rm -rf gpurun_out/fitprof
    SHEPSEG_FIT_TIMING=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fitprof -- python3 tools/perf_fit.py 40000 6 > gpurun_out/fitprof.log 2>&1
    python3 tools/kstats.py --top 12 "$(find gpurun_out/fitprof -name '*kernel_stats.csv' | head -1)"
    grep "kmeans fit: n=" gpurun_out/fitprof.log | tail -1
    for kn in "k_elk2_filter<0>" k_elk2_visit; do
      python3 tools/kstats.py --calls "$kn" "$(find gpurun_out/fitprof -name '*kernel_trace.csv' | head -1)" 10 | cut -c1-700
    done
    rm -f gpurun_out/fitprof/*/*kernel_trace.csv ;;
fit-shard)  # one rank's share of the row-sharded E-step (fit_elkan.h FitShard): the benchmark fit with rank 0 of N (= $1,
            # default 8) played alone -- its filter / visit on n / N rows, the full M-step (the other shards' labels go
            # stale: a timing, not a fit); then the whole fit with ONE process playing all N ranks (bit-exact: tests)
    rm -rf gpurun_out/fitprof
    SHEPSEG_FIT_SHARDS=${1:-8} SHEPSEG_FIT_SHARD_ONLY=0 SHEPSEG_FIT_TIMING=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fitprof -- python3 tools/perf_fit.py 40000 6 > gpurun_out/fitprof.log 2>&1
    python3 tools/kstats.py --top 9 "$(find gpurun_out/fitprof -name '*kernel_stats.csv' | head -1)"
    grep "kmeans fit: n=" gpurun_out/fitprof.log | tail -1; rm -f gpurun_out/fitprof/*/*kernel_trace.csv
    SHEPSEG_FIT_SHARDS=${1:-8} SHEPSEG_FIT_TIMING=1 timeout -k 10 300 python3 tools/perf_fit.py 40000 6 2>&1 | grep "kmeans fit: n=" | tail -1 ;;
tiletrace)  # kernel sequence (durations, gaps) of ONE tile run alone -> gpurun_out/tiletrace.txt
    rm -rf gpurun_out/tt
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o run -- python tools/perf_tile.py ${1:-4096} > gpurun_out/tt.log 2>&1 &&
    python3 tools/kstats.py --sequence "$(ls gpurun_out/tt/*kernel_trace.csv gpurun_out/tt/*/*kernel_trace.csv 2>/dev/null | head -1)" > gpurun_out/tiletrace.txt
    rm -rf gpurun_out/tt; cat gpurun_out/tiletrace.txt ;;
two-ranks)  # rehearsal of `bench.py --gpus 2`: two ranks share the one GPU, strips over sockets (not a scaling datum)
    for m in parallel sequential; do
      SHEPSEG_COMM=socket SHEPSEG_DEVICE=0 SHEPSEG_STITCH=$m timeout -k 10 500 python bench.py --gpus 2 --workers 12 --steps 2 --cpu-sample 0 > gpurun_out/two_$m.log 2>&1 || { tail -20 gpurun_out/two_$m.log; exit 1; }
      grep "^{" gpurun_out/two_$m.log | tail -1 | cut -c1-1400
    done
    # the same for the sharded statistics (C5): two ranks over sockets (device buffers staged through the host),
    # then one rank under RCCL itself (ncclAllGather / ncclAllReduce of the device buffers at world size 1)
    SHEPSEG_COMM=socket SHEPSEG_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --workload c5 --steps 2 --cpu-sample 0 > gpurun_out/two_c5.log 2>&1 || { tail -20 gpurun_out/two_c5.log; exit 1; }
    grep "^{" gpurun_out/two_c5.log | tail -1 | cut -c1-1400
    SHEPSEG_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout -k 10 300 python bench.py --gpus 1 --workload c5 --steps 3 --cpu-sample 0 > gpurun_out/one_c5.log 2>&1 || { tail -20 gpurun_out/one_c5.log; exit 1; }
    grep "^{" gpurun_out/one_c5.log | tail -1 | cut -c1-1400 ;;
dump-sample) # the k-means sub-sample of a benchmark raster for oracle/refgen/gen_golden_c3_fit.py: dump-sample SEED BANDS NAME
    timeout -k 10 300 python - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from pyshepseg_amd import tiling, shepseg
(seed, nb, name) = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
ras = tiling.DeviceRaster.synth(seed, nb, 40000, 40000)
img = tiling.readSubsampledImage(ras, list(range(1, nb + 1)), np.sqrt(1e6 / (40000 * 40000)))
np.save('gpurun_out/%s_sample.npy' % name, img)
km = shepseg.fitSpectralClusters(img, 60, 100, None, True)
print(name, 'fit', km.n_iter_, km.fit_path_, float(km.cluster_centers_.sum()))
np.save('gpurun_out/%s_centres_device.npy' % name, km.cluster_centers_)
PY
    ;;
ubench)     # micro-benchmarks of instruction issue / branch cost / float64 chains for a lone wavefront
    for b in ${@:-issue branch f64chain mfma_f64_order mfma_f32_order}; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/ub_$b tools/ubench/$b.hip 2>/dev/null && timeout -k 5 60 /tmp/ub_$b
    done ;;
sumlists)   # the M-step's row-order sums alone, with the adding wavefront's cycle split (tools/ubench/sumlists.hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -w -I pyshepseg_amd/csrc -o /tmp/ub_sumlists tools/ubench/sumlists.hip &&
    timeout -k 5 120 /tmp/ub_sumlists ;;
*) echo "tasks: tests bench sweep walk small fit fit-shard tiletrace two-ranks dump-sample ubench sumlists (and tools/refresh_profiles.sh TAG)"; exit 2 ;;
esac
