# parity tests, then single-tile stage timings, then the default bench
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; rc=$?; tail -5 gpurun_out/tg.log
[ $rc = 0 ] || exit 1
SHEPSEG_SMALL_TIMING=1 timeout -k 10 300 python tools/perf_tile.py 4096 > gpurun_out/r2_tile.log 2>&1 && tail -4 gpurun_out/r2_tile.log &&
timeout -k 10 400 python bench.py --cpu-sample 0 --steps 5 > gpurun_out/bd.log 2>&1; tail -1 gpurun_out/bd.log | cut -c1-1800
