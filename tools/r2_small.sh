R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
SHEPSEG_SMALL_TIMING=1 timeout -k 10 300 python tools/perf_tile.py 4096 > gpurun_out/r2_small.log 2>&1; grep -B60 "^rep 2" gpurun_out/r2_small.log | grep -A60 "^rep 1" | head -70
