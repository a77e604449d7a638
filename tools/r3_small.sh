# round 3: the pass loop -- parity tests of the elimination stages, then its per-pass timing on one 4096^2 tile
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_tile.py tests/test_gpu_tiling.py -x -q -m gpu > gpurun_out/r3_small_tests.log 2>&1 || { tail -30 gpurun_out/r3_small_tests.log; exit 1; }
tail -2 gpurun_out/r3_small_tests.log
SHEPSEG_SMALL_TIMING=1 timeout -k 10 120 python tools/perf_tile.py 4096 > gpurun_out/r3_small.log 2>&1 || { tail -5 gpurun_out/r3_small.log; exit 1; }
grep "^small loop" gpurun_out/r3_small.log | tail -1; grep "  pass " gpurun_out/r3_small.log | tail -49 | awk 'NR%6==1'; grep "^rep 2" gpurun_out/r3_small.log
