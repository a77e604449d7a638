# round 3: phase profile of the register-window replay (DFS_PROF build) on one 4096^2 tile
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
env SHEPSEG_LIBPATH=$R/pyshepseg_amd/libshepseg_hip_prof.so SHEPSEG_DFS_STATS=1 timeout -k 10 300 python tools/perf_tile.py 4096 > gpurun_out/r3_walkprof.log 2>&1 || { tail -5 gpurun_out/r3_walkprof.log; exit 1; }
grep -A9 "^dfs:" gpurun_out/r3_walkprof.log | tail -10; grep "^rep 2" gpurun_out/r3_walkprof.log
