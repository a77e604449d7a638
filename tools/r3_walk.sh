# round 3: the register-window replay (dfs_split_win) -- parity tests of the clump stage, then the
# per-component statistics of one 4096^2 tile with the old and the new walk (+ the DFS_PROF build)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_tile.py -x -q -m gpu > gpurun_out/r3_walk_tests.log 2>&1 || { tail -30 gpurun_out/r3_walk_tests.log; exit 1; }
tail -2 gpurun_out/r3_walk_tests.log
for spec in SHEPSEG_DFS_OLDWALK=1 SHEPSEG_DFS_OLDWALK=0 SHEPSEG_LIBPATH=$R/pyshepseg_amd/libshepseg_hip_prof.so; do
  echo "== $spec"
  env $spec SHEPSEG_DFS_STATS=1 timeout -k 10 120 python tools/perf_tile.py 4096 > gpurun_out/r3_dfs.log 2>&1 || { tail -5 gpurun_out/r3_dfs.log; exit 1; }
  grep -A4 "^dfs:" gpurun_out/r3_dfs.log | tail -5; grep "prof\|asm runs" gpurun_out/r3_dfs.log | tail -3; grep "^rep 2" gpurun_out/r3_dfs.log
done
