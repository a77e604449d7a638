# DFS walker statistics on one 4096^2 tile under a few pool settings
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "A=0" "SHEPSEG_DFS_PER_WG=4" "SHEPSEG_DFS_PER_WG=2" "SHEPSEG_DFS_PER_WG=1"; do
  echo "== $spec"
  env $spec SHEPSEG_DFS_STATS=1 timeout -k 10 300 python tools/perf_tile.py 4096 0 > gpurun_out/r2_dfs.log 2>&1 || { tail -5 gpurun_out/r2_dfs.log; exit 1; }
  grep -A13 "^dfs:" gpurun_out/r2_dfs.log | tail -14; grep "^rep 2" gpurun_out/r2_dfs.log
done
