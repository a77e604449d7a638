# DFS walker statistics on one 4096^2 tile 
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "$@"; do
  echo "== $spec"
  env $(echo $spec | tr ',' ' ') SHEPSEG_DFS_STATS=1 timeout -k 10 300 python tools/perf_tile.py 4096 > gpurun_out/r2_dfs.log 2>&1 || { tail -5 gpurun_out/r2_dfs.log; exit 1; }
  grep -A5 "^dfs:" gpurun_out/r2_dfs.log | tail -6; grep "^rep 2" gpurun_out/r2_dfs.log
done
