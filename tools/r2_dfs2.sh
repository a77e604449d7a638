# default bench under a few walker-pool shapes (walkers per workgroup, 2-KiB granules in the pool)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "SHEPSEG_DFS_PER_WG=8,SHEPSEG_DFS_POOL=34" "SHEPSEG_DFS_PER_WG=4,SHEPSEG_DFS_POOL=24" "SHEPSEG_DFS_PER_WG=4,SHEPSEG_DFS_POOL=18" "SHEPSEG_DFS_PER_WG=2,SHEPSEG_DFS_POOL=17" "SHEPSEG_DFS_PER_WG=1,SHEPSEG_DFS_POOL=17"; do
  envs=$(echo "$spec" | tr ',' ' ')
  echo "== $spec"
  env $envs timeout -k 10 200 python bench.py --cpu-sample 0 --steps 4 > gpurun_out/dfs2.log 2>&1 || { tail -5 gpurun_out/dfs2.log; exit 1; }
  tail -1 gpurun_out/dfs2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_s'], d['roofline']['avg_launch_ms'], d['roofline']['device_ms_by_kernel'])"
done
