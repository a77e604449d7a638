# default bench under "ENV=..,ENV=..:bench args" settings: bash tools/r2_sweep.sh "A=1,B=2:--workers 24" ...
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "$@"; do
  envs=$(echo "${spec%%:*}" | tr ',' ' ')
  args="${spec#*:}"
  [ "$args" = "$spec" ] && args=""
  echo "== $spec"
  env $envs timeout -k 10 200 python bench.py --cpu-sample 0 --steps 4 $args > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  tail -1 gpurun_out/sweep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['device_ms_by_kernel']; print(d['ms_per_step'], d['config']['step_s'], 'dfs %.1f small %.1f ms/tile' % (k['k_dfs_pool']/144, k['k_small_loop']/144))"
done
