# diagnostic: how fast is the step when the latency-bound phases are taken out (labels are wrong)?
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "A=0" "SHEPSEG_DBG_SKIP_DFS=1" "SHEPSEG_DBG_SKIP_SMALL=1" "SHEPSEG_DBG_SKIP_DFS=1,SHEPSEG_DBG_SKIP_SMALL=1" "SHEPSEG_DBG_SKIP_DFS=1,SHEPSEG_DBG_SKIP_SMALL=1,SHEPSEG_FILL_MAX=0" "SHEPSEG_DBG_SKIP_DFS=1,SHEPSEG_DBG_SKIP_SMALL=1,SHEPSEG_FILL_MAX=2"; do
  envs=$(echo "$spec" | tr ',' ' ')
  echo "== $spec"
  env $envs timeout -k 10 200 python bench.py --cpu-sample 0 --steps 3 > gpurun_out/skip.log 2>&1 || { tail -5 gpurun_out/skip.log; exit 1; }
  tail -1 gpurun_out/skip.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_s'], d['config']['host_timers_s'])"
done
